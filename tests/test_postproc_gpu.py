"""GPU parity of the connected-component size filter (csrc/postproc.hip) against scipy.ndimage.label + bincount, which is
what skimage.morphology.remove_small_objects (behind MONAI's RemoveSmallObjects) does. Integer work: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(mask, min_size, connectivity):
    from scipy import ndimage
    structure = ndimage.generate_binary_structure(2, connectivity)
    out = np.zeros_like(mask)
    for b in range(mask.shape[0]):
        lab, _ = ndimage.label(mask[b], structure)
        sizes = np.bincount(lab.ravel())
        keep = sizes >= min_size
        keep[0] = False
        out[b] = keep[lab]
    return out.astype(np.uint8)


@pytest.mark.parametrize("connectivity", [1, 2])
def test_remove_small_objects_matches_scipy(hip_lib_built, connectivity):
    import torch
    from octa_autosegmentation_amd.models import postprocess
    rng = np.random.default_rng(5 + connectivity)
    masks = []
    for dens in (0.35, 0.5, 0.62):                     # around the percolation threshold: components of every size
        masks.append(rng.random((97, 131)) < dens)
    m = np.stack(masks).astype(np.uint8)
    got = postprocess.remove_small_objects_device(torch.from_numpy(m).cuda(), 20, connectivity).cpu().numpy()
    assert (got == _ref(m, 20, connectivity)).all()
    # vessel-like mask from the rasteriser fixture, the config's min_size
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_golden.npz"))
    lab = np.unpackbits(np.asarray(g["graph0_label_packed"])).reshape(1, 1216, 1216).astype(np.uint8)
    noise = (rng.random(lab.shape) < 0.02).astype(np.uint8)
    mm = np.maximum(lab, noise)
    got = postprocess.remove_small_objects_device(torch.from_numpy(mm).cuda(), 160, connectivity).cpu().numpy()
    assert (got == _ref(mm, 160, connectivity)).all()


def test_postprocess_list_from_config(hip_lib_built):
    import torch
    from octa_autosegmentation_amd.models import postprocess
    cfg = [{"name": "Activations", "sigmoid": True}, {"name": "AsDiscrete", "threshold": 0.5}, {"name": "RemoveSmallObjects", "min_size": 160}]
    g = torch.Generator(device="cuda").manual_seed(2)
    logits = torch.randn(2, 1, 200, 200, device="cuda", generator=g)
    logits[:, :, 50:90, 40:120] += 4.0                 # one big blob per image
    out = postprocess.postprocess_prediction(logits, cfg)
    assert out.dtype == torch.uint8 and out.shape == logits.shape
    want = _ref((logits[:, 0].cpu().numpy() >= 0).astype(np.uint8), 160, 1)
    assert (out[:, 0].cpu().numpy() == want).all()
