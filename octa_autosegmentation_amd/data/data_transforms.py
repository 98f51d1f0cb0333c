"""Dictionary transforms of the training / validation / test configs, without MONAI (reference data/data_transforms.py;
registry :587-611). Every transform keeps its YAML name and constructor arguments and is a callable dict -> dict, so the
reference's config lists drive them unchanged. Tensors live on the GPU from the moment they are loaded: samples are
transformed where they will be consumed instead of in CPU loader workers.

Two groups:
* the reference's OWN transforms on the hot path -- LoadGraphAndFilterByRandomRadiusd (:358-387), ToGrayScaled (:389-400),
  SpeckleBrightnesd (:25-42), AddRandomBackgroundNoised (:498-516), ImageToImageTranslationd (:327-356);
* the MONAI transforms those configs name (LoadImaged, ScaleIntensityd, EnsureChannelFirstd, Resized, RandFlipd, RandRotate90d,
  RandRotated, Rotate90d, Flipd, AsDiscreted, CastToTyped; post-processing: Activations, AsDiscrete, RemoveSmallObjects,
  CastToType), restated from MONAI's documented behaviour. MONAI is neither in the image nor under /root/reference:
  parity with MONAI itself -- in particular its per-transform random streams -- is UNPINNED; the geometry is pinned against
  the torch ops MONAI delegates to (tests/test_augment_gpu.py, tests/test_data_pipeline.py).
"""
import csv
import pickle
import random as _py_random

import numpy as np
import torch

from .._native import advance_python_random
from ..vessel_graph_generation import tree2img
from ..vessel_graph_generation.tree2img import rasterize_forest


_GRAPH_CACHE = {}          # (path, mtime, size) -> (edges float64 [n,7] on the host, the same on the device)
_GRAPH_CACHE_MAX = 4096


def load_graph_cached(path):
    """A graph CSV parsed once per process (native reader) and kept on the device: an epoch of the training loop re-reads every
    file (the reference parses 1.2 MB of text per sample and epoch in its loader workers)."""
    import os
    from .. import graph_io
    st = os.stat(path)
    key = (path, st.st_mtime_ns, st.st_size)
    hit = _GRAPH_CACHE.get(key)
    if hit is None:
        e = graph_io.read_csv_native(path)
        hit = (e, torch.from_numpy(e).to(default_device()))
        if len(_GRAPH_CACHE) >= _GRAPH_CACHE_MAX:
            _GRAPH_CACHE.pop(next(iter(_GRAPH_CACHE)))
        _GRAPH_CACHE[key] = hit
    return hit


def _as_keys(keys):
    return [keys] if isinstance(keys, str) else list(keys)


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class MapTransform:
    """dict -> dict transform over `keys` (MONAI's MapTransform convention: a missing key is an error unless
    allow_missing_keys)."""

    def __init__(self, keys, allow_missing_keys: bool = False):
        self.keys = _as_keys(keys)
        self.allow_missing_keys = allow_missing_keys

    def present(self, data):
        for k in self.keys:
            if k in data:
                yield k
            elif not self.allow_missing_keys:
                raise KeyError(f"{type(self).__name__}: key {k!r} is missing from the sample ({sorted(data)})")


class Randomizable:
    """Owns a numpy RandomState `R`; `set_random_state(seed)` as in MONAI (get_data_augmentations seeds every random
    transform with General.seed, data_transforms.py:606-607)."""

    R = np.random.RandomState()

    def set_random_state(self, seed=None, state=None):
        self.R = state if state is not None else np.random.RandomState(seed)
        return self


class Compose:
    def __init__(self, transforms=None):
        self.transforms = list(transforms or [])

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data

    def batchable(self, has_background=True):
        """True when the chain holds a transform with a mini-batch form (`batch_apply`: the frozen generator of
        ImageToImageTranslationd) and may be cut there without changing any random decision. "All prefixes, one batched call, all
        suffixes" runs sample 2's prefix BEFORE sample 1's suffix, so it consumes every SHARED random stream in the per-sample
        order only if no such stream is drawn from on both sides of the cut. Every transform declares its streams
        (`rng_streams(has_background)`: "python" = the global `random`, "numpy" = numpy's global stream, "torch" = torch's global
        generator; a transform's own RandomState is private and never conflicts); a transform that declares nothing counts as
        deterministic only if it is not Randomizable and has no `R` -- an undeclared Randomizable that was never seeded still draws from
        the CLASS-level `Randomizable.R`, which every unseeded instance shares: that is a shared stream ("class_R")."""
        cut = [i for i, t in enumerate(self.transforms) if hasattr(t, "batch_apply")]
        if len(cut) != 1:
            return False

        def streams(ts):
            out = set()
            for t in ts:
                f = getattr(t, "rng_streams", None)
                if f is not None:
                    out |= set(f(has_background))
                elif isinstance(t, Randomizable) or hasattr(t, "R"):
                    if getattr(t, "R", None) is Randomizable.R:      # never seeded: the class-level RandomState all unseeded instances share
                        out.add("class_R")
            return out

        return not (streams(self.transforms[:cut[0]]) & streams(self.transforms[cut[0] + 1:]))

    def call_batch(self, items):
        """[sample dict] -> [sample dict]: the chain applied to a mini-batch, the `batch_apply` transform once for all samples."""
        cut = next(i for i, t in enumerate(self.transforms) if hasattr(t, "batch_apply"))
        samples = []
        for data in items:
            for t in self.transforms[:cut]:
                data = t(data)
            samples.append(data)
        samples = self.transforms[cut].batch_apply(samples)
        out = []
        for data in samples:
            for t in self.transforms[cut + 1:]:
                data = t(data)
            out.append(data)
        return out


# ---- the reference's own transforms ---------------------------------------------------------------------------------

class LoadGraphAndFilterByRandomRadiusd(MapTransform):
    """Graph CSV -> grey image per key, key i rendered at image_resolutions[i] with min_radius[i]; the first key creates the
    dropout `blackdict`, later keys reuse it (reference :358-387). Rendering is the HIP rasteriser (bit-exact with the
    reference's matplotlib/Agg output); tensors are returned on the GPU.

    Fast path (max_dropout_prob == 0, no blackdict file -- the segmentation configs): the CSV is parsed by the native
    reader (octa_csv_parse_edges, the reference's "Legacy" string branch of tree2img.py:73-76 with strtod) and the radius
    window is applied on the device; the global `random` stream is advanced exactly as the reference's loop would (one
    draw for the dropout probability on the first key, one draw per in-range edge on every key).
    """

    def __init__(self, keys, allow_missing_keys: bool = False, image_resolutions=[[304, 304]], min_radius=[0], max_dropout_prob=0,
                 MIP_axis=2) -> None:
        super().__init__(keys, allow_missing_keys)
        self.min_radius = min_radius
        self.image_resolutions = image_resolutions
        self.max_dropout_prob = max_dropout_prob
        self.MIP_axis = MIP_axis

    def rng_streams(self, has_background=True):
        return {"python"}           # tree2img.py:62,78: the dropout draws come from the global `random`

    def __call__(self, data):
        data = dict(data)
        if "blackdict" in data:
            with open(data["blackdict"], mode="rb") as file:
                blackdict = pickle.load(file)
        else:
            blackdict = None
        fast = self.max_dropout_prob == 0 and blackdict is None and torch.cuda.is_available()
        for i, key in enumerate(self.keys):
            if key not in data:
                if self.allow_missing_keys:
                    continue
                raise KeyError(f"LoadGraphAndFilterByRandomRadiusd: key {key!r} is missing")
            path = data[key]
            if fast:
                e, d_edges = load_graph_cached(path)
                lo = float(self.min_radius[i])
                n_in = int(np.count_nonzero((e[:, 6] >= lo) & (e[:, 6] <= 1.0)))
                # p = random() ** 10 * max_dropout_prob on the first key (tree2img.py:62), then `random() < p` per surviving edge (:78)
                advance_python_random(n_in + (1 if i == 0 else 0))
                img = tree2img.rasterize_edges_device(d_edges, np.array([0, len(e)]), self.image_resolutions[i], self.MIP_axis,
                                                      min_radius=lo, max_radius=1.0)[0]
                data[key] = img.to(torch.float32)
            else:
                with open(path, newline='') as csvfile:
                    f = list(csv.DictReader(csvfile))
                img, blackdict = rasterize_forest(f, self.image_resolutions[i], self.MIP_axis, min_radius=self.min_radius[i],
                                                  max_dropout_prob=self.max_dropout_prob, blackdict=blackdict)
                data[key] = torch.tensor(img.astype(np.float32)).to(default_device())
        return data


class ToGrayScaled(MapTransform):
    """RGB -> grey with Pillow's "L" weights, (19595 R + 38470 G + 7471 B + 0x8000) >> 16 on uint8-truncated values
    (reference :389-400 goes through PIL.Image.convert("L")); single-channel inputs pass through truncated to uint8."""

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            x = data[key]
            u = x.to(torch.uint8).to(torch.int64)         # astype(np.uint8): truncation, wraps like numpy for out-of-range values
            if x.dim() == 3 and x.shape[-1] in (3, 4):
                u = (19595 * u[..., 0] + 38470 * u[..., 1] + 7471 * u[..., 2] + 0x8000) >> 16
            data[key] = u.to(torch.float32)
        return data


class SpeckleBrightnesd(MapTransform):
    """Speckle component of the noise model (reference :25-42): a 9x9 control grid in [0.5, 1) bilinearly upsampled to the
    image, R = C - U (1 - C), img * R, then divided by the max and shifted by the min. The two random tensors come from
    torch's CPU generator exactly as in the reference (same values for the same torch.manual_seed), the arithmetic runs
    on the tensor's device."""

    def rng_streams(self, has_background=True):
        return {"torch"}

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            img = data[key]
            c = torch.rand((1, 1, 9, 9)) * 0.5 + 0.5
            if img.is_cuda and img.dim() == 3 and img.shape[0] == 1:
                from .gpu_augment import speckle_brightness             # csrc/augment.hip: two streaming passes, same draws
                u = torch.rand((1,) + tuple(img.shape[-2:]))
                data[key] = speckle_brightness(img.float(), c.reshape(1, 9, 9).to(img.device), u.to(img.device)).to(img.dtype)
                continue
            C = torch.nn.functional.interpolate(c, size=img.shape[-2:], mode="bilinear").squeeze(0)
            R = C - (torch.rand_like(C) * (1 - C))
            img = img * R.to(img.device)
            img = img / img.max()
            img = img - img.min()
            data[key] = img
        return data


class AddRandomBackgroundNoised(MapTransform):
    """img = max(img, background * U(0,1)) with uniform noise standing in for a missing background tile (reference
    :498-516); the per-pixel factors come from numpy's global stream, as in the reference."""

    def __init__(self, keys, delete_background=True) -> None:
        super().__init__(keys, True)
        self.delete_background = delete_background

    def rng_streams(self, has_background=True):
        return {"numpy"} if has_background else {"numpy", "torch"}      # torch.rand stands in for a missing background tile

    def __call__(self, data):
        data = dict(data)
        for key in self.keys:
            if key in data:
                img = data[key]
                # (torch.rand stands in for a missing tile: this transform then shares torch's stream with SpeckleBrightnesd and
                # must not be reordered against it -- image_dataset.ListDataset.get_batch checks for the key)
                noise = data["background"].to(img.device) if "background" in data else torch.rand(img.shape).to(img.device)
                speckle = torch.from_numpy(np.random.uniform(0, 1, tuple(img.shape))).to(img.device)
                if img.is_cuda and img.dtype == torch.float32 and noise.dtype == torch.float32 and noise.shape == img.shape:
                    from .gpu_augment import background_noise           # csrc/augment.hip
                    data[key] = background_noise(img, noise, speckle)
                else:
                    data[key] = torch.maximum(img, noise * speckle)      # float64 product, as torch promotes in the reference
        if self.delete_background and "background" in data:
            del data["background"]
        return data


class ImageToImageTranslationd(MapTransform):
    """Frozen-generator contrast adaptation (reference :327-356) -- on the GPU (the reference runs resnetGenerator9 inside
    CPU loader workers: 6.1 s per image). `model` may be passed directly; otherwise resnetGenerator9 weights
    are loaded from `model_path` (checkpoint dict with key 'model')."""

    def __init__(self, model_path=None, keys=("image",), model_config: dict = None, allow_missing_keys: bool = False,
                 model=None, device=None, amp=None) -> None:
        super().__init__(keys, allow_missing_keys)
        from ..models.base_model_abc import load_checkpoint_file
        from ..models.networks import MODEL_DICT
        if model is None:
            if model_config is not None and model_config.get("name", "resnetGenerator9") != "resnetGenerator9":
                raise NotImplementedError("only resnetGenerator9 translation models are on the MI355X hot path")
            model = MODEL_DICT["resnetGenerator9"]()
            if model_path is not None:
                ckpt = load_checkpoint_file(model_path, "cpu")
                model.load_state_dict(ckpt["model"])
                import sys
                print(f"Loaded network weights from epoch {ckpt['epoch']}.", file=sys.stderr)
        self.device = torch.device(device) if device is not None else default_device()
        self.model = model.to(self.device).eval().requires_grad_(False)
        # bf16 autocast on the GPU: the generator's 3x3 stages run the MFMA convolution kernels and its stems the thin-conv
        # kernels (models/networks.py), as in the GAN-seg training step; `amp: false` keeps the fp32 torch modules
        self.amp = (self.device.type == "cuda") if amp is None else (bool(amp) and self.device.type == "cuda")

    def _translate(self, x):
        with torch.no_grad(), torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=self.amp):
            return self.model(x.float().to(self.device)).float()

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            data[key] = self._translate(data[key].unsqueeze(0)).squeeze(0)
        return data

    def batch_apply(self, samples):
        """The same transform on a list of samples with ONE generator pass per key over the stacked mini-batch (InstanceNorm is
        per sample, so each image gets what its own pass would give)."""
        samples = [dict(d) for d in samples]
        for key in self.keys:
            have = [d for d in samples if key in d]
            if len(have) != len(samples) and not self.allow_missing_keys:
                raise KeyError(f"ImageToImageTranslationd: key {key!r} is missing from a sample")
            if not have:
                continue
            if len({tuple(d[key].shape) for d in have}) == 1:
                y = self._translate(torch.stack([d[key] for d in have]))
                for i, d in enumerate(have):
                    d[key] = y[i]
            else:
                for d in have:
                    d[key] = self._translate(d[key].unsqueeze(0)).squeeze(0)
        return samples


# ---- MONAI transforms named by the configs (restated; see the module docstring) ------------------------------------------------

class LoadImaged(MapTransform):
    """PNG -> float32 tensor on the GPU. MONAI's PILReader hands images over with the first two axes swapped (its
    `reverse_indexing`: [W, H(, C)]); the configs undo that with Rotate90d(k=1) + Flipd(spatial_axis=0), so the swap is kept."""

    def __init__(self, keys, allow_missing_keys: bool = False, image_only: bool = True, **unused) -> None:
        super().__init__(keys, allow_missing_keys)

    def __call__(self, data):
        from PIL import Image
        data = dict(data)
        for key in self.present(data):
            a = np.asarray(Image.open(data[key]))
            if a.dtype == bool:
                a = a.astype(np.uint8)
            a = np.swapaxes(a, 0, 1)
            data[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(default_device())
        return data


class ScaleIntensityd(MapTransform):
    def __init__(self, keys, minv=0.0, maxv=1.0, allow_missing_keys: bool = False, **unused) -> None:
        super().__init__(keys, allow_missing_keys)
        self.minv, self.maxv = minv, maxv

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            x = data[key].to(torch.float32)
            mn, mx = x.min(), x.max()
            span = mx - mn
            scaled = (x - mn) / torch.where(span > 0, span, torch.ones_like(span)) * (self.maxv - self.minv) + self.minv
            data[key] = torch.where(span > 0, scaled, x * self.minv)      # constant image: MONAI returns arr * minv
        return data


class EnsureChannelFirstd(MapTransform):
    def __init__(self, keys, channel_dim=None, strict_check: bool = True, allow_missing_keys: bool = False, **unused) -> None:
        super().__init__(keys, allow_missing_keys)
        self.channel_dim = channel_dim

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            x = data[key]
            if self.channel_dim == "no_channel" or x.dim() == 2:
                x = x.unsqueeze(0)
            elif isinstance(self.channel_dim, int):
                x = x.movedim(self.channel_dim, 0)
            data[key] = x
        return data


class Resized(MapTransform):
    def __init__(self, keys, spatial_size, mode="area", allow_missing_keys: bool = False, align_corners=None, **unused) -> None:
        super().__init__(keys, allow_missing_keys)
        self.size, self.mode, self.align_corners = [int(v) for v in spatial_size], mode, align_corners

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            x = data[key]
            if list(x.shape[1:]) == self.size:
                continue
            kw = {"align_corners": self.align_corners} if self.mode in ("bilinear", "bicubic") else {}
            data[key] = torch.nn.functional.interpolate(x.float().unsqueeze(0), size=self.size, mode=self.mode, **kw).squeeze(0)
        return data


def _spatial_dims(axes):
    axes = [axes] if isinstance(axes, int) else list(axes)
    return [a + 1 for a in axes]                     # channel-first tensors: spatial axis a is tensor dim a + 1


class Flipd(MapTransform):
    def __init__(self, keys, spatial_axis=None, allow_missing_keys: bool = False, **unused) -> None:
        super().__init__(keys, allow_missing_keys)
        self.spatial_axis = spatial_axis

    def flip(self, x):
        dims = list(range(1, x.dim())) if self.spatial_axis is None else _spatial_dims(self.spatial_axis)
        return torch.flip(x, dims)

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            data[key] = self.flip(data[key])
        return data


class RandFlipd(Flipd, Randomizable):
    def __init__(self, keys, prob=0.1, spatial_axis=None, allow_missing_keys: bool = False, **unused) -> None:
        Flipd.__init__(self, keys, spatial_axis, allow_missing_keys)
        self.prob = prob

    def __call__(self, data):
        data = dict(data)
        do = self.R.rand() < self.prob
        for key in self.present(data):
            if do:
                data[key] = self.flip(data[key])
        return data


class Rotate90d(MapTransform):
    def __init__(self, keys, k=1, spatial_axes=(0, 1), allow_missing_keys: bool = False, **unused) -> None:
        super().__init__(keys, allow_missing_keys)
        self.k, self.spatial_axes = k, tuple(spatial_axes)

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            data[key] = torch.rot90(data[key], self.k, _spatial_dims(self.spatial_axes))
        return data


class RandRotate90d(MapTransform, Randomizable):
    def __init__(self, keys, prob=0.1, max_k=3, spatial_axes=(0, 1), allow_missing_keys: bool = False, **unused) -> None:
        MapTransform.__init__(self, keys, allow_missing_keys)
        self.prob, self.max_k, self.spatial_axes = prob, max_k, tuple(spatial_axes)

    def __call__(self, data):
        data = dict(data)
        k = self.R.randint(self.max_k) + 1
        do = self.R.rand() < self.prob
        for key in self.present(data):
            if do:
                data[key] = torch.rot90(data[key], k, _spatial_dims(self.spatial_axes))
        return data


def rotate2d(x, angle, mode="bilinear", padding_mode="zeros", align_corners=False):
    """Rotation of a channel-first 2-D image about its centre, output size = input size (MONAI Rotate with keep_size)."""
    c, s = float(np.cos(angle)), float(np.sin(angle))
    H, W = x.shape[-2:]
    # normalised coordinates are anisotropic on non-square images: a rotation in pixel space is [[c, -s W/H... ]] there
    theta = torch.tensor([[[c, -s * H / W, 0.0], [s * W / H, c, 0.0]]], dtype=torch.float32, device=x.device)
    grid = torch.nn.functional.affine_grid(theta, (1, x.shape[0], H, W), align_corners=align_corners)
    return torch.nn.functional.grid_sample(x.float().unsqueeze(0), grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners).squeeze(0)


class RandRotated(MapTransform, Randomizable):
    def __init__(self, keys, range_x=0.0, range_y=0.0, range_z=0.0, prob=0.1, keep_size=True, mode="bilinear", padding_mode="border",
                 align_corners=False, allow_missing_keys: bool = False, **unused) -> None:
        MapTransform.__init__(self, keys, allow_missing_keys)
        rng = lambda r: (-abs(r), abs(r)) if np.isscalar(r) else tuple(sorted(r))
        self.range_x, self.range_y, self.range_z = rng(range_x), rng(range_y), rng(range_z)
        self.prob, self.mode, self.padding_mode, self.align_corners = prob, mode, padding_mode, align_corners
        if not keep_size:
            raise NotImplementedError("RandRotated(keep_size=False) is not used by the OCTA configs")

    def __call__(self, data):
        data = dict(data)
        do = self.R.rand() < self.prob
        if do:
            x = self.R.uniform(low=self.range_x[0], high=self.range_x[1])
            self.R.uniform(low=self.range_y[0], high=self.range_y[1])
            self.R.uniform(low=self.range_z[0], high=self.range_z[1])
        for key in self.present(data):
            if do:
                data[key] = rotate2d(data[key], x, self.mode, self.padding_mode, self.align_corners)
        return data


class AsDiscreted(MapTransform):
    def __init__(self, keys, threshold=None, argmax=False, allow_missing_keys: bool = False, **unused) -> None:
        super().__init__(keys, allow_missing_keys)
        self.threshold, self.argmax = threshold, argmax

    def __call__(self, data):
        data = dict(data)
        for key in self.present(data):
            x = data[key]
            if self.argmax:
                x = x.argmax(dim=0, keepdim=True).to(torch.float32)
            if self.threshold is not None:
                x = (x >= self.threshold).to(x.dtype if x.is_floating_point() else torch.float32)
            data[key] = x
        return data


class CastToTyped(MapTransform):
    def __init__(self, keys, dtype=torch.float32, allow_missing_keys: bool = False) -> None:
        super().__init__(keys, allow_missing_keys)
        self.dtype = dtype

    def __call__(self, data):
        data = dict(data)
        dts = self.dtype if isinstance(self.dtype, (list, tuple)) else [self.dtype] * len(self.keys)
        for key, dt in zip(self.keys, dts):
            if key in data:
                data[key] = data[key].to(dt)
            elif not self.allow_missing_keys:
                raise KeyError(f"CastToTyped: key {key!r} is missing")
        return data


# ---- post-processing (array transforms applied to one decollated sample, configs `post_processing`) ------------------

class Activations:
    def __init__(self, sigmoid=False, softmax=False, **unused):
        self.sigmoid, self.softmax = sigmoid, softmax

    def __call__(self, x):
        if self.sigmoid:
            x = torch.sigmoid(x.float())
        if self.softmax:
            x = torch.softmax(x.float(), dim=0)
        return x


class AsDiscrete:
    def __init__(self, threshold=None, argmax=False, **unused):
        self.threshold, self.argmax = threshold, argmax

    def __call__(self, x):
        if self.argmax:
            x = x.argmax(dim=0, keepdim=True).to(torch.float32)
        if self.threshold is not None:
            x = (x >= self.threshold).to(torch.float32)
        return x


class RemoveSmallObjects:
    """Connected components smaller than min_size removed (MONAI -> skimage.morphology.remove_small_objects; connectivity 1
    = 4-neighbourhood). On the GPU through the union-find kernel of csrc/postproc.hip, bit-exact with scipy.ndimage.label +
    bincount (tests/test_postproc_gpu.py)."""

    def __init__(self, min_size=64, connectivity=1, **unused):
        self.min_size, self.connectivity = int(min_size), int(connectivity)

    def __call__(self, x):
        from ..models.postprocess import remove_small_objects_device
        if not x.is_cuda:
            raise RuntimeError("RemoveSmallObjects runs on the GPU (no CPU fallback)")
        keep = remove_small_objects_device((x != 0).to(torch.uint8), self.min_size, self.connectivity)
        return x * keep.to(x.dtype)


class CastToType:
    def __init__(self, dtype=torch.float32):
        self.dtype = dtype

    def __call__(self, x):
        return x.to(self.dtype)


TRANSFORMS = {c.__name__: c for c in (
    LoadGraphAndFilterByRandomRadiusd, ToGrayScaled, SpeckleBrightnesd, AddRandomBackgroundNoised, ImageToImageTranslationd,
    LoadImaged, ScaleIntensityd, EnsureChannelFirstd, Resized, Flipd, RandFlipd, Rotate90d, RandRotate90d, RandRotated, AsDiscreted,
    CastToTyped, Activations, AsDiscrete, RemoveSmallObjects, CastToType)}


def get_data_augmentations(aug_config, seed=None, dtype=torch.float32):
    """YAML list -> transform objects (reference data_transforms.py:587-611): `name` picks the class, the other entries
    are its keyword arguments; `dtype: dtype` placeholders of CastToType* become `dtype` (bf16 under AMP training -- the
    reference's fp16, image_dataset.py:46, is its CUDA autocast type); random transforms are seeded with `seed`.
    Names outside the hot path raise."""
    if aug_config is None:
        return []
    augs = []
    for aug_d in aug_config:
        aug_d = dict(aug_d)
        name = aug_d.pop("name")
        if name not in TRANSFORMS:
            raise NotImplementedError(f"transform {name} is outside the MI355X hot path (MONAI is not a dependency)")
        if name.startswith("CastToType"):
            islist = isinstance(aug_d["dtype"], list)
            types = [dtype if t == "dtype" else (getattr(torch, t) if isinstance(t, str) else t) for t in (aug_d["dtype"] if islist else [aug_d["dtype"]])]
            aug_d["dtype"] = types if islist else types[0]
        obj = TRANSFORMS[name](**aug_d)
        if isinstance(obj, Randomizable):
            obj.set_random_state(seed=seed)
        augs.append(obj)
    return augs
