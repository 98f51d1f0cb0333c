"""Drop-in for the reference's generate_vessel_graph.py on MI355X.

Same flags (`--config_file`, `--num_samples`, `--debug`, `--threads`, dotted `--A.B.c value` overrides) and the
same outputs per sample under `<output.directory>/<YYYYmmdd_HHMMSS>_<uuid4>/`: `config.yml`, `<name>.csv`
(`node1,node2,radius`), `art_ven_img_gray.png` (+ `art_ven_img_gray.npy` for save_3D_volumes: npy), written
from GPU results: all samples are simulated in lock-step batches by the HIP simulator and rasterised by the HIP
rasteriser. Additive flags: `--seed S` (sample k uses random.seed(S+k); np.random.seed(S+k); the reference never
seeds), `--batch B` (samples per GPU batch, default 128), `--device N`.
"""
import argparse
import os
import random
import warnings
from datetime import datetime
from uuid import uuid4

import numpy as np
import yaml


def main(argv=None):
    parser = argparse.ArgumentParser(description='')
    parser.add_argument('--config_file', type=str, required=True)
    parser.add_argument('--num_samples', type=int, default=1)
    parser.add_argument('--debug', action="store_true")
    parser.add_argument('--threads', type=int, default=-1, help="accepted for compatibility; samples run on the GPU")
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--batch', type=int, default=128)
    parser.add_argument('--device', type=int, default=0)
    args, unknown = parser.parse_known_args(argv)
    if args.debug:
        warnings.filterwarnings('error')
    assert os.path.isfile(args.config_file), f"Error: Your provided config path {args.config_file} does not exist!"
    with open(os.path.abspath(args.config_file), "r") as f:
        config = yaml.safe_load(f)

    from octa_autosegmentation_amd import graph_io, pipeline
    from octa_autosegmentation_amd.utils.config_overrides import apply_cli_overrides_from_unknown_args
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    apply_cli_overrides_from_unknown_args(config, unknown)
    out_cfg = config['output']
    assert out_cfg.get('save_3D_volumes') in [None, 'npy', 'nifti'], \
        f"Your provided option {out_cfg.get('save_3D_volumes')} for 'save_3D_volumes' does not exist. Choose one of 'null', 'npy' or 'nifti'."
    if out_cfg.get('save_3D_volumes') == 'nifti':
        raise NotImplementedError("nifti output needs nibabel, which is not part of the MI355X image; use 'npy'")

    import torch
    torch.cuda.set_device(args.device)
    seed0 = args.seed if args.seed is not None else random.SystemRandom().randrange(0, 2 ** 31 - args.num_samples - 1)
    done = 0
    while done < args.num_samples:
        B = min(args.batch, args.num_samples - done)
        gen = pipeline.TripleGenerator(config, B)
        seeds = np.arange(seed0 + done, seed0 + done + B, dtype=np.int64).astype(np.uint32)
        out = gen.generate(seeds, want_label=False)
        res = out["result"]
        images = out["image"].cpu().numpy()
        for k in range(B):
            out_dir = os.path.join(os.path.abspath(out_cfg['directory']), datetime.now().strftime('%Y%m%d_%H%M%S') + "_" + str(uuid4()))
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, 'config.yml'), 'w') as f:
                yaml.dump(config, f)
            edges = res.sample_edges(k)
            if out_cfg.get('save_trees', True):
                graph_io.write_csv(edges, os.path.join(out_dir, os.path.basename(out_dir) + '.csv'))
            if out_cfg.get("save_3D_volumes"):
                shape = np.array([config['Greenhouse']['SimulationSpace'][a] for a in ("no_voxel_x", "no_voxel_y", "no_voxel_z")])
                vol_dim = [int(d) for d in shape * out_cfg['image_scale_factor']]
                d_edges = torch.from_numpy(np.ascontiguousarray(edges)).cuda()
                na = int(res.n_art[k])
                vols = tree2img.voxelize_edges_device(d_edges, np.array([0, na, len(edges)]), vol_dim)
                vol = torch.maximum(vols[0], vols[1]).cpu().numpy().astype(np.uint8)
                np.save(f'{out_dir}/art_ven_img_gray.npy', vol)
            if out_cfg.get("save_2D_image", True):
                tree2img.save_2d_img(images[k], out_dir, "art_ven_img_gray")
        gen.close()
        done += B
        print(f"generated {done}/{args.num_samples} vessel graphs")


if __name__ == '__main__':
    main()
