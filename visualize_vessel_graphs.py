"""Drop-in for the reference's visualize_vessel_graphs.py on MI355X: CSV graphs -> image / binarised label PNG /
3-D volume with the reference's flags (`--source_dir --out_dir --resolution --save_2d/--no_save_2d --save_3d
--save_3d_as --mip_axis --binarize --num_samples --max_dropout_prob --ignore_z --threads`). Rendering runs in
the HIP rasteriser / voxeliser; label PNGs are the bit-exact Floyd-Steinberg binarisation (mode "1")."""
import argparse
import os
import pickle
import re
from glob import glob

import numpy as np


def _natural_key(s):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', s)]


def main(argv=None):
    parser = argparse.ArgumentParser(description='')
    parser.add_argument('--source_dir', type=str, required=True)
    parser.add_argument('--out_dir', type=str, required=True)
    parser.add_argument('--resolution', type=str, default='1216,1216,16')
    parser.add_argument('--save_2d', action='store_true')
    parser.add_argument('--no_save_2d', action="store_false", dest="save_2d")
    parser.add_argument('--save_3d', action="store_true")
    parser.add_argument('--save_3d_as', choices=[".nii.gz", ".npy"], default=".nii.gz")
    parser.add_argument('--mip_axis', type=int, default=2)
    parser.add_argument('--binarize', action="store_true")
    parser.add_argument('--num_samples', type=int, default=9999999)
    parser.add_argument('--max_dropout_prob', type=float, default=0)
    parser.add_argument('--ignore_z', action="store_true", default=False)
    parser.add_argument('--threads', type=int, default=-1)
    parser.set_defaults(save_2d=True)
    args = parser.parse_args(argv)
    resolution = np.array([int(d) for d in args.resolution.split(',')])
    assert not args.save_3d or len(resolution) == 3, "If you want to generate the 3d volume, you need to specify the resolution of all three dimensions."
    assert os.path.isdir(args.source_dir), f"The provided source directory {args.source_dir} does not exist."
    assert args.mip_axis in [0, 1, 2], "The axis must be '0' (x), '1' (y) or '2' (z)."
    assert args.save_3d or args.save_2d, "You must either activate saving the 2D image or the 3D volume."
    if args.save_3d and args.save_3d_as == ".nii.gz":
        raise NotImplementedError("nifti output needs nibabel, which is not part of the MI355X image; use --save_3d_as .npy")
    os.makedirs(args.out_dir, exist_ok=True)
    img_res = [int(r) for i, r in enumerate(resolution) if not (len(resolution) == 3 and i == args.mip_axis)]

    import csv
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from octa_autosegmentation_amd import graph_io
    from octa_autosegmentation_amd.output_files import SampleFileWriter, default_threads
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    files = sorted(glob(os.path.join(args.source_dir, "**", "*.csv"), recursive=True), key=_natural_key)[:args.num_samples]
    assert len(files) > 0, f"Your provided source directory {args.source_dir} does not contain any csv files."
    n_threads = args.threads if args.threads > 0 else default_threads()
    if args.save_2d and not args.save_3d and args.max_dropout_prob == 0:
        # the common case (labels / images without dropout): whole chunks of graphs go through the native CSV reader (host
        # threads), ONE rasteriser launch sequence and the native PNG encoder. The reference draws one `random()` per edge even
        # at dropout 0 (tree2img.py:62,78); nothing else reads that stream in this script, so the draws are not replayed.
        writer = SampleFileWriter(n_threads)
        chunk = 32 if max(img_res) > 512 else 128
        with ThreadPoolExecutor(n_threads) as readers:
            for c0 in range(0, len(files), chunk):
                part = files[c0:c0 + chunk]
                edges = list(readers.map(graph_io.read_csv_native, part))
                off = np.concatenate(([0], np.cumsum([len(e) for e in edges]))).astype(np.int64)
                d_edges = torch.from_numpy(np.concatenate(edges, axis=0)).cuda()
                img = tree2img.rasterize_edges_device(d_edges, off, img_res, args.mip_axis, min_radius=0.0, max_radius=1.0)
                out = (tree2img.binarize_label_device(img) if args.binarize else img).cpu().numpy()
                writer.wait()
                for path, a in zip(part, out):
                    name = os.path.basename(path)[:-4]
                    if args.binarize:
                        writer.pending.append(writer.pool.submit(tree2img.save_label_png, a, os.path.join(args.out_dir, name + "_label.png")))
                    else:
                        writer.pending.append(writer.pool.submit(tree2img.save_2d_img, a, args.out_dir, name))
        writer.close()
        print(f"rendered {len(files)} vessel graphs")
        return
    for path in files:
        name = os.path.basename(path)[:-4]
        with open(path, newline='') as fh:
            forest = list(csv.DictReader(fh))
        if args.save_3d:
            vol, black_dict = tree2img.voxelize_forest(forest, [int(r) for r in resolution], max_dropout_prob=args.max_dropout_prob, ignore_z=args.ignore_z)
            out_name = name + ("_3d_label" if args.binarize else "_3d")
            if args.binarize:
                vol = vol >= 1
            np.save(os.path.join(args.out_dir, out_name + ".npy"), vol.astype(np.bool_))
            if args.max_dropout_prob > 0:
                with open(os.path.join(args.out_dir, out_name + "_blackdict.pkl"), 'wb') as f:
                    pickle.dump(black_dict, f)
        if args.save_2d:
            img, black_dict = tree2img.rasterize_forest(forest, img_res, args.mip_axis, max_dropout_prob=args.max_dropout_prob)
            if args.binarize:
                d = torch.from_numpy(img.astype(np.uint8)[None]).cuda()
                bits = tree2img.binarize_label_device(d)[0].cpu().numpy()
                tree2img.save_label_png(bits, os.path.join(args.out_dir, name + "_label.png"))
            else:
                tree2img.save_2d_img(img, args.out_dir, name)
            if args.max_dropout_prob > 0:
                with open(os.path.join(args.out_dir, name + "_blackdict.pkl"), 'wb') as f:
                    pickle.dump(black_dict, f)
    print(f"rendered {len(files)} vessel graphs")


if __name__ == "__main__":
    main()
