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
from PIL import Image


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
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    files = sorted(glob(os.path.join(args.source_dir, "**", "*.csv"), recursive=True), key=_natural_key)[:args.num_samples]
    assert len(files) > 0, f"Your provided source directory {args.source_dir} does not contain any csv files."
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
                Image.fromarray(bits > 0).save(os.path.join(args.out_dir, name + "_label.png"))
            else:
                Image.fromarray(img.astype(np.uint8)).save(os.path.join(args.out_dir, name + ".png"))
            if args.max_dropout_prob > 0:
                with open(os.path.join(args.out_dir, name + "_blackdict.pkl"), 'wb') as f:
                    pickle.dump(black_dict, f)
    print(f"rendered {len(files)} vessel graphs")


if __name__ == "__main__":
    main()
