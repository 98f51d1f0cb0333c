"""train.py (configs/config_ves_seg-S.yml through the device-side loader) against the bare DynUNet-S step: the two legs of bench.py
alone. usage: python tools/time_train_cli.py [epochs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    u = bench.unet_train_bench(dev, 4, None, 1)
    print("bare step", round(u["value"], 1), "imgs/s", round(u["ms_per_step"], 2), "ms", flush=True)
    c = bench.train_cli_leg(n_graphs=int(sys.argv[2]) if len(sys.argv) > 2 else 64, epochs=epochs, extra_args=os.environ.get("OCTA_TRAIN_ARGS", "").split())
    print("train.py", [round(v, 1) for v in c["imgs_per_s_per_epoch"]], flush=True)
    print("ratio", round(c["value"] / u["value"], 3))
