"""configs[2] plumbing speed: train.py with the reference's S config on N freshly generated full-length graphs (development aid).
  python tools/time_train_cli.py [N=64] [epochs=3] [num_workers=1]"""
import os, sys, glob, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
E = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = sys.argv[3] if len(sys.argv) > 3 else "1"
import generate_vessel_graph, train as train_cli
from octa_autosegmentation_amd.utils import configs
tmp = tempfile.mkdtemp(prefix="octa_cli_")
t0 = time.time()
generate_vessel_graph.main(["--config_file", configs.GENERATOR_CONFIG, "--num_samples", str(N), "--seed", "1", "--output.directory", os.path.join(tmp, "graphs")])
print(f"generated {N} graphs in {time.time() - t0:.1f} s", flush=True)
csvs = os.path.join(tmp, "graphs", "**", "*.csv")
ov = ["--Train.data.image.files", csvs, "--Train.data.label.files", csvs, "--Train.epochs", str(E), "--Train.epochs_decay", "0", "--General.seed", "3",
      "--Output.save_dir", os.path.join(tmp, "results"), "--num_workers", W]
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_ves_seg-S.yml")))
cfg.pop("Validation"); cfg.pop("Test")
p = os.path.join(tmp, "cfg.yml"); yaml.safe_dump(cfg, open(p, "w"))
train_cli.main(["--config_file", p] + ov)
