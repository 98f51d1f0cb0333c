"""Writing the generator's per-sample files in parallel: `<dir>/<name>.csv` (graph), `<dir>/art_ven_img_gray.png`
(304x304 image) and optionally `<dir>/<name>_label.png` (1216x1216 binarised label) -- the files the reference produces
with generate_vessel_graph.py:43-86 followed by visualize_vessel_graphs.py:95-101 (docker/dockershell.sh:10-17).

All formatting / encoding is native (csrc/fileio.cpp, called through ctypes with the GIL released), so a pool of host
threads scales with cores: the 128-sample batch of BASELINE configs[1] is 150 MB of CSV text."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import graph_io
from .vessel_graph_generation import tree2img


def default_threads():
    return max(1, min(32, (os.cpu_count() or 2) - 1))


class SampleFileWriter:
    def __init__(self, threads=None):
        self.pool = ThreadPoolExecutor(max_workers=threads or default_threads())
        self.pending = []

    def submit(self, out_dir, name, edges=None, image=None, label_bits=None, config=None, volume=None):
        """Queue one sample's files. edges float64 [n,7]; image uint8 [H,W]; label_bits uint8 [H,W] (non-zero = white)."""
        os.makedirs(out_dir, exist_ok=True)
        if config is not None:
            import yaml
            with open(os.path.join(out_dir, "config.yml"), "w") as f:
                yaml.dump(config, f)
        if edges is not None:
            self.pending.append(self.pool.submit(graph_io.write_csv, edges, os.path.join(out_dir, name + ".csv")))
        if image is not None:
            self.pending.append(self.pool.submit(tree2img.save_2d_img, image, out_dir, "art_ven_img_gray"))
        if label_bits is not None:
            self.pending.append(self.pool.submit(tree2img.save_label_png, label_bits, os.path.join(out_dir, name + "_label.png")))
        if volume is not None:
            self.pending.append(self.pool.submit(np.save, os.path.join(out_dir, "art_ven_img_gray.npy"), volume))

    def wait(self):
        """Block until everything queued so far is on disk; a failed write raises here."""
        pending, self.pending = self.pending, []
        for f in pending:
            f.result()

    def close(self):
        self.wait()
        self.pool.shutdown()
