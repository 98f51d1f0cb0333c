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
        self._cfg_seen = None         # the last configuration dumped (a deep copy) and its YAML text: a run writes the same config.yml
        self._cfg_text = None         # next to every sample, and yaml.dump of it costs 1.2 ms of the submitting thread per sample

    def submit(self, out_dir, name, edges=None, image=None, label_bits=None, config=None, volume=None, volume_format="npy"):
        """Queue one sample's files. edges float64 [n,7]; image uint8 [H,W]; label_bits uint8 [H,W] (non-zero = white)."""
        os.makedirs(out_dir, exist_ok=True)
        if config is not None:
            if self._cfg_text is None or config != self._cfg_seen:
                import copy
                import yaml
                self._cfg_seen, self._cfg_text = copy.deepcopy(config), yaml.dump(config)
            self.pending.append(self.pool.submit(_write_text, os.path.join(out_dir, "config.yml"), self._cfg_text))
        if edges is not None:
            self.pending.append(self.pool.submit(graph_io.write_csv, edges, os.path.join(out_dir, name + ".csv")))
        if image is not None:
            self.pending.append(self.pool.submit(tree2img.save_2d_img, image, out_dir, "art_ven_img_gray"))
        if label_bits is not None:
            self.pending.append(self.pool.submit(tree2img.save_label_png, label_bits, os.path.join(out_dir, name + "_label.png")))
        if volume is not None:
            if volume_format == "nifti":
                self.pending.append(self.pool.submit(write_nifti_u8, os.path.join(out_dir, "art_ven_img_gray.nii.gz"), volume))
            else:
                self.pending.append(self.pool.submit(np.save, os.path.join(out_dir, "art_ven_img_gray.npy"), volume))

    def wait(self):
        """Block until everything queued so far is on disk; a failed write raises here."""
        pending, self.pending = self.pending, []
        for f in pending:
            f.result()

    def close(self):
        self.wait()
        self.pool.shutdown()


def _write_text(path, text):
    with open(path, "w") as f:
        f.write(text)


def write_nifti_u8(path, volume):
    """`nib.save(nib.Nifti1Image(volume, np.eye(4)), path)` (generate_vessel_graph.py:75-77) for a uint8 volume without nibabel (not in
    the MI355X image): a single-file NIfTI-1 (`n+1`), 348-byte header + 4 empty extension bytes, voxels in Fortran order, gzip when
    the name ends in .gz. Identity sform (code 2, "aligned"), qform code 0, unit voxel sizes, no intensity scaling -- what nibabel
    derives from an identity affine. Written from the NIfTI-1 specification; nibabel is absent, so byte identity with its files is
    unpinned (readers compare header fields, not bytes)."""
    import gzip
    import struct
    vol = np.asarray(volume)
    if vol.dtype != np.uint8 or vol.ndim != 3:
        raise ValueError("write_nifti_u8 expects a 3-D uint8 volume")
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)                                   # sizeof_hdr
    hdr[38] = ord("r")                                                    # regular
    struct.pack_into("<8h", hdr, 40, 3, vol.shape[0], vol.shape[1], vol.shape[2], 1, 1, 1, 1)   # dim
    struct.pack_into("<hh", hdr, 70, 2, 8)                                # datatype = DT_UINT8, bitpix
    struct.pack_into("<8f", hdr, 76, 1, 1, 1, 1, 1, 1, 1, 1)              # pixdim (qfac = 1)
    struct.pack_into("<f", hdr, 108, 352.0)                               # vox_offset
    struct.pack_into("<ff", hdr, 112, float("nan"), float("nan"))         # scl_slope / scl_inter: not set
    struct.pack_into("<hh", hdr, 252, 0, 2)                               # qform_code = unknown, sform_code = aligned
    struct.pack_into("<4f", hdr, 280, 1, 0, 0, 0)                         # srow_x
    struct.pack_into("<4f", hdr, 296, 0, 1, 0, 0)                         # srow_y
    struct.pack_into("<4f", hdr, 312, 0, 0, 1, 0)                         # srow_z
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + b"\0\0\0\0" + vol.tobytes(order="F")
    if str(path).endswith(".gz"):
        with open(path, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)
