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
        self.threads = threads or default_threads()
        self.pool = ThreadPoolExecutor(max_workers=self.threads)
        self.batch_pool = ThreadPoolExecutor(max_workers=2)       # each job is ONE native call that runs its own `threads` threads
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

    def submit_batch(self, out_dirs, names, edges=None, edge_off=None, images=None, label_bits=None, config=None, label_width=None, on_done=None):
        """Queue the files of a whole batch as ONE native call (csrc/fileio.cpp octa_write_sample_files: its own threads take the samples
        from a counter, nothing runs under the interpreter lock). out_dirs / names: per sample; edges float64 [rows, 7] of the whole batch
        with edge_off int64 [B + 1]; images uint8 [B, h, w]; label_bits uint8 [B, H, W] (non-zero = white) -- or, with label_width = W, the
        rows packed on the device, uint8 [B, H, (W + 7) // 8] (tree2img.pack_label_bits_device). on_done() is called when the files are
        written (or the call has failed): the caller's staging buffers are free again. Round 6: submitted sample by sample (four futures
        each) the CLI's on-disk rate was bound by the lock the generator threads need too."""
        import ctypes
        from . import _native
        B = len(out_dirs)
        assert len(names) == B
        cfg = None
        if config is not None:
            if self._cfg_text is None or config != self._cfg_seen:
                import copy
                import yaml
                self._cfg_seen, self._cfg_text = copy.deepcopy(config), yaml.dump(config)
            cfg = self._cfg_text.encode()
        e = off = None
        if edges is not None:
            e = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 7)
            off = np.ascontiguousarray(edge_off, dtype=np.int64)
            assert len(off) == B + 1 and int(off[-1]) <= len(e)
        img = np.ascontiguousarray(images, dtype=np.uint8) if images is not None else None
        lab = np.ascontiguousarray(label_bits, dtype=np.uint8) if label_bits is not None else None
        assert img is None or (img.ndim == 3 and img.shape[0] == B)
        assert lab is None or (lab.ndim == 3 and lab.shape[0] == B)
        packed = label_width is not None
        lab_w = int(label_width) if packed else (int(lab.shape[2]) if lab is not None else 0)
        assert lab is None or not packed or lab.shape[2] == (lab_w + 7) // 8
        dirs_c = (ctypes.c_char_p * B)(*[os.fsencode(d) for d in out_dirs])
        names_c = (ctypes.c_char_p * B)(*[os.fsencode(n) for n in names])
        threads = self.threads

        def job(keep=(e, off, img, lab, dirs_c, names_c, cfg)):        # the arrays stay alive until the call has returned
            p = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
            rc = _native.lib().octa_write_sample_files(
                B, ctypes.cast(dirs_c, ctypes.c_void_p), ctypes.cast(names_c, ctypes.c_void_p), p(e), p(off), p(img), int(img.shape[2]) if img is not None else 0,
                int(img.shape[1]) if img is not None else 0, p(lab), lab_w,
                int(lab.shape[1]) if lab is not None else 0, 1 if packed else 0, cfg, len(cfg) if cfg is not None else 0, -1, int(threads))
            try:
                _native.check(rc, "octa_write_sample_files")
            finally:
                if on_done is not None:
                    on_done()
        self.pending.append(self.batch_pool.submit(job))

    def wait(self):
        """Block until everything queued so far is on disk; a failed write raises here."""
        pending, self.pending = self.pending, []
        for f in pending:
            f.result()

    def close(self):
        self.wait()
        self.pool.shutdown()
        self.batch_pool.shutdown()


class HostStaging:
    """Pinned host buffers a generator thread copies a finished batch into (edge list, images, packed label rows): the writers' inputs.
    Round 6: `tensor.cpu()` allocated and page-faulted ~0.5 GB of pageable memory per 512-sample batch and copied into it through the
    runtime's bounce buffers; a pinned block that is kept is one DMA. A set is in use from `fetch` until the writer's `on_done`."""

    def __init__(self):
        self.buf = {}

    def _get(self, key, shape, dtype):
        import torch
        n = int(np.prod(shape))
        t = self.buf.get(key)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty((max(n, 1) * 9 // 8,), dtype=dtype, pin_memory=True)         # a little headroom: batches differ in their edge counts
            self.buf[key] = t
        return t[:n].view(shape)

    def fetch(self, **tensors):
        """CUDA tensors -> numpy views of pinned copies (None stays None); one stream synchronisation for all of them."""
        import torch
        out = {}
        for k, t in tensors.items():
            if t is None:
                out[k] = None
                continue
            h = self._get(k, tuple(t.shape), t.dtype)
            h.copy_(t, non_blocking=True)
            out[k] = h
        torch.cuda.current_stream().synchronize()
        return {k: (v.numpy() if v is not None else None) for k, v in out.items()}


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
