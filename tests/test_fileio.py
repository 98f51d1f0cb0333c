"""CPU: the native file formats (csrc/fileio.cpp) against the libraries the reference writes them with --
numpy's str(ndarray) + CPython's repr(float) through csv.writer (generate_vessel_graph.py:59-66), csv.DictReader +
float() (tree2img.py:73-76), Pillow's PNG codec (tree2img.py:282-292, visualize_vessel_graphs.py:99)."""
import os

import numpy as np
import pytest

from octa_autosegmentation_amd import graph_io
from octa_autosegmentation_amd.vessel_graph_generation import tree2img

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz")


def _edge_cases(n, seed):
    rng = np.random.default_rng(seed)
    e = np.zeros((n, 7))
    e[:, 0:3] = rng.uniform(0, 1, (n, 3)); e[:, 3:6] = rng.uniform(0, 1, (n, 3))
    e[:, 2] = rng.uniform(-2e-3, 0.0131, n)                                            # thin z: many vectors switch to scientific
    e[:, 5] = rng.uniform(0, 0.0131, n) * rng.choice([1, 1e-1, 1e-2, 1e-3], n)
    e[::7, 0] = 0.0                                                                    # wall roots
    e[::11, 4] = 1 - 1e-6
    e[::13, 1] = rng.integers(0, 512, len(e[::13])) / 512.0                            # exact ties at the 8th decimal (k / 512)
    e[::17, 3] = 2.0 ** -rng.integers(1, 40, len(e[::17]))                             # ties in scientific notation too (2^-13 ...)
    e[::19, 0:3] = np.round(rng.uniform(0, 1, (len(e[::19]), 3)), rng.integers(1, 6))  # short decimals: trimming / padding
    e[::23, 0:3] = rng.uniform(0, 1, (len(e[::23]), 3)) * 10.0 ** rng.integers(-12, 9, (len(e[::23]), 1))   # up to 1e8: unique digits end early
    e[::29, 3] = -e[::29, 3]; e[::31, 0:3] = -e[::31, 0:3]                             # signs: left padding
    e[5, 0:3] = [0.5, 0.25, 0.125]; e[6, 3:6] = [1e-5, 2e-6, 3e-9]; e[7, 0:3] = [0.99999999499, 0.123456785, 0.000123456789]
    e[8, 0:3] = 0; e[9, 0:3] = [1, 2, 3]; e[10, 3:6] = 1e-5; e[11, 0:3] = [2 ** -13, 1, 0.5]; e[12, 0:3] = [1e22, 1, 1]
    e[13, 0:3] = [123456789.0, 1, 1]; e[14, 0:3] = [-0.0, 0.5, 0.25]
    e[:, 6] = rng.uniform(1e-4, 3e-2, n) * rng.choice([1, 1, 1, 1e-3, 1e3, 1e20], n)   # repr(): fixed and exponent forms
    e[3, 6] = 1.0; e[4, 6] = 1e16; e[15, 6] = 0.0001; e[16, 6] = 0.00001; e[17, 6] = 123456.0; e[18, 6] = 5e-324; e[19, 6] = 1.7976931348623157e308
    return e


def test_formatter_is_byte_identical_to_numpy_and_cpython(hip_lib_built):
    e = _edge_cases(40000, 0)
    got = graph_io.edges_to_csv_bytes(e)
    want = graph_io.edges_to_csv_text_numpy(e).encode()
    if got != want:
        g, w = got.split(b"\r\n"), want.split(b"\r\n")
        bad = [i for i, (a, b) in enumerate(zip(g, w)) if a != b]
        raise AssertionError(f"{len(bad)} rows differ, first: {g[bad[0]]!r} vs {w[bad[0]]!r}")


def test_formatter_reproduces_the_reference_made_csv_files(hip_lib_built):
    """tests/golden/sim_golden.npz holds CSV files written by the reference itself (tools/make_golden_sim.py): parse them
    natively, format them again natively -- the bytes must come back."""
    g = np.load(GOLDEN)
    n = 0
    for k in g.files:
        if k.endswith("_csv"):
            text = g[k].tobytes()
            edges = graph_io.parse_csv_bytes(text)
            assert graph_io.edges_to_csv_bytes(edges) == text, k
            n += 1
    assert n >= 8


def test_native_reader_equals_csv_module(tmp_path, hip_lib_built):
    e = _edge_cases(5000, 1)
    path = str(tmp_path / "g.csv")
    graph_io.write_csv(e, path)
    assert open(path, "rb").read() == graph_io.edges_to_csv_text_numpy(e).encode()
    a, b = graph_io.read_csv(path), graph_io.read_csv_native(path)
    assert a.shape == b.shape == (5000, 7) and (a == b).all()
    assert (b[:, 6] == e[:, 6]).all()                      # repr(float) round-trips
    # the arithmetic emulation of the text round trip (used to render labels from device data) agrees wherever a coordinate
    # keeps fewer than 16 significant digits at 8 decimals -- everything the simulator can produce (|x| <= 1)
    small = (np.abs(e[:, 0:6]) < 1e6).all(axis=1)
    assert small.sum() > 4500 and (b[small, 0:6] == graph_io.edges_as_read_back(e)[small, 0:6]).all()
    with pytest.raises(Exception):
        graph_io.parse_csv_bytes(b"node1,node2,radius\r\n[0.1 0.2 abc],[0. 0. 0.],0.1\r\n")


def test_png_files_decode_like_pillows(tmp_path, hip_lib_built):
    from PIL import Image
    rng = np.random.default_rng(2)
    for (h, w) in ((304, 304), (37, 53), (1216, 1216)):
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        img[h // 3: h // 2] = 0
        tree2img.save_2d_img(img, str(tmp_path), f"g{h}")
        back = Image.open(str(tmp_path / f"g{h}.png"))
        assert back.mode == "L" and (np.array(back) == img).all()
        bits = (img > 127).astype(np.uint8) * 255
        p = str(tmp_path / f"b{h}.png")
        tree2img.save_label_png(bits, p)
        back = Image.open(p)
        assert back.mode == "1" and (np.array(back.convert("L")) == bits).all()
        ref = str(tmp_path / f"r{h}.png")
        Image.fromarray(bits > 0).save(ref)               # what visualize_vessel_graphs.py:99 writes
        assert (np.array(Image.open(ref).convert("L")) == np.array(back.convert("L"))).all()


def test_nifti_writer_header_and_voxels(tmp_path):
    """generate_vessel_graph.py:75-77 (`save_3D_volumes: nifti`) without nibabel: the file read back by the NIfTI-1 layout -- header
    fields a reader checks, Fortran-ordered voxels behind vox_offset."""
    import gzip
    import struct
    from octa_autosegmentation_amd.output_files import write_nifti_u8
    rng = np.random.default_rng(4)
    vol = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    path = tmp_path / "art_ven_img_gray.nii.gz"
    write_nifti_u8(str(path), vol)
    raw = gzip.open(path, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack_from("<8h", raw, 40) == (3, 7, 5, 3, 1, 1, 1, 1)
    assert struct.unpack_from("<hh", raw, 70) == (2, 8) and struct.unpack_from("<f", raw, 108)[0] == 352.0
    assert struct.unpack_from("<hh", raw, 252) == (0, 2)
    assert struct.unpack_from("<12f", raw, 280) == (1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    assert len(raw) == 352 + vol.size
    back = np.frombuffer(raw, np.uint8, vol.size, 352).reshape(vol.shape, order="F")
    assert (back == vol).all()
    with pytest.raises(ValueError):
        write_nifti_u8(str(tmp_path / "x.nii"), vol.astype(np.float32))


def test_batch_writer_writes_the_files_of_the_per_sample_path(tmp_path, hip_lib_built):
    """octa_write_sample_files (one native call per batch, its own threads: the CLI's path since round 6) against the per-sample
    writers it replaces: every file byte for byte -- config.yml, <name>.csv (ragged row counts, an EMPTY graph), art_ven_img_gray.png,
    <name>_label.png -- and directories created with their parents; a failure names the file."""
    import filecmp
    from octa_autosegmentation_amd import _native
    from octa_autosegmentation_amd.output_files import SampleFileWriter
    rng = np.random.default_rng(5)
    rows = [700, 0, 1, 333, 1200]
    B = len(rows)
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    e = np.zeros((int(off[-1]), 7))
    e[:, 0:3] = rng.uniform(0, 1, (len(e), 3))
    e[:, 3:6] = e[:, 0:3] + rng.normal(0, 0.01, (len(e), 3))
    e[:, 6] = rng.uniform(1e-4, 1e-2, len(e))
    e[5, 0:3] = [1e-9, 0.5, 123.0]                                     # a scientific-notation row
    img = rng.integers(0, 256, (B, 76, 52), dtype=np.uint8)
    lab = ((rng.uniform(0, 1, (B, 67, 131)) > 0.7) * 255).astype(np.uint8)      # a width that is not a multiple of 8
    cfg = {"Greenhouse": {"d": 0.1, "modes": [{"I": 3}]}, "output": {"directory": "x"}}
    w = SampleFileWriter(3)
    try:
        w.submit_batch([str(tmp_path / "new" / "deep" / f"d{k}") for k in range(B)], [f"s{k}" for k in range(B)], edges=e, edge_off=off, images=img, label_bits=lab, config=cfg)
        for k in range(B):
            w.submit(str(tmp_path / "old" / f"d{k}"), f"s{k}", edges=e[off[k]:off[k + 1]], image=img[k], label_bits=lab[k], config=cfg)
        w.wait()
        for k in range(B):
            for f in (f"s{k}.csv", "art_ven_img_gray.png", f"s{k}_label.png", "config.yml"):
                assert filecmp.cmp(tmp_path / "new" / "deep" / f"d{k}" / f, tmp_path / "old" / f"d{k}" / f, shallow=False), (k, f)
        # labels handed over as mode "1" rows (what tree2img.pack_label_bits_device makes on the GPU): the same files, and the caller is told
        done = []
        w.submit_batch([str(tmp_path / "packed" / f"d{k}") for k in range(B)], [f"s{k}" for k in range(B)], label_bits=np.packbits(lab > 0, axis=2),
                       label_width=lab.shape[2], on_done=lambda: done.append(1))
        w.wait()
        assert done == [1]
        for k in range(B):
            assert filecmp.cmp(tmp_path / "packed" / f"d{k}" / f"s{k}_label.png", tmp_path / "old" / f"d{k}" / f"s{k}_label.png", shallow=False)
        # only what was asked for
        w.submit_batch([str(tmp_path / "part" / "a")], ["a"], images=img[:1])
        w.wait()
        assert sorted(os.listdir(tmp_path / "part" / "a")) == ["art_ven_img_gray.png"]
        # a directory that cannot be created fails loudly, with its name
        (tmp_path / "file").write_text("x")
        w.submit_batch([str(tmp_path / "file" / "sub")], ["a"], images=img[:1])
        with pytest.raises(_native.OctaHipError, match="file"):
            w.wait()
    finally:
        w.pending = []
        w.close()
