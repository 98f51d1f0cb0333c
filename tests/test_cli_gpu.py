"""GPU: the drop-in CLIs produce the reference's files with the reference's contents."""
import glob
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generate_and_visualize_cli(tmp_path, hip_lib_built):
    import sys
    sys.path.insert(0, ROOT)
    import generate_vessel_graph
    import visualize_vessel_graphs
    from PIL import Image
    from octa_autosegmentation_amd import graph_io
    from oracle import octa_oracle, sim_oracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    out_dir = tmp_path / "graphs"
    generate_vessel_graph.main(["--config_file", str(cfg_path), "--num_samples", "2", "--seed", "0",
                                "--output.directory", str(out_dir), "--Greenhouse.modes", yaml.safe_dump(
                                    [dict(cfg["Greenhouse"]["modes"][0], I=30), dict(cfg["Greenhouse"]["modes"][1], I=20)], default_flow_style=True).strip()])
    dirs = sorted(glob.glob(str(out_dir / "*")))
    assert len(dirs) == 2
    texts = set()
    for d in dirs:
        name = os.path.basename(d)
        assert os.path.exists(os.path.join(d, "config.yml")) and os.path.exists(os.path.join(d, "art_ven_img_gray.png"))
        texts.add(open(os.path.join(d, name + ".csv"), newline="").read())
    assert {g["run_s0_30_20_csv"].tobytes().decode(), g["run_s1_30_20_csv"].tobytes().decode()} == texts
    # visualize: label PNG == oracle raster of the read-back CSV + Floyd-Steinberg
    vis = tmp_path / "vis"
    visualize_vessel_graphs.main(["--source_dir", str(out_dir), "--out_dir", str(vis), "--resolution", "1216,1216,16", "--binarize"])
    for d in dirs:
        name = os.path.basename(d)
        e = graph_io.read_csv(os.path.join(d, name + ".csv"))
        want = octa_oracle.fs_dither(octa_oracle.rasterize(e, [1216, 1216]))
        got = np.array(Image.open(str(vis / (name + "_label.png"))).convert("L"))
        assert Image.open(str(vis / (name + "_label.png"))).mode == "1"
        assert (got == want).all()


def test_generate_cli_keeps_batches_in_flight(tmp_path, hip_lib_built):
    """--inflight: ten samples in batches of four (4 + 4 + 2) from three generator threads, labels included: every sample's files
    arrive, and the CSV texts are the oracle's for seeds 0..9 whatever the order the batches finish in."""
    import sys
    sys.path.insert(0, ROOT)
    import generate_vessel_graph
    from PIL import Image
    from octa_autosegmentation_amd import graph_io
    from oracle import octa_oracle, sim_oracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = 10, 5
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    out_dir = tmp_path / "graphs"
    generate_vessel_graph.main(["--config_file", str(cfg_path), "--num_samples", "10", "--batch", "4", "--inflight", "3", "--seed", "0", "--labels",
                                "--output.directory", str(out_dir), "--output.save_3D_volumes", "nifti"])
    dirs = sorted(glob.glob(str(out_dir / "*")))
    assert len(dirs) == 10
    want = {}
    for seed in range(10):
        e, _ = sim_oracle.simulate(cfg, seed)
        want[sim_oracle.edges_to_csv_text(e)] = e
    for d in dirs:
        name = os.path.basename(d)
        text = open(os.path.join(d, name + ".csv"), newline="").read()
        assert text in want
        e = want.pop(text)
        label = np.array(Image.open(os.path.join(d, name + "_label.png")).convert("L"))
        assert (label == octa_oracle.fs_dither(octa_oracle.rasterize(graph_io.edges_as_read_back(e), [1216, 1216]))).all()
        import gzip
        import struct
        raw = gzip.open(os.path.join(d, "art_ven_img_gray.nii.gz"), "rb").read()          # save_3D_volumes: nifti (generate_vessel_graph.py:75-77)
        dims = struct.unpack_from("<8h", raw, 40)
        # 304 x 304 x (3 + the voxeliser's padding of the thin axis, tree2img.py: the same volume `npy` output stores)
        assert dims[0] == 3 and dims[1:3] == (304, 304) and dims[3] >= 3 and len(raw) == 352 + dims[1] * dims[2] * dims[3]
    assert not want


def test_generate_cli_device_groups_and_rank_plans_give_the_same_files(tmp_path, hip_lib_built, monkeypatch):
    """--devices (round 4): two generator groups (the one GPU of the box listed twice) taking launches from one queue, and the two halves
    of a 2-rank torchrun job (RANK 0 and RANK 1 run one after the other), write the same CSV texts as a single-device run: sample k is
    seeded by --seed + k wherever it runs (the reference fans samples out over a process pool, generate_vessel_graph.py:112-129)."""
    import hashlib
    import sys
    sys.path.insert(0, ROOT)
    import generate_vessel_graph
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = 10, 5
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))

    def run(out, extra, env=()):
        for k, v in env:
            monkeypatch.setenv(k, v)
        generate_vessel_graph.main(["--config_file", str(cfg_path), "--num_samples", "11", "--batch", "3", "--seed", "40", "--output.directory", str(out)] + extra)
        for k, _ in env:
            monkeypatch.delenv(k)
        texts = []
        for d in glob.glob(str(out / "*")):
            name = os.path.basename(d)
            texts.append(hashlib.sha256(open(os.path.join(d, name + ".csv"), "rb").read()).hexdigest())
        return sorted(texts)

    one = run(tmp_path / "one", ["--device", "0"])
    assert len(one) == 11 and len(set(one)) == 11
    assert run(tmp_path / "two", ["--devices", "0,0", "--inflight", "2"]) == one
    halves = run(tmp_path / "r0", [], env=(("RANK", "0"), ("WORLD_SIZE", "2"), ("LOCAL_RANK", "0"), ("LOCAL_WORLD_SIZE", "2"), ("OCTA_NO_AFFINITY", "1")))
    halves += run(tmp_path / "r1", [], env=(("RANK", "1"), ("WORLD_SIZE", "2"), ("LOCAL_RANK", "0"), ("LOCAL_WORLD_SIZE", "2"), ("OCTA_NO_AFFINITY", "1")))
    assert sorted(halves) == one
    with pytest.raises(ValueError):
        generate_vessel_graph.main(["--config_file", str(cfg_path), "--num_samples", "1", "--devices", "0-9", "--output.directory", str(tmp_path / "bad")])


def test_side_stream_work_has_its_own_context(hip_lib_built):
    """utils/aside.py: the scored sample's post-transform (component labelling) runs on a side stream with a context of its own -- the
    rasteriser of the next step may use the calling thread's context on the main stream at the same time (train.py --num_workers 0)."""
    import torch
    from octa_autosegmentation_amd import _native
    from octa_autosegmentation_amd.utils import aside
    mine = _native.ctx().value
    with aside.aside(torch.device("cuda", 0)):
        inside = _native.ctx().value
        assert torch.cuda.current_stream() != torch.cuda.default_stream()
    assert inside != mine and _native.ctx().value == mine
