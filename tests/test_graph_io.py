"""CPU: CSV text format helpers and the arithmetic emulation of numpy's position printing."""
import numpy as np
import pytest
import yaml

from octa_autosegmentation_amd import graph_io


def _parse_text_rows(pos):
    out = np.zeros_like(pos)
    for i, row in enumerate(pos):
        out[i] = graph_io.parse_legacy_position(str(np.array(row)))
    return out


def test_read_back_emulation_matches_numpy_printing():
    rng = np.random.default_rng(0)
    n = 6000
    pos = rng.uniform(-0.01, 1.02, (n, 3))
    pos[:, 2] = rng.uniform(-2e-3, 0.0131, n)          # thin z: many rows switch to scientific notation
    pos[::7, 2] = rng.uniform(-9e-5, 9e-5, n)[::7]
    pos[::11, 0] = 0.0                                 # wall roots
    pos[::13, 1] = 1 - 1e-6
    pos[5] = [0.5, 0.25, 0.125]
    pos[6] = [1e-5, 2e-6, 3e-9]
    want = _parse_text_rows(pos)
    got = graph_io.positions_as_read_back(pos)
    assert (want == got).all(), np.nonzero((want != got).any(axis=1))[0][:10]


def test_read_back_on_simulated_graph():
    import os
    from oracle import sim_oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"] = 30
    cfg["Greenhouse"]["modes"][1]["I"] = 20
    edges, _ = sim_oracle.simulate(cfg, 0)
    text = graph_io.edges_to_csv_text(edges)
    assert text.encode() == g["run_s0_30_20_csv"].tobytes()
    import csv, io
    rows = list(csv.DictReader(io.StringIO(text)))
    parsed = np.array([graph_io.parse_legacy_position(r["node1"]) + graph_io.parse_legacy_position(r["node2"]) + [float(r["radius"])] for r in rows])
    assert (graph_io.edges_as_read_back(edges) == parsed).all()


def test_batched_bifurcation_service_is_bitwise_the_reference_formula():
    """The batched host service (inlined np.cov, stacked eig, cached angles) against the plain
    per-request formula that mirrors greenhouse.py:205-233 line by line."""
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse as gh
    rng = np.random.default_rng(42)
    m = 300
    recs = np.zeros((m, gh._REC_DOUBLES))
    counts = np.zeros(m, np.int32)
    for i in range(m):
        n = int(rng.integers(2, 60))
        counts[i] = n
        pos = rng.uniform(0.1, 0.9, 3) * np.array([1, 1, 0.0131])
        spread = rng.uniform(0.002, 0.08)
        atts = pos + rng.normal(0, spread, (n, 3)) * np.array([1, 1, 0.05]) + rng.normal(0, 0.02, 3) * np.array([1, 1, 0])
        recs[i, 1:4] = pos
        recs[i, 4:7] = [0.0025 / 3, [2.55, 2.9][i % 2], rng.uniform(0.012, 0.034)]
        recs[i, 7:7 + 3 * n] = atts.ravel()
    fast = gh.bifurcation_children_batch(recs, counts)
    for i in range(m):
        n = counts[i]
        p1, p2 = gh.bifurcation_children(recs[i, 1:4].copy(), recs[i, 7:7 + 3 * n].reshape(n, 3).copy(),
                                         float(recs[i, 4]), float(recs[i, 5]), float(recs[i, 6]))
        assert (fast[i, 0:3] == p1).all() and (fast[i, 3:6] == p2).all(), i


def _shard_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from octa_autosegmentation_amd.utils import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = [sharding.rank_seeds(rank, step, 128) for step in range(3)]
    t = sharding.max_over_ranks(1.0 + rank, dist)          # what bench.py does with its wall time
    dist.barrier()
    np.save(out + f".{rank}.npy", np.concatenate(seeds))
    if rank == 0:
        np.save(out + ".t.npy", np.array([t]))
    dist.destroy_process_group()


def test_two_rank_sample_sharding_is_disjoint_and_time_is_max(tmp_path):
    """bench.py's N > 1 path without a GPU: two gloo ranks take disjoint seeds (no data-path collective) and agree on
    the MAX of their step times."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "shard")
    mp.spawn(_shard_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert len(set(a.tolist())) == len(a) == 384 and len(set(b.tolist())) == 384
    assert not set(a.tolist()) & set(b.tolist())
    assert float(np.load(out + ".t.npy")[0]) == 2.0
    # ranks own disjoint 2^26-seed ranges: rank 0 at step 150 no longer meets rank 1 at step 50 (round 2: 100000 * rank + 1000 * step),
    # and leaving the range fails instead of wrapping
    from octa_autosegmentation_amd.utils import sharding
    far = set(sharding.rank_seeds(0, 150, 128).tolist())
    assert not far & set(sharding.rank_seeds(1, 50, 128).tolist()) and not far & set(sharding.rank_seeds(1, 0, 128).tolist())
    assert sharding.rank_seeds(7, 500000, 128).max() < sharding.rank_seeds(8, 0, 128).min()
    with pytest.raises(OverflowError):
        sharding.rank_seeds(0, 2 ** 26 // 128, 128)
