"""What the 8-GPU run relies on, exercised without a GPU: eight gloo ranks take disjoint seeds and disjoint launch plans, exchange
gradients through the flat arena exactly like one process over the whole batch, agree on the number of ranks that answered the
all-reduce (the bench line's `rccl_ranks`), and get disjoint CPU shares and a sane per-rank host budget (the reference
has no multi-GPU path: models/networks.py:900 is a commented-out DataParallel, generate_vessel_graph.py:112-129 a process pool)."""
import os
import socket

import numpy as np
import pytest
import torch

from octa_autosegmentation_amd.utils import sharding

CFG = {"General": {"amp": False, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                            "kernel_size": [3, 3, 3], "strides": [1, 2, 1], "upsample_kernel_size": [2, 1],
                                            "filters": [4, 8, 8]}},
       "Train": {"lr": 1e-3, "loss": "DiceBCELoss", "epochs": 3, "epochs_decay": 1}}
WORLD = 8


def test_device_spec_and_launch_plans():
    assert sharding.parse_devices("0-7") == list(range(8))
    assert sharding.parse_devices("0,2,5-6") == [0, 2, 5, 6]
    assert sharding.parse_devices("all", 4) == [0, 1, 2, 3]
    assert sharding.parse_devices("0,0") == [0, 0]              # two generator groups on one GPU
    assert sharding.parse_devices(None) is None
    for bad in ("", "3-1", "x"):
        with pytest.raises(ValueError):
            sharding.parse_devices(bad)
    with pytest.raises(ValueError):
        sharding.parse_devices("0-8", 8)
    # the ranks' plans partition the job, whatever its size
    for n, b in ((10000, 512), (4096, 512), (7, 512), (0, 128), (513, 512)):
        plans = [sharding.plan_batches(n, b, r, WORLD) for r in range(WORLD)]
        flat = sorted(p for pl in plans for p in pl)
        assert flat == sharding.plan_batches(n, b)
        assert sum(c for _, c in flat) == n
        covered = np.zeros(n, int)
        for s, c in flat:
            covered[s:s + c] += 1
        assert (covered == 1).all()
        assert max(len(p) for p in plans) - min(len(p) for p in plans) <= 1


def test_host_budget_and_affinity_shares():
    one = sharding.host_budget(1, 2, cores=256)
    assert one["writers"] == 16 and one["spin_scans"] == 4096            # the measured optimum of one generator process on the MI355X host
    eight = sharding.host_budget(8, 2, cores=256)
    assert eight["cpu_share"] == 32 and eight["spin_scans"] < one["spin_scans"] and 2 <= eight["writers"] <= 16
    assert 8 * (2 + eight["writers"]) <= 256                              # generator + writer threads of all ranks fit the host
    tight = sharding.host_budget(8, 4, cores=16)
    assert tight["spin_scans"] <= 16 and tight["writers"] == 2
    shares = [sharding.affinity_for_local_rank(r, 8, 256) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert sorted(c for s in shares for c in s) == list(range(256))      # disjoint and complete
    assert all(max(s[:16]) < 64 for s in shares[:4]) and all(min(s[:16]) >= 64 for s in shares[4:])   # GPUs 0-3 / 4-7: socket 0 / 1
    assert sharding.affinity_for_local_rank(0, 1, 8) == list(range(8))
    # another kernel numbering (round-4 advisor item): sockets interleaved (package = cpu % 2), SMT sibling of core c is c + 16 -- the
    # shares follow the host's topology files, not a fixed formula: still disjoint, complete, whole cores, and one socket per rank
    topo = sorted((c % 2, [c, c + 16]) for c in range(16))
    shares = [sharding.affinity_for_local_rank(r, 8, topology=topo) for r in range(8)]
    assert sorted(c for s in shares for c in s) == list(range(32)) and all(len(s) == 4 for s in shares)
    assert all({c % 16 for c in s} == {c for c in s if c < 16} for s in shares)                     # a core comes with its sibling
    assert all(len({c % 2 for c in s}) == 1 for s in shares)                                           # one socket per rank
    assert [s[0] % 2 for s in shares] == [0, 0, 0, 0, 1, 1, 1, 1]                                      # local ranks 0-3 / 4-7: socket 0 / 1
    real = sharding.affinity_for_local_rank(0, 1)                                                      # this host's own topology
    assert real == sorted(real) and len(real) >= 1


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world), OCTA_NO_AFFINITY="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    budget = sharding.apply_host_budget(generator_threads=2, set_affinity=False)
    # what bench.py reports as rccl_ranks: an all-reduce of ones
    ones = torch.ones(1)
    dist.all_reduce(ones)
    seeds = np.concatenate([sharding.rank_seeds(rank, step, 128) for step in range(2)])
    plan = sharding.plan_batches(5000, 512, rank, world)
    torch.manual_seed(100 + rank)                      # different init per rank: must be overwritten by rank 0's
    tr = SegmentationTrainer(CFG, "cpu", channels_last=False)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(world, 1, 16, 16, generator=g)
    y = (torch.rand(world, 1, 16, 16, generator=g) > 0.6).float()
    tr.perform_training_step({"image": x[rank:rank + 1], "label": y[rank:rank + 1]})
    t = sharding.max_over_ranks(1.0 + rank, dist)
    np.save(out + f".seeds.{rank}.npy", seeds)
    np.save(out + f".plan.{rank}.npy", np.array(plan, dtype=np.int64).reshape(-1, 2))
    if rank == 0:
        torch.save({"state": {k: v.clone() for k, v in tr.model.state_dict().items() if not k.startswith("skip_layers")},
                    "ranks": float(ones.item()), "t": t, "spin": os.environ["OCTA_SIM_SPIN_SCANS"], "budget": budget}, out + ".pt")
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_seeds_plans_arena_and_rank_count(tmp_path):
    import torch.multiprocessing as mp
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(WORLD, port, out), nprocs=WORLD, join=True)
    got = torch.load(out + ".pt", weights_only=False)
    assert got["ranks"] == float(WORLD) and got["t"] == float(WORLD)
    assert int(got["spin"]) < 4096                                   # eight ranks on one host: the service threads give their cores back sooner
    seeds = [np.load(out + f".seeds.{r}.npy") for r in range(WORLD)]
    allseeds = np.concatenate(seeds)
    assert len(set(allseeds.tolist())) == len(allseeds) == WORLD * 256
    plans = sorted(tuple(p) for r in range(WORLD) for p in np.load(out + f".plan.{r}.npy").tolist())
    assert plans == sorted(sharding.plan_batches(5000, 512))
    # the eight ranks' step (one sample each, gradients averaged through the flat arena) == one process over the batch of eight
    torch.manual_seed(100)
    ref = SegmentationTrainer(CFG, "cpu", channels_last=False)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(WORLD, 1, 16, 16, generator=g)
    y = (torch.rand(WORLD, 1, 16, 16, generator=g) > 0.6).float()
    ref.optimizer.zero_grad()
    l = sum(ref.loss_function(ref.model(x[i:i + 1]), y[i:i + 1]) for i in range(WORLD)) / WORLD
    l.backward()
    ref.optimizer.step()
    for k, v in ref.model.state_dict().items():
        if k.startswith("skip_layers"):
            continue
        assert torch.allclose(got["state"][k], v, atol=2e-6), k
