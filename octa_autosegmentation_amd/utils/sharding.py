"""Sample sharding of the generator path across ranks: samples are independent units, every rank
generates its own seeds, results are placement-independent because seeding is per sample; the only communication is
the barrier and the MAX of the wall time that bench.py reports."""
import numpy as np


MAX_RANKS = 64            # ranks of one job (one node has 8 GPUs; the seed space is split for up to 64)
SEEDS_PER_RANK = (2 ** 32) // MAX_RANKS


def rank_seeds(rank, step, batch, base=7):
    """Seeds of the `batch` samples rank `rank` generates in step `step`. The 32-bit seed space (numpy's `seed(int)` and the
    simulator's `random.seed` take 32 bits here) is cut into MAX_RANKS disjoint ranges of 2^26 seeds; inside its range a rank
    walks consecutive blocks of `batch` seeds, so ranks never meet and steps never repeat for step * batch < 2^26 - base
    (524 000 steps of 128 samples); beyond that the call FAILS instead of wrapping into another rank's range."""
    rank, step, batch, base = int(rank), int(step), int(batch), int(base)
    assert batch > 0 and step >= 0 and 0 <= rank < MAX_RANKS and base >= 0
    first = base + step * batch
    if first + batch > SEEDS_PER_RANK:
        raise OverflowError(f"rank_seeds: step {step} x batch {batch} (+ base {base}) leaves rank {rank}'s range of {SEEDS_PER_RANK} seeds")
    return (np.arange(batch, dtype=np.int64) + rank * SEEDS_PER_RANK + first).astype(np.uint32)


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX of a python float over all ranks (identity without torch.distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
