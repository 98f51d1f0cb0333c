"""Sample sharding of the generator path across ranks: samples are independent units, every rank
generates its own seeds, results are placement-independent because seeding is per sample; the only communication is
the barrier and the MAX of the wall time that bench.py reports."""
import numpy as np


def rank_seeds(rank, step, batch, base=7):
    """Seeds of the `batch` samples rank `rank` generates in step `step`: disjoint across ranks (< 100 000 steps x
    batch) and across steps (batch <= 1000)."""
    assert 0 < batch <= 1000 and step >= 0 and rank >= 0
    return (np.arange(batch, dtype=np.int64) + 100000 * rank + 1000 * step + base).astype(np.uint32)


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX of a python float over all ranks (identity without torch.distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
