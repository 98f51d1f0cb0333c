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


def parse_devices(spec, n_visible=None):
    """`--devices` of the generator CLI: "0-7", "0,2,5", "all" or a single index -> list of device indices (duplicates allowed: two
    generator groups on one GPU, which is how a one-GPU box exercises the multi-group path). Raises on an index the process cannot see."""
    if spec is None:
        return None
    spec = str(spec).strip().lower()
    if spec in ("all", "*"):
        if n_visible is None:
            raise ValueError("--devices all needs the number of visible GPUs")
        out = list(range(int(n_visible)))
    else:
        out = []
        for part in spec.split(","):
            part = part.strip()
            if not part:
                continue
            if "-" in part:
                a, b = part.split("-", 1)
                a, b = int(a), int(b)
                if b < a:
                    raise ValueError(f"--devices: empty range {part!r}")
                out.extend(range(a, b + 1))
            else:
                out.append(int(part))
    if not out:
        raise ValueError("--devices names no device")
    if n_visible is not None:
        bad = [d for d in out if d < 0 or d >= int(n_visible)]
        if bad:
            raise ValueError(f"--devices: {bad} not among the {n_visible} visible GPUs")
    return out


def plan_batches(num_samples, batch, rank=0, world=1):
    """(start, count) of every simulator launch this PROCESS runs. Sample k is seeded by seed0 + k wherever it runs, so the plan only
    decides placement: the whole job is cut into launches of `batch` samples and launch j goes to rank j % world (torchrun: one
    generator process per GPU; the reference fans samples out over a process pool, generate_vessel_graph.py:112-129). Inside a process
    the launches are taken from one queue by all of its device groups."""
    num_samples, batch, rank, world = int(num_samples), int(batch), int(rank), int(world)
    assert num_samples >= 0 and batch > 0 and 0 <= rank < world
    plan, done, j = [], 0, 0
    while done < num_samples:
        n = min(batch, num_samples - done)
        if j % world == rank:
            plan.append((done, n))
        done += n
        j += 1
    return plan


def host_budget(local_world, generator_threads=1, cores=None):
    """Host-side budget of ONE process when `local_world` processes share the node (train.py / bench.py under torchrun, or the generator
    CLI's device groups counted as local_world): every generator thread runs a mailbox service loop that spins between tickets and calls
    LAPACK for the leaf bifurcations, and the file writers are host threads too. Returns the knobs the callers apply:
    `spin_scans` (idle mailbox scans before the service thread starts sleeping; OCTA_SIM_SPIN_SCANS), `writers` (file-writer threads),
    `blas_threads` (OPENBLAS_NUM_THREADS / OMP_NUM_THREADS for this process) and `cpu_share` (cores this process may count on).
    Measured on the two-socket 256-thread host of the MI355X boxes: 16 writers per generator process is the optimum when one process
    has the node to itself (DESIGN.md 4.2b'); the rest scales that down per process and never lets the spinning service threads of
    all processes exceed half of the cores."""
    import os
    cores = int(cores or os.cpu_count() or 1)
    local_world = max(1, int(local_world))
    share = max(1, cores // local_world)
    spinning = max(1, int(generator_threads))
    return {
        "cpu_share": share,
        "spin_scans": 4096 if local_world == 1 and share >= 4 * spinning else (256 if share >= 2 * spinning else 16),
        "writers": max(2, min(16, share // 2)),
        "blas_threads": 1,
    }


def _cpu_topology():
    """[(package id, [logical CPUs of one physical core])] from /sys/devices/system/cpu/cpu*/topology, sorted by package then core;
    None when the files are not there (non-Linux, restricted containers)."""
    import glob
    import os
    cores = {}
    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
        try:
            cpu = int(os.path.basename(d)[3:])
            with open(os.path.join(d, "topology", "thread_siblings_list")) as f:
                sib = f.read().strip()
            with open(os.path.join(d, "topology", "physical_package_id")) as f:
                pkg = int(f.read().strip())
        except (OSError, ValueError):
            continue
        members = []
        for part in sib.split(","):
            lo, _, hi = part.partition("-")
            members += list(range(int(lo), int(hi or lo) + 1))
        cores[(pkg, min(members))] = sorted(members)
        del cpu
    if not cores:
        return None
    return [(k[0], v) for k, v in sorted(cores.items())]


def affinity_for_local_rank(local_rank, local_world, cores=None, topology="auto"):
    """CPU set of local rank `local_rank`: a contiguous 1/local_world share of the PHYSICAL cores with all their SMT siblings, taken from
    the host's own topology files (packages in order, so that on the two-socket MI355X nodes -- GPUs 0-3 on socket 0, 4-7 on socket 1 --
    a rank's service and writer threads stay on its GPU's socket whatever the kernel's CPU numbering is). `topology`: "auto" reads
    /sys/devices/system/cpu; a list [(package, [cpus of a core])] is used as given (tests); None or an unreadable /sys falls back to the
    numbering of those hosts (core c and its sibling c + cores/2). Returns a sorted list; the caller applies it with
    os.sched_setaffinity (and ignores a refusal: containers may pin the process already)."""
    import os
    local_world = max(1, int(local_world))
    local_rank = int(local_rank) % local_world
    topo = _cpu_topology() if topology == "auto" and cores is None else (topology if isinstance(topology, list) else None)
    if topo:
        per = max(1, len(topo) // local_world)
        mine = topo[local_rank * per:(local_rank + 1) * per] or topo[-per:]
        return sorted(c for _, members in mine for c in members)
    cores = int(cores or os.cpu_count() or 1)
    phys = cores // 2 if cores >= 4 else cores            # fallback: SMT siblings of core c are c and c + cores/2
    per = max(1, phys // local_world)
    base = list(range(local_rank * per, min(phys, (local_rank + 1) * per)))
    sibs = [c + phys for c in base if c + phys < cores] if cores >= 4 else []
    return sorted(base + sibs)


def _limit_blas_threads(n):
    """The BLAS that numpy (and the native bifurcation service, which dlopens the same library) already LOADED reads
    OPENBLAS_NUM_THREADS at load time only: set the count through the library itself (threadpoolctl -> openblas_set_num_threads).
    Returns the libraries that were limited (for the bench line), or a reason."""
    try:
        import threadpoolctl
        ctl = threadpoolctl.ThreadpoolController().select(user_api="blas")
        ctl.limit(limits=int(n))           # stays in force for the life of the process (no context manager: never restored)
        return [f"{lib.internal_api}:{lib.num_threads}" for lib in ctl.lib_controllers]
    except Exception as e:  # noqa: BLE001 -- a budget, not a requirement
        return f"not applied ({type(e).__name__}: {e})"


def apply_host_budget(local_rank=None, local_world=None, generator_threads=1, set_affinity=True):
    """Apply host_budget() / affinity_for_local_rank() to this process from LOCAL_RANK / LOCAL_WORLD_SIZE (torchrun) unless given.
    Environment variables the user set win. Returns the budget (with the affinity that was applied, or None)."""
    import os
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    b = host_budget(local_world, generator_threads)
    os.environ.setdefault("OCTA_SIM_SPIN_SCANS", str(b["spin_scans"]))
    user_set = "OPENBLAS_NUM_THREADS" in os.environ
    os.environ.setdefault("OPENBLAS_NUM_THREADS", str(b["blas_threads"]))      # for libraries loaded from here on and for child processes
    os.environ.setdefault("OMP_NUM_THREADS", str(b["blas_threads"]))
    # numpy / torch are imported long before this call in every entry point: the already-loaded BLAS is limited through its own API
    b["blas_limited"] = "left alone (OPENBLAS_NUM_THREADS set by the user)" if user_set else _limit_blas_threads(b["blas_threads"])
    b["affinity"] = None
    if set_affinity and local_world > 1 and hasattr(os, "sched_setaffinity") and os.environ.get("OCTA_NO_AFFINITY") != "1":
        cpus = affinity_for_local_rank(local_rank, local_world)
        try:
            allowed = os.sched_getaffinity(0)
            cpus = [c for c in cpus if c in allowed] or sorted(allowed)
            os.sched_setaffinity(0, cpus)
            b["affinity"] = cpus
        except OSError:
            pass
    return b
