"""GPU only: SQ counters of the simulator's persistent kernel over one 512-sample launch -- what its waves spend their cycles on.

    python tools/sim_counters.py [--batch 512] [--lib NAME] [COUNTER ...]

One `rocprofv3 --pmc <counter>` pass per counter over `bench.py --pmc-child` (one warm-up and one measured launch, nothing else on the
GPU; counters in their own passes, no tracing -- MI355X_MICROARCH.md's recipe). Prints each counter's per-launch value and a few
ratios: instructions per wave-cycle, share of the wave-cycles spent waiting, average cycles a vector-memory / LDS instruction is in
flight (SQ_INST_LEVEL_* / SQ_INSTS_*)."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "sim_persistent_kernel"
DEFAULT = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
           "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
           "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT", "SQ_INSTS_SMEM", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS",
           "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_THREAD_CYCLES_VALU", "SQ_INST_CYCLES_VMEM"]


def one(counter, lib, batch):
    exe = shutil.which("rocprofv3")
    tmp = tempfile.mkdtemp(prefix="octa_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        if lib:
            env["OCTA_HIP_LIB"] = lib
        r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child",
                            "--batch", str(batch)], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        if r.returncode != 0:
            return None
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == counter and KERNEL in row.get("Kernel_Name", ""):
                    tot += float(row["Counter_Value"])
                    n += 1
        return tot / n if n else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = sys.argv[1:]
    batch, lib = 512, None
    if "--batch" in args:
        i = args.index("--batch"); batch = int(args[i + 1]); del args[i:i + 2]
    if "--lib" in args:
        i = args.index("--lib"); lib = os.path.join(ROOT, "gpurun_variants", f"liboctahip_{args[i + 1]}.so"); del args[i:i + 2]
    v = {}
    for c in (args or DEFAULT):
        v[c] = one(c, lib, batch)
        print(f"{c:24s} {v[c] if v[c] is None else format(v[c], '.4g')}", flush=True)
    g = lambda k: v.get(k) or 0.0
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        insts = g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_LDS") + g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR") + g("SQ_INSTS_FLAT") + g("SQ_INSTS_SMEM")
        print(f"instructions per wave-cycle              {insts / wc:.4f}   (VALU {g('SQ_INSTS_VALU') / wc:.4f}, SALU {g('SQ_INSTS_SALU') / wc:.4f}, LDS {g('SQ_INSTS_LDS') / wc:.4f}, "
              f"VMEM {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR')) / wc:.4f})")
        print(f"wave-cycles waiting (any / for an instruction)   {g('SQ_WAIT_ANY') / wc:.3f} / {g('SQ_WAIT_INST_ANY') / wc:.3f};  issuing {g('SQ_ACTIVE_INST_ANY') / wc:.3f} "
              f"(VALU {g('SQ_ACTIVE_INST_VALU') / wc:.3f}, scalar {g('SQ_ACTIVE_INST_SCA') / wc:.3f}, LDS {g('SQ_ACTIVE_INST_LDS') / wc:.3f}, VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:.3f})")
    vm = g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")
    if vm and g("SQ_INST_LEVEL_VMEM"):
        print(f"average cycles a vector-memory instruction is in flight   {g('SQ_INST_LEVEL_VMEM') / vm:.0f}")
    if g("SQ_INSTS_LDS") and g("SQ_INST_LEVEL_LDS"):
        print(f"average cycles an LDS instruction is in flight            {g('SQ_INST_LEVEL_LDS') / g('SQ_INSTS_LDS'):.0f}")
    if g("SQ_ACTIVE_INST_LDS") and g("SQ_LDS_BANK_CONFLICT"):
        print(f"LDS bank-conflict cycles / LDS active cycles              {g('SQ_LDS_BANK_CONFLICT') / g('SQ_ACTIVE_INST_LDS'):.3f}")
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print(f"lanes active per VALU instruction                          {g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU') if g('SQ_ACTIVE_INST_VALU') else 0:.1f} (thread-cycles / active cycles)")


if __name__ == "__main__":
    main()
