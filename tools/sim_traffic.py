"""GPU only: L2-miss traffic of the simulator's persistent kernel per 512-sample launch, FETCH_SIZE and WRITE_SIZE apart.

    python tools/sim_traffic.py [--batch 512] [NAME ...]

Each NAME is a variant library gpurun_variants/liboctahip_NAME.so (tools/build_sim_variant.py; `default` = the regular build). One
`rocprofv3 --pmc <counter>` pass per counter and library over `bench.py --pmc-child` (one warm-up and one measured launch), nothing
else on the GPU -- MI355X_MICROARCH.md's recipe: counters in their own passes, no tracing. Printed per library: FETCH_SIZE and
WRITE_SIZE in GB per launch as the counters report them (KB units), the corrected sum 2 x FETCH + WRITE that bench.py's
`roofline.traffic` carries, and the difference to the first library of the list. With -DOCTA_SIM_DUP=<bit> variants (an idempotent
part of the iteration run twice) that difference is the part's share of the traffic."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "sim_persistent_kernel"


def counters(lib, batch):
    exe = shutil.which("rocprofv3")
    if exe is None:
        raise SystemExit("rocprofv3 not on PATH")
    out = {}
    tmp = tempfile.mkdtemp(prefix="octa_pmc_", dir="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            env = dict(os.environ, TMPDIR="/tmp")
            if lib:
                env["OCTA_HIP_LIB"] = lib
            r = subprocess.run([exe, "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child",
                                "--batch", str(batch)], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            if r.returncode != 0:
                raise SystemExit(f"rocprofv3 --pmc {c} failed (rc {r.returncode}): {r.stdout[-400:]}")
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == c and KERNEL in row.get("Kernel_Name", ""):
                        tot += float(row["Counter_Value"])
                        n += 1
            out[c] = tot / max(n, 1) * 1024.0 / 1e9
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    args = sys.argv[1:]
    batch = 512
    if "--batch" in args:
        i = args.index("--batch")
        batch = int(args[i + 1])
        del args[i:i + 2]
    names = args or ["default"]
    first = None
    for name in names:
        lib = None if name == "default" else os.path.join(ROOT, "gpurun_variants", f"liboctahip_{name}.so")
        c = counters(lib, batch)
        tot = 2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]
        if first is None:
            first = (c, tot)
        print(f"{name:12s} FETCH_SIZE {c['FETCH_SIZE']:8.1f} GB  WRITE_SIZE {c['WRITE_SIZE']:8.1f} GB  2F+W {tot:8.1f} GB per {batch}-sample launch"
              f"   delta: fetch {c['FETCH_SIZE'] - first[0]['FETCH_SIZE']:+8.1f}  write {c['WRITE_SIZE'] - first[0]['WRITE_SIZE']:+8.1f}  2F+W {tot - first[1]:+8.1f}"
              f"   per sample and iteration: {tot / batch / 250 * 1e3:6.2f} MB", flush=True)


if __name__ == "__main__":
    main()
