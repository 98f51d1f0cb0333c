"""Build liboctahip.so (HIP, gfx950) in-tree with hipcc. `python -m octa_autosegmentation_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liboctahip.so")
SOURCES = ["common.cpp", "bif_native.cpp", "fileio.cpp", "sim_api.cpp", "raster.hip", "sim.hip", "sim.hip@large", "voxel.hip", "graphio.hip", "order.hip", "norm.hip", "conv.hip", "conv_f32.hip", "augment.hip", "postproc.hip", "loss.hip", "blur.hip", "thin_conv.hip"]
HEADERS = ["common.h", "raster_core.h", "sim_core.h", "sim_host.h", "gpow.h", "glibc_pow_tables.h", "glibc_trig.h", "glibc_trig_tables.h",
           os.path.join("..", "..", "include", "octa_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-ldl", "-lz"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s.partition("@")[0]) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _stale(obj, src, hdr_time):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return os.path.getmtime(src) > t or hdr_time > t


def build(force=False, verbose=False):
    """One object per source (compiled in parallel, only the stale ones), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = hipcc_path()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    common = ["common.h", os.path.join("..", "..", "include", "octa_hip.h")]
    own = {"raster.hip": ["raster_core.h"], "sim.hip": ["sim_core.h", "sim_host.h", "gpow.h", "glibc_pow_tables.h", "glibc_trig.h", "glibc_trig_tables.h"]}
    mt = lambda hs: max(os.path.getmtime(os.path.join(CSRC, h)) for h in hs)
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl", "-lz")] + os.environ.get("OCTA_EXTRA_HIPCC_FLAGS", "").split()   # experiments only
    jobs = []
    for s in SOURCES:
        # "file@variant": the same source compiled again with -DOCTA_SIM_LARGE=1 (the wide-field build of the simulator, csrc/sim_api.cpp)
        name, _, variant = s.partition("@")
        src, obj = os.path.join(CSRC, name), os.path.join(objdir, s + ".o")
        extra = ["-DOCTA_SIM_LARGE=1"] if variant == "large" else []
        if force or _stale(obj, src, mt(common + own.get(name, []))):
            jobs.append([hipcc] + cflags + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for res in ex.map(run, jobs):
            if res.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + res.stdout)
    res = run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [os.path.join(objdir, s + ".o") for s in SOURCES] + ["-ldl", "-lz"])
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
