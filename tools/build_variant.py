"""Build an experiment variant of liboctahip.so: one source file (every build of it) recompiled with extra -D flags, every other
object taken from the regular build. `python tools/build_variant.py NAME conv.hip -DFOO -DBAR=1` writes
gpurun_variants/liboctahip_NAME.so; select it at run time with OCTA_HIP_LIB=<path>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octa_autosegmentation_amd import build as B

def main():
    name, which, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build()
    objdir = os.path.join(B.CSRC, "build")
    outdir = os.path.join(ROOT, "gpurun_variants")
    os.makedirs(outdir, exist_ok=True)
    hipcc = B.hipcc_path()
    cflags = [f for f in B.FLAGS if f not in ("-shared", "-ldl", "-lz")]
    objs = []
    procs = []
    for s in B.SOURCES:
        src, _, variant = s.partition("@")
        if src != which:
            objs.append(os.path.join(objdir, s + ".o"))
            continue
        if variant == "large" and "--with-large" not in flags:
            objs.append(os.path.join(objdir, s + ".o"))
            continue
        obj = os.path.join(outdir, f"{name}.{s}.o")
        extra = ["-DOCTA_SIM_LARGE=1"] if variant == "large" else []
        cmd = [hipcc] + cflags + extra + [f for f in flags if f != "--with-large"] + ["-c", os.path.join(B.CSRC, src), "-o", obj]
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise SystemExit("failed: " + " ".join(cmd))
    lib = os.path.join(outdir, f"liboctahip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-ldl", "-lz"])
    print(lib)

if __name__ == "__main__":
    main()
