"""GPU: does the NUMA node of the service thread / the pinned mailbox memory matter? Runs tools/sim_mailbox_profile.py's measurement with
the process bound to the CPUs of one node at a time (the mailbox is allocated and served by this process's threads)."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
gpu_nodes = [open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")]
print("nodes:", [os.path.basename(n) for n in nodes], "GPU numa_node entries:", gpu_nodes)
for n in nodes:
    cpus = open(os.path.join(n, "cpulist")).read().strip()
    code = f"import os,re;\ncl=[]\nfor part in '{cpus}'.split(','):\n    a,_,b=part.partition('-'); cl+=list(range(int(a), int(b or a)+1))\nos.sched_setaffinity(0, cl)\nimport runpy, sys\nsys.argv=['x','512']\nrunpy.run_path('{root}/tools/sim_mailbox_profile.py', run_name='__main__')\n"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ))
    line = [l for l in out.stdout.splitlines() if l.startswith("B=")]
    print(os.path.basename(n), "cpus", cpus[:40], "->", line[-1][:170] if line else out.stderr[-300:])
