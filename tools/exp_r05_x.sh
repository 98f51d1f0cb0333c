#!/bin/bash
# kernel time by family in the on-the-fly U-Net training loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python -c "
import train_synthetic
info = train_synthetic.run(steps=256, batch=4, gen_batch=512, seed0=500000, log=False, warmup=128)
print(info)
" > /tmp/kt2.log 2>&1
tail -1 /tmp/kt2.log | cut -c1-400
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
fam = {}
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name']); n = re.sub(r'^void ', '', n).split('(')[0]
    ms = int(r['TotalDurationNs']) / 1e6
    k = ('sim' if n.startswith('sim_') or 'kd_' in n else 'raster' if n.startswith('raster') or 'dither' in n else
         'conv' if ('conv3x3' in n or 'wgrad' in n) else 'norm' if n.startswith('in_nhwc') else 'torch' if n.startswith('at::') else 'other:' + n[:40])
    fam[k] = fam.get(k, 0) + ms
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v:10.1f} ms  {k}")
PY
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
print("-- torch kernels")
for r in rows:
    n = re.sub(r'^void ', '', r['Name'])
    if n.startswith('at::'):
        ms = int(r['TotalDurationNs']) / 1e6
        if ms > 8: print(f"{ms:8.1f} ms {int(r['Calls']):6d} calls avg {float(r['AverageNs'])/1e3:7.1f} us  {n[:230]}")
PY
