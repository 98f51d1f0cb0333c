#!/bin/bash
cd "$(dirname "$0")/.."
for mb in 0 200 400; do
  OCTA_NORM_CACHE_MB=$mb bash tools/step_trace.sh r05_mall$mb > /dev/null 2>&1
  python - $mb <<'PY'
import csv,sys,collections
mb=sys.argv[1]
rows=list(csv.DictReader(open(f'gpurun_out/r05_mall{mb}_step_launches.csv')))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=r['kernel'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0]
    if k.startswith('in_nhwc'): agg[k][0]+=1; agg[k][1]+=float(r['dur_us'])
print('OCTA_NORM_CACHE_MB',mb,'step kernel ms',sum(float(r['dur_us']) for r in rows)/1e3, {k:(v[0],round(v[1])) for k,v in agg.items()})
PY
done
