for cfg in "512 3 6 3 256" "512 3 6 3 128" "512 4 8 4 128" "768 3 6 3 256" "512 4 8 4 256" "384 4 8 4 192"; do set -- $cfg; OCTA_SIM_GRID=$5 python bench.py --batch $1 --inflight $2 --steps $3 --warmup $4 --no-train --no-cpu-baseline --no-files 2>/dev/null | python -c "
import sys,json
g='$5'
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch',d['config']['batch_per_gpu'],'inflight',d['config']['steps_in_flight'],'grid',g,'value %.1f'%d['value'],'launch_ms %.0f'%d['roofline']['avg_launch_ms'])"; done
