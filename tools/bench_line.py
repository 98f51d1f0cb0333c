"""Pick fields out of bench.py's JSON line: `python bench.py ... | python tools/bench_line.py value slot_cycle.render_enqueue_ms files.cli_pipelined.value`."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
out = []
for path in sys.argv[1:]:
    v = d
    for k in path.split("."):
        v = v[k] if isinstance(v, dict) and k in v else None
        if v is None:
            break
    out.append(f"{path}={v:.4g}" if isinstance(v, float) else f"{path}={v}")
print(" ".join(out))
