#!/bin/bash
# Quick simulator check on the GPU box: parity tests of the simulator + the headline bench leg alone (no CPU baseline / training legs).
# usage: tools/sim_quick.sh <tag>
tag=${1:-quick}
python -m pytest tests/test_sim_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x -k "not voxel and not dynunet and not unet and not conv" 2>&1 | tail -5 > gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_tests.log
python bench.py --no-cpu-baseline --no-train --no-end-to-end --no-files > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("samples/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "launch ms", round(r["avg_launch_ms"], 1),
      "per-sample ms", round(r["serial_depth"]["per_sample_device_ms"], 1), "solo launch", round(r["serial_depth"]["solo_launch_ms"], 1))
print({k: v for k, v in r["serial_depth"].items() if "phase" in k or "ms" in k})
PY
