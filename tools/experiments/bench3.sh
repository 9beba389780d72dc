#!/bin/bash
# three bench lines of the current build: default window (200 frames), the driver's window (20 frames from frame 5), 1024^3
tag=${1:-cur}
mkdir -p gpurun_out/$tag
python bench.py --no-cpu-baseline > gpurun_out/$tag/b512.json 2>gpurun_out/$tag/b512.err
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/$tag/b512_driver.json 2>/dev/null
python bench.py --no-cpu-baseline --volume 1024 --steps 60 > gpurun_out/$tag/b1024.json 2>/dev/null
python - "$tag" <<'PY'
import json, sys
for f in ("b512", "b512_driver", "b1024"):
    try:
        j = json.loads(open(f"gpurun_out/{sys.argv[1]}/{f}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, j["value"], "fps", j["stage_us"]["integrate"], "us integrate", r["frac"], "frac | icp", j["stage_us"]["icp"], "raycast", j["stage_us"]["raycast"])
    except Exception as e:
        print(f, "failed", e)
PY
