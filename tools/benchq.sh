timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b.log 2>&1; python - <<PY
import json
for l in open("gpurun_out/b.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"]); print({k:(round(v["sweep_ms"],3),round(v["level_ms"],3),v["tests"]) for k,v in d["levels"].items()})
PY
