# usage: tools/benchopt.sh tag [--option k=v ...]  -- one quick bench line with per-level times
tag=$1; shift
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-chromosome "$@" > gpurun_out/b_$tag.log 2>&1; python - <<PY
import json
for l in open("gpurun_out/b_$tag.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]); print({k:(round(v["sweep_ms"],3),round(v["level_ms"],3)) for k,v in d["levels"].items()})
PY
