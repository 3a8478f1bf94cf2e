# A/B of prebuilt library variants on the GPU box: bash tools/variants.sh variants/libA.so variants/libB.so ...
for v in "$@"; do
  cp $v ci-gwas_amd/csrc/libcusk_hip.so
  echo "== $v"
  bash tools/l1_time.sh l1_exp=0 l1_exp=0 2>&1 | cut -c 1-100
  python bench.py --workload chromosome --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d.get('scale',d)
print('  chromosome blocks/s %.0f compute_ms %.3f' % (s['blocks_per_sec'], s['compute_ms']))"
done
