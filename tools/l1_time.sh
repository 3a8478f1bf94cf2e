# level-1 row kernel timing under engine options: bash tools/l1_time.sh "l1_exp=0" "l1_exp=1024" ...
for o in "$@"; do
  args=""; for kv in $(echo $o | tr ',' ' '); do args="$args --option $kv"; done
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-chromosome $args 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$o', 'ms_per_step %.4f' % d['ms_per_step'], 'l1_kernel_ms %.4f' % d['roofline']['kernel_ms_per_step'], 'levels', {k: round(v['level_ms'],4) for k,v in d['levels'].items()})"
done
