# PMC pass(es) over the default bench: bash tools/pmc.sh "<kernel substring>" "CTR1 CTR2 ..." ["CTR3 ..."]
# one rocprofv3 run per counter group (separate passes), per-kernel averages printed and kept in gpurun_out/pmc_*.txt
R=$PWD; PAT="$1"; shift; i=0
for G in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc$i
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 150 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/pmc$i -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chromosome $BENCH_EXTRA > $R/gpurun_out/pmc$i.log 2>&1) || { tail -5 $R/gpurun_out/pmc$i.log; exit 1; }
  python3 - "$R/gpurun_out/pmc$i" "$PAT" <<'PY' | tee $R/gpurun_out/pmc_$i.txt
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:50], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, c) in sorted(acc.items()):
    print(f"{kn:50s} {cn:28s} launches {c:3d} avg {v/c:.5g}")
PY
done
