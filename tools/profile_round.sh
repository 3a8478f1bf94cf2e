# Profiles of the default bench for profiles/<tag>_*: run on the GPU box from the repo root:  bash tools/profile_round.sh <tag>
# 1. plain bench line; 2. rocprofv3 --kernel-trace --stats; 3. PMC passes (separate runs): FETCH_SIZE | WRITE_SIZE | SQ set
TAG=${1:-r01}; R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 -c "import torch" > /dev/null 2>&1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1 && grep '^{' $O/bench.log > $O/${TAG}_bench.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-chromosome > $O/trace.log 2>&1)
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_kernel_stats.csv
# the same trace, per kernel and GRID: a kernel that also runs on the small problems of the process (parity run of the
# oracle-sized block, second stages of the end-to-end block) averages over unlike launches in the stats file above
python3 - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" > $O/${TAG}_bench_kernel_stats_by_grid.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]))
    acc[k][0] += 1; acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("Name,Workgroups,WorkgroupSize,Calls,TotalDurationNs,AverageNs")
for (kn, g, w), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'"{kn}",{g},{w},{c},{int(t)},{t / c:.1f}')
PY
# the hetcor engine on the same block (bench.py --engine cuskss): bench line + kernel stats
timeout -k 10 600 python3 bench.py --engine cuskss --steps 20 --warmup 3 --no-chromosome > $O/bench_cuskss.log 2>&1 && grep '^{' $O/bench_cuskss.log > $O/${TAG}_bench_cuskss.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ss -o t -- python3 $R/bench.py --engine cuskss --steps 20 --warmup 3 --no-cpu-baseline --no-chromosome > $O/trace_ss.log 2>&1)
cp $(find $O/trace_ss -name '*kernel_stats.csv' | head -1) $O/${TAG}_bench_cuskss_kernel_stats.csv
# whole-chromosome job (the `scale` leg): kernel stats of the batched block driver alone
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chrom -o t -- python3 $R/bench.py --workload chromosome --steps 5 --warmup 1 > $O/trace_chrom.log 2>&1)
cp $(find $O/trace_chrom -name '*kernel_stats.csv' | head -1) $O/${TAG}_chromosome_kernel_stats.csv
grep '^{' $O/trace_chrom.log > $O/${TAG}_chromosome_bench.json
# kernel timeline of one headline step
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-chromosome > $O/p.log 2>&1); python3 tools/timeline.py $O/prof/r_results.db > $O/${TAG}_timeline.txt
# the sepselect path: kernel stats of one timing run, JSON line of the tool beside it
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss -o s -- python3 $R/tests/perf_sepselect.py --traits 80 --markers 8000 --sample-pairs 20 > $O/ss.log 2>&1)
cp $(find $O/ss -name '*kernel_stats.csv' | head -1) $O/${TAG}_sepselect_kernel_stats.csv
grep '^{' $O/ss.log > $O/${TAG}_sepselect.json
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/pmc$i -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chromosome > $O/pmc$i.log 2>&1)
done
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys, collections, json
O, TAG = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
rows = []
for f in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
# a kernel is also launched on the small problems of the run (parity check, second stages): only the launches with the
# kernel's largest grid -- the headline block -- are averaged
big = collections.defaultdict(int)
for r in rows:
    kn = r["Kernel_Name"].split("(")[0]
    big[kn] = max(big[kn], int(r["Grid_Size"]))
for r in rows:
    kn = r["Kernel_Name"].split("(")[0]
    if int(r["Grid_Size"]) != big[kn]:
        continue
    k = (kn, r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open(f"{O}/{TAG}_pmc_summary.txt", "w") as out:
    out.write("rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chromosome  (MI355X, three separate passes:\n"
              "FETCH_SIZE | WRITE_SIZE | SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)\n"
              "per-launch averages over the launches with the kernel's largest grid (the headline block); FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them\n\n")
    last = None
    for (kn, cn), (v, c) in sorted(acc.items()):
        if kn != last:
            out.write(kn + "\n"); last = kn
        out.write(f"    {cn:28s} launches {c:4d}  avg {v/c:.5g}\n")
def per_launch(kpat):
    f = [v / c for (kn, cn), (v, c) in acc.items() if kpat in kn and cn == "FETCH_SIZE"]
    w = [v / c for (kn, cn), (v, c) in acc.items() if kpat in kn and cn == "WRITE_SIZE"]
    return (f[0] if f else 0.0) * 1024, (w[0] if w else 0.0) * 1024
K1 = "level1_rows2_kernel<0, false, 256, true>"
f1, w1 = per_launch(K1)
# calibration (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"): the level-0 kernel
# streams the upper triangle of the 10020^2 fp32 matrix once with the same 4-byte-per-lane coalesced loads
f0, _ = per_launch("level0_wide2_kernel")
n0 = 10020
known0 = 4.0 * n0 * (n0 - 1) / 2
cal = known0 / f0 if f0 > 0 else 1.0
json.dump({"level1": f1 * 2.0 + w1, "level1_guide_x2": f1 * 2.0 + w1, "level1_calibrated": f1 * cal + w1,
           "level1_fetch_reported": f1, "level1_write": w1, "fetch_calibration": cal,
           "level0_fetch_reported": f0, "level0_known_bytes": known0, "kernel": K1,
           "note": "HBM-side bytes per launch of " + K1 + ", separate PMC passes (" + TAG + "): FETCH_SIZE x 1024 x correction + WRITE_SIZE x 1024. "
                   "`level1` (what bench.py prints as roofline.traffic) uses the guide's gfx950 correction, FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM: "
                   "128-byte requests tallied at 64 bytes); `level1_calibrated` uses this run's own calibration instead: level0_wide2_kernel reads a known " +
                   str(int(known0)) + " bytes (upper triangle, 4-byte-per-lane coalesced loads) and FETCH_SIZE reports 1/fetch_calibration of them"},
          open(f"{O}/pmc_traffic.json", "w"), indent=1)
PY
ls $O | head -30
