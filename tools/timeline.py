"""Print the kernel timeline of the last complete bench step from a rocprofv3 rocpd database."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/r_results.db")
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "level0_" in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
prev_end = t0
for r in rows[i0:i1]:
    print(f"{(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:7.1f}  gap {(r[1]-prev_end)/1e3:6.1f} s{r[3]} {r[0][:64]}")
    prev_end = max(prev_end, r[2])
