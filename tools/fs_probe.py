"""How fast are small result files on this box's TMPDIR right now?  125 files / 8.6 MB per pass like a 25-block batch."""
import os, sys, tempfile, time
import numpy as np
d = tempfile.mkdtemp(prefix="fsprobe_", dir=os.environ.get("TMPDIR", "/tmp"))
buf = np.full(270_000 // 4, -1, np.int32)
small = np.zeros(5_000, np.int32)
ts = []
for p in range(6):
    o = os.path.join(d, str(p)); os.makedirs(o)
    t0 = time.perf_counter()
    for b in range(25):
        for ext, a in ((".mdim", small[:3]), (".ixs", small[:70]), (".adj", small), (".corr", small), (".sep", buf)):
            with open(os.path.join(o, f"b{b}{ext}"), "wb") as f:
                f.write(a.tobytes())
    ts.append((time.perf_counter() - t0) * 1e3)
print("fs probe ms per 125 files:", [round(t, 2) for t in ts], flush=True)
import shutil; shutil.rmtree(d)
