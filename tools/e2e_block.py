"""Wall-clock of `mps cusk` on one synthetic 10k-SNP block, end to end (files in, files out), with a phase breakdown
(the host program prints its own timings with CUSK_TIMING=1)."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from cigwas_amd import synth
from cigwas_amd.cli import MPS_PATH

m, p, N = int(os.environ.get("M", 10000)), 20, 16384
d = tempfile.mkdtemp(prefix="e2e_", dir="/tmp")
t0 = time.time()
bed, phen, means, stds, _ = synth.synth_bed_block(m, N, p, block_index=0)
stem = os.path.join(d, "blk")
synth.write_bfiles(stem, bed, N, means, stds)
synth.write_phen(os.path.join(d, "blk.phen"), phen, N, p)
with open(os.path.join(d, "blk.blocks"), "w") as f:
    f.write(f"1\t0\t{m - 1}\n")
print(f"generated + wrote inputs in {time.time() - t0:.1f} s", flush=True)
out = os.path.join(d, "out"); os.makedirs(out)
env = dict(os.environ, CUSK_TIMING="1")
for rep in range(1):
    t0 = time.perf_counter()
    r = subprocess.run([MPS_PATH, "cusk", os.path.join(d, "blk.phen"), stem, os.path.join(d, "blk.blocks"), "0.0001", "5", "14", "1", out, "0"],
                       capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    print(f"run {rep}: rc={r.returncode} wall {dt:.3f} s")
    print("\n".join(l for l in r.stdout.splitlines() if "[t]" in l or "level" in l.lower() or "Retained" in l)[:3000])
    if r.returncode: print(r.stderr[-2000:])
import hashlib
print(sorted((f, os.path.getsize(os.path.join(out, f)), hashlib.sha256(open(os.path.join(out, f), "rb").read()).hexdigest()[:16])
             for f in os.listdir(out)))
