"""Wall-clock of `mps block` on one synthetic chromosome (files in, .blocks out), phases with CUSK_TIMING=1."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from cigwas_amd import synth
from cigwas_amd.cli import MPS_PATH

m, N, width, maxb = int(os.environ.get("M", 40000)), 16384, int(os.environ.get("W", 2000)), int(os.environ.get("B", 10000))
d = tempfile.mkdtemp(prefix="blk_", dir="/tmp")
t0 = time.time()
bed, _phen, means, stds, _ = synth.synth_bed_block(m, N, 1, block_index=7)
stem = os.path.join(d, "chr")
synth.write_bfiles(stem, bed, N, means, stds)
print(f"generated + wrote inputs in {time.time() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
r = subprocess.run([MPS_PATH, "block", stem, str(maxb), "200", str(width)], capture_output=True, text=True,
                   env=dict(os.environ, CUSK_TIMING="1"))
print(f"rc={r.returncode} wall {time.perf_counter() - t0:.3f} s")
print("\n".join(l for l in r.stdout.splitlines() if "[t]" in l or "Partitioned" in l))
if r.returncode: print(r.stderr[-2000:])
print(open(stem + f"_m{maxb}.blocks").read())
