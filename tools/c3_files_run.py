"""BASELINE config 3 through the FILE path once: `mps cuskss` on a 50,000-SNP x 20-trait summary-statistics block
(5 GB mxm lower triangle, mxp / pxp text), max level 5, two stages -- wall clock and a cross-check of the written
adjacency against the engine fed directly with the same matrix.  usage: python tools/c3_files_run.py [markers]"""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
import torch
import cigwas_amd as cg
from cigwas_amd import synth
from cigwas_amd.cli import MPS_PATH

m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
p, N, alpha, l1, l2 = 20, 16384, 1e-4, 5, 5
n = m + p
d = tempfile.mkdtemp(prefix="c3_", dir="/tmp")
t0 = time.time()
Ct = synth.synth_corr_block_torch(m, p, N, block_index=3)
Ch = Ct.cpu().numpy()
del Ct
torch.cuda.empty_cache()
with open(os.path.join(d, "mxm.bin"), "wb") as f:  # lower triangle incl. diagonal, row-major
    for i0 in range(0, m, 2048):
        rows = Ch[i0:min(i0 + 2048, m), :m]
        f.write(np.concatenate([rows[k, : i0 + k + 1] for k in range(rows.shape[0])]).astype(np.float32).tobytes())
names = [f"T{k}" for k in range(p)]
with open(os.path.join(d, "mxp.txt"), "w") as f:
    f.write("chr snp ref " + " ".join(names) + "\n")
    f.write("".join(f"1 rs{i} A " + " ".join("%.9g" % v for v in Ch[i, m:]) + "\n" for i in range(m)))
with open(os.path.join(d, "pxp.txt"), "w") as f:
    f.write(" ".join(names) + "\n")
    for a in range(p):
        f.write(names[a] + " " + " ".join("%.9g" % v for v in Ch[m + a, m:]) + "\n")
with open(os.path.join(d, "blocks.txt"), "w") as f:
    f.write(f"1\t0\t{m - 1}\n")
out = os.path.join(d, "out"); os.makedirs(out)
print(f"inputs written in {time.time() - t0:.1f} s ({os.path.getsize(os.path.join(d, 'mxm.bin')) / 1e9:.2f} GB mxm)", flush=True)
t0 = time.perf_counter()
r = subprocess.run([MPS_PATH, "cuskss", os.path.join(d, "mxm.bin"), os.path.join(d, "mxp.txt"), "NULL", os.path.join(d, "pxp.txt"), "NULL",
                    "NULL", "0", os.path.join(d, "blocks.txt"), "NULL", str(alpha), str(l1), str(l2), "1", str(N), out],
                   capture_output=True, text=True)
dt = time.perf_counter() - t0
print(f"mps cuskss rc={r.returncode} wall {dt:.1f} s")
print(r.stdout[-600:], r.stderr[-600:])
stem = os.path.join(out, f"1_0_{m - 1}")
nv, nph, ml = [int(v) for v in open(stem + ".mdim").read().split()]
ixs = np.fromfile(stem + ".ixs", np.int32)
adj = np.fromfile(stem + ".adj", np.int32).reshape(nv, nv)
# the same two stages on the engine directly (text round trip of mxp/pxp: %.9g is exact for float32)
eng = cg.Engine(0)
Cd = cg.DeviceArray(Ch)
ti = np.zeros(n, np.int32); ti[m:] = 1
th = cg.hetcor_threshold(alpha)
st = eng.run_hetcor(Cd.ptr, n, th, l1, ess_uniform=float(N), time_index=ti)
G = eng.adjacency_bits()
print(f"direct engine stage one: {st.total_ms:.1f} ms, {sum(st.tests):.3e} tests; file path retained {nv - nph} markers")
# retained set = traits + markers adjacent to a trait (depth 1) after stage one; every retained marker of the file
# run must be among them, and edges of the file run must be edges of stage one
bits = np.unpackbits(G.view(np.uint8), axis=1, bitorder="little")[:, :n]
keep1 = np.nonzero(bits[m:, :m].any(axis=0))[0]
assert set(ixs[:-p].tolist()) <= set(keep1.tolist()), "file run retained a marker the direct stage one does not"
sub = bits[np.ix_(ixs, ixs)]
assert np.all(adj <= sub), "file run has an edge the direct stage one does not have"
print("C3 file path consistent with the direct engine run; files:", sorted((f, os.path.getsize(os.path.join(out, f))) for f in os.listdir(out)))
