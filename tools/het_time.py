"""Per-level times of the hetcor engine with per-pair effective sample sizes (cuskss-het) on a 10k-SNP block."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import cigwas_amd as cg
from cigwas_amd import synth

m, p, N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 15, 16384
n = m + p
Ct = synth.synth_corr_block_torch(m, p, N, block_index=5)
g = torch.Generator(device="cuda"); g.manual_seed(5)
Nt = torch.full((n, n), float(N), dtype=torch.float32, device="cuda")
e_ = (torch.rand((n, p), generator=g, device="cuda") * 0.5 + 0.5) * N
Nt[:, m:] = e_; Nt[m:, :] = e_.T; Nt[m:, m:] = torch.maximum(Nt[m:, m:], Nt[m:, m:].T)
torch.cuda.synchronize()
eng = cg.Engine(0)
th = cg.hetcor_threshold(1e-4)
ti = np.zeros(n, np.int32); ti[m:] = 1
for rep in range(3):
    st = eng.run_hetcor(Ct.data_ptr(), n, th, 5, N_dev=Nt.data_ptr(), time_index=ti)
print("total_ms", st.total_ms, {l: (round(st.kernel_ms[l], 3), round(st.level_ms[l], 3), st.tests[l], st.max_degree[l]) for l in range(st.levels_run)})
