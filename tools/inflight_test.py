"""How much does running K independent blocks concurrently on one GPU (K engines, K host threads) buy?"""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import cigwas_amd as cg
from cigwas_amd import synth

m, p, N = 10000, 20, 16384
n = m + p
bed, phen, means, stds, _ = synth.synth_bed_block(m, N, p, block_index=0)
e0 = cg.Engine(0)
Cd = cg.DeviceArray(nbytes=4 * n * n)
e0.corr_build(bed, phen, m, N, p, means, stds, Cd.ptr)
Th = cg.threshold_array(N, 1e-4)
for K in (1, 2, 3, 4):
    engs = [cg.Engine(0) for _ in range(K)]
    for e in engs:
        e.set_option("assume_symmetric", 1)
        e.run_skeleton(Cd.ptr, n, Th, 5)
    steps = 20
    tests = [0] * K
    def work(i):
        for _ in range(steps):
            st = engs[i].run_skeleton(Cd.ptr, n, Th, 5)
            tests[i] += sum(st.tests)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"K={K}: {sum(tests)/dt:.4g} tests/s, {dt/steps*1e3:.3f} ms per round of {K} blocks, {dt/steps/K*1e3:.3f} ms/block", flush=True)
    for e in engs:
        e.close()
