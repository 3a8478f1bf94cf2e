"""GPU, BASELINE.json's full sizes (configs 3 and 5: 50k x 20 and 30k x 15 heterogeneous), beyond what the
oracle (or the reference: SURVEY 0.9) can run -- size-independent properties instead of an element-wise oracle:
symmetry, monotonicity, determinism, cross-engine agreement, and exact CPU re-evaluation of sampled decisions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ML = 14


def _bits_to_deg(bits):
    return np.array([int(np.unpackbits(r.view(np.uint8)).sum()) for r in bits])


def _get(bits, i, j):
    return (int(bits[i, j >> 6]) >> (j & 63)) & 1


@pytest.mark.timeout(900)
def test_config3_cuskss_50k_properties(oracle, synth):
    import torch

    import cigwas_amd as cg

    m, p, N, alpha, lmax = 50000, 20, 16384, 1e-4, 5
    n = m + p
    Ct = synth.synth_corr_block_torch(m, p, N, block_index=3)
    torch.cuda.synchronize()
    eng = cg.Engine(0)
    th = cg.hetcor_threshold(alpha)
    ti = np.zeros(n, np.int32)
    ti[m:] = 1
    st = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, ess_uniform=float(N), time_index=ti)
    bits = eng.adjacency_bits()
    assert st.level == lmax + 1 and st.tests[1] > 0 and st.exact_fallbacks == 0
    # (1) undirected: bit (i,j) == bit (j,i) on a sample of rows; diagonal clear
    rng = np.random.default_rng(0)
    rows = np.concatenate([rng.integers(0, n, 300), np.arange(m, n)])
    for i in rows:
        nb = np.nonzero(np.unpackbits(bits[i].view(np.uint8), bitorder="little")[:n])[0]
        assert i not in nb
        for j in nb[:50]:
            assert _get(bits, int(j), int(i)) == 1
    # (2) deterministic: a second run gives the identical bitmap
    st2 = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, ess_uniform=float(N), time_index=ti)
    assert np.array_equal(eng.adjacency_bits(), bits)
    # (3) cross-engine: with a uniform sample size both engines threshold identically, so the Skeleton
    #     engine (different kernels: pair kernel, min-rank selection, finaliser) must give the same graph
    Th = cg.threshold_array(N, alpha)
    st3 = eng.run_skeleton(Ct.data_ptr(), n, Th, lmax)
    bits3 = eng.adjacency_bits()
    assert np.array_equal(bits3, bits)
    # (4) monotone: the final graph is a subgraph of the level-0 graph
    st0 = eng.run_skeleton(Ct.data_ptr(), n, Th, 0)
    bits0 = eng.adjacency_bits()
    assert np.all((bits & ~bits0) == 0)
    assert st0.level == 1
    # (5) sampled separating sets re-evaluated exactly on the CPU (oracle arithmetic): z < threshold, members adjacent to X at level 0
    eng.run_skeleton(Ct.data_ptr(), n, Th, lmax)
    x, y, lv, z, S = eng.sepsets()
    assert len(x) == sum(st3.removed[1:]) and len(x) > 1000
    pick = rng.choice(len(x), 400, replace=False)
    for r in pick:
        idx = np.array([x[r], y[r]] + [s for s in S[r] if s >= 0])
        sub = Ct[idx][:, idx].cpu().numpy()
        rho, zz = oracle.ci_test(sub, 0, 1, np.arange(2, len(idx), dtype=np.int32))
        assert zz < Th[lv[r]] and abs(zz - z[r]) <= 1e-6
        for s in S[r]:
            if s >= 0:
                assert _get(bits0, int(x[r]), int(s)) == 1
    eng.close()
    del Ct


@pytest.mark.timeout(900)
def test_config5_cuskss_het_30k_properties(oracle, synth):
    import torch

    import cigwas_amd as cg

    m, p, N, alpha, lmax = 30000, 15, 16384, 1e-4, 5
    n = m + p
    Ct = synth.synth_corr_block_torch(m, p, N, block_index=5)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    # per-pair effective sample sizes on the trait rows/columns (standard-error based): U(0.5, 1) N
    Nt = torch.full((n, n), float(N), dtype=torch.float32, device="cuda")
    e = (torch.rand((n, p), generator=g, device="cuda") * 0.5 + 0.5) * N
    Nt[:, m:] = e
    Nt[m:, :] = e.T
    Nt[m:, m:] = torch.maximum(Nt[m:, m:], Nt[m:, m:].T)
    torch.cuda.synchronize()
    eng = cg.Engine(0)
    th = cg.hetcor_threshold(alpha)
    ti = np.zeros(n, np.int32)
    ti[m:] = 1
    st = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, N_dev=Nt.data_ptr(), time_index=ti)
    bits = eng.adjacency_bits()
    assert st.level == lmax + 1 and st.exact_fallbacks == 0
    st2 = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, N_dev=Nt.data_ptr(), time_index=ti)
    assert np.array_equal(eng.adjacency_bits(), bits)  # deterministic adjacency despite racing removals
    # the filtered engine and the exact-only engine agree
    eng.set_option("fast", 0)
    st3 = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, N_dev=Nt.data_ptr(), time_index=ti)
    assert np.array_equal(eng.adjacency_bits(), bits)
    eng.set_option("fast", 1)
    # marker-marker pairs have ESS = N everywhere, so a uniform-ESS run can only differ on trait rows/columns
    eng.run_hetcor(Ct.data_ptr(), n, th, 0, ess_uniform=float(N), time_index=ti)
    b0u = eng.adjacency_bits()
    eng.run_hetcor(Ct.data_ptr(), n, th, 0, N_dev=Nt.data_ptr(), time_index=ti)
    b0h = eng.adjacency_bits()
    w = (m + 63) // 64
    diff = b0u ^ b0h
    mask_last = np.uint64((1 << (m & 63)) - 1) if (m & 63) else np.uint64(0xFFFFFFFFFFFFFFFF)
    d_mm = diff[:m, :w].copy()
    if m & 63:
        d_mm[:, w - 1] &= mask_last
    assert not d_mm[:, : w].any()
    # lower effective sample sizes raise thresholds: the heterogeneous level-0 graph is a subgraph of the uniform one
    assert np.all((b0h & ~b0u) == 0)
    eng.close()


@pytest.mark.timeout(900)
def test_config2_cusk_10k_block_from_bed_matches_oracle(oracle, synth):
    """BASELINE.json config 2 at full size: one LD block of 10,000 SNPs x 10 traits from packed .bed, max level 3.
    The correlation matrix is built on the device; the oracle sweeps that very matrix on the host cores.  Adjacency,
    level counter and every separating set must be identical, Fisher z of the winners within 1e-6."""
    import cigwas_amd as cg

    m, p, N, alpha, lmax = 10000, 10, 16384, 1e-4, 3
    n = m + p
    bed, phen, means, stds, _G = synth.synth_bed_block(m, N, p, block_index=2)
    del _G
    eng = cg.Engine(0)
    Cd = cg.DeviceArray(nbytes=4 * n * n)
    eng.corr_build(bed, phen, m, N, p, means, stds, Cd.ptr)
    Th = cg.threshold_array(N, alpha)
    st = eng.run_skeleton(Cd.ptr, n, Th, lmax)  # level 0 also verifies that the device-built matrix is symmetric
    G = eng.adjacency()
    x, y, lv, z, S = eng.sepsets()
    Ch = Cd.download(np.float32, (n, n))
    assert np.array_equal(Ch, Ch.T)
    # spot check of the build itself against the oracle's correlation build (bit-exact SNP x SNP, 1e-5 with traits)
    sel = np.r_[0:40, 5000:5040]
    o_mxm, o_mxp, _ = oracle.corr_pearson_npn(bed[sel], phen, len(sel), N, p, means[sel], stds[sel])
    want = oracle.square_from_cusk_corrs(o_mxm, o_mxp, np.zeros(p * (p - 1) // 2, np.float32), len(sel), p)
    assert np.array_equal(Ch[np.ix_(sel, sel)], want[: len(sel), : len(sel)])
    assert np.allclose(Ch[np.ix_(sel, np.arange(m, n))], want[: len(sel), len(sel):], atol=1e-5, rtol=0)
    ref = oracle.skeleton(Ch, oracle.threshold_array(N, alpha), lmax)
    assert st.level == ref.level == lmax + 1
    assert np.array_equal(G, ref.G)
    rx, ry = np.nonzero(ref.sepset[:, :, 0] != -1)
    assert len(rx) == len(x) > 100000 and np.array_equal(rx, x) and np.array_equal(ry, y)
    assert np.array_equal(ref.sepset[rx, ry], S)
    pm = eng.pmax(Cd.ptr)  # the winners' z, max over both directions, level-0 z of level-0 removals
    assert np.max(np.abs(pm - ref.pmax)) <= 1e-6
    assert st.exact_fallbacks == 0
    Cd.free()
    eng.close()


@pytest.mark.timeout(900)
def test_hetcor_10k_block_matches_oracle(oracle, synth):
    """A 10,000-SNP x 15-trait summary-statistics block with per-pair effective sample sizes on the trait rows and a
    time index, max level 3: hetcor engine vs the oracle, adjacency bit for bit (uniform-ESS form as well)."""
    import torch

    import cigwas_amd as cg

    m, p, N, alpha, lmax = 10000, 15, 16384, 1e-4, 3
    n = m + p
    Ct = synth.synth_corr_block_torch(m, p, N, block_index=11)
    rng = np.random.default_rng(11)
    Nh = np.full((n, n), float(N), np.float32)
    e = (rng.uniform(0.5, 1.0, (n, p)) * N).astype(np.float32)
    Nh[:, m:] = e
    Nh[m:, :] = e.T
    Nh[m:, m:] = np.maximum(Nh[m:, m:], Nh[m:, m:].T)
    Nh[m + 1, 17] = Nh[17, m + 1] = np.nan  # an NA correlation: kept at level 0 (hetcor-cuPC-S.cu:351-352)
    ti = np.zeros(n, np.int32)
    ti[m:] = 1 + (np.arange(p) % 3)
    Nt = torch.from_numpy(Nh).cuda()
    torch.cuda.synchronize()
    eng = cg.Engine(0)
    th = cg.hetcor_threshold(alpha)
    st = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, N_dev=Nt.data_ptr(), time_index=ti)
    G = eng.adjacency()
    Ch = Ct.cpu().numpy()
    ref = oracle.hetcor_skeleton(Ch, np.ones((n, n), np.int32), Nh, oracle.hetcor_threshold(alpha), lmax, ti)
    assert st.level == ref.level
    assert np.array_equal(G, ref.G)
    assert G[m + 1, 17] == ref.G[m + 1, 17]
    st_u = eng.run_hetcor(Ct.data_ptr(), n, th, lmax, ess_uniform=float(N), time_index=ti)
    ref_u = oracle.hetcor_skeleton(Ch, np.ones((n, n), np.int32), np.full((n, n), N, np.float32), oracle.hetcor_threshold(alpha), lmax, ti)
    assert st_u.level == ref_u.level and np.array_equal(eng.adjacency(), ref_u.G)
    eng.close()
    del Ct, Nt
