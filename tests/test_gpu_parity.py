"""GPU parity: the HIP engine (through the C ABI) vs the CPU oracle on the same inputs.

Bars: adjacency, level counter and separation-set indices bit-exact; Fisher z / pMax
within 1e-6 (north_star); SNP x SNP correlations bit-exact (integer counts + identical
epilogue), SNP x trait and trait x trait within 1e-5 (the reference's own test tolerance,
tests/corr_tests.cpp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ML = 14


@pytest.fixture(scope="module")
def cg():
    import cigwas_amd

    return cigwas_amd


@pytest.fixture(scope="module")
def eng(cg):
    e = cg.Engine(0)
    yield e
    e.close()


def _dense_sepsets(n, x, y, S):
    d = np.full((n, n, ML), -1, np.int32)
    d[x, y] = S
    return d


def _check_skeleton(cg, eng, oracle, Cm, Th, maxlevel):
    n = Cm.shape[0]
    ref = oracle.skeleton(Cm, Th, maxlevel)
    Cd = cg.DeviceArray(Cm)
    st = eng.run_skeleton(Cd.ptr, n, Th, maxlevel)
    G = eng.adjacency()
    assert st.level == ref.level
    assert np.array_equal(G, ref.G)
    x, y, lv, z, S = eng.sepsets()
    assert np.array_equal(_dense_sepsets(n, x, y, S), ref.sepset)
    pm = eng.pmax(Cd.ptr)
    assert np.allclose(pm, ref.pmax, rtol=0, atol=1e-6)
    # the engine may evaluate more tests than the sequential schedule (parallel lanes), never fewer removals
    for l in range(1, ML + 1):
        assert st.tests[l] >= ref.tests[l] or st.tests[l] == 0 == ref.tests[l]
    # ... and the device computes, from the selected ranks, exactly the number of tests the sequential schedule runs
    assert list(st.canonical_tests[: ref.level + 1]) == [int(v) for v in ref.tests[: ref.level + 1]]
    Cd.free()
    return st, ref


def test_kat_n10_compat_abi(cg, oracle, kat):
    c = kat["cupc_n10"]
    n = c["n"]
    Cm = np.array(c["C"], np.float32).reshape(n, n)
    Th = cg.threshold_array(c["sample_size"], c["alpha"])
    G, level, pmax, sep = cg.Skeleton(Cm, Th, c["max_level"])
    assert list(G.ravel()) == c["A"]
    ref = oracle.skeleton(Cm, Th, c["max_level"])
    assert level == ref.level and np.array_equal(sep, ref.sepset)
    assert np.allclose(pmax, ref.pmax, rtol=0, atol=1e-6)
    N = np.full((n, n), c["sample_size"], np.float32)
    G2, level2 = cg.hetcor_skeleton(Cm, np.ones((n, n), np.int32), N, cg.hetcor_threshold(c["alpha"]), c["max_level"],
                                    np.zeros(n, np.int32))
    assert list(G2.ravel()) == c["A"]


@pytest.mark.parametrize("n,seed,maxlevel", [(12, 1, 14), (24, 2, 14), (40, 3, 6), (64, 4, 4), (90, 5, 3)])
def test_skeleton_random_sem(cg, eng, oracle, synth, n, seed, maxlevel):
    Cm = synth.random_corr(n, seed=seed, k=6 * n)
    Th = cg.threshold_array(6 * n, 0.01)
    _check_skeleton(cg, eng, oracle, Cm, Th, maxlevel)


def test_skeleton_dense_high_levels(cg, eng, oracle, synth):
    """few samples -> many surviving edges -> exercises the SVD inverse up to l = 8"""
    Cm = synth.random_corr(18, seed=11, k=400, strength=1.3)
    Th = cg.threshold_array(400, 0.2)
    st, ref = _check_skeleton(cg, eng, oracle, Cm, Th, 8)
    assert st.level >= 5


@pytest.mark.parametrize("nleaf,nhub,fast", [(16, 3, 1), (17, 3, 1), (15, 2, 0)])
def test_skeleton_deepest_levels(cg, oracle, synth, nleaf, nhub, fast):
    """hubs adjacent to every leaf keep their degree to the end: all of levels 1..14 run (float2 filter kernel up to
    level 8, scalar filter kernel with the incremental factorisation from level 9, SVD exact path for the rechecks
    and the winners' z), bit-exact against the oracle; fast = 0 is the exact arithmetic everywhere"""
    Cm = synth.hub_corr(nleaf, nhub, seed=4)
    Th = cg.threshold_array(20000, 0.01)
    e = cg.Engine(0)
    e.set_option("fast", fast)
    st, ref = _check_skeleton(cg, e, oracle, Cm, Th, 14)
    assert st.levels_run == 15 and ref.level == 15 and all(t > 0 for t in st.tests[9:15])
    # and through the hetcor engine (uniform effective sample size): adjacency only
    n = nleaf + nhub
    th = cg.hetcor_threshold(0.01)
    ti = np.zeros(n, np.int32)
    ref2 = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), 20000, np.float32), th, 14, ti)
    Cd = cg.DeviceArray(Cm)
    st2 = e.run_hetcor(Cd.ptr, n, th, 14, ess_uniform=20000.0, time_index=ti)
    assert st2.level == ref2.level and np.array_equal(e.adjacency(), ref2.G)
    Cd.free()
    e.close()


@pytest.mark.parametrize("m,p,maxlevel", [(400, 6, 3), (800, 10, 5)])
def test_skeleton_ld_block(cg, eng, oracle, synth, m, p, maxlevel):
    Cm = synth.synth_corr_block(m, p, N=4096, block_index=m)
    Th = cg.threshold_array(4096, 1e-4)
    st, ref = _check_skeleton(cg, eng, oracle, Cm, Th, maxlevel)
    assert sum(st.removed[1:]) > 0


@pytest.mark.parametrize("m,p,maxlevel", [(400, 6, 3), (800, 10, 5)])
def test_hetcor_uniform_ess(cg, eng, oracle, synth, m, p, maxlevel):
    Cm = synth.synth_corr_block(m, p, N=4096, block_index=m + 1)
    n = m + p
    th = cg.hetcor_threshold(1e-4)
    ti = np.zeros(n, np.int32)
    ti[m:] = 1
    ref = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), 4096, np.float32), th, maxlevel, ti)
    Cd = cg.DeviceArray(Cm)
    st = eng.run_hetcor(Cd.ptr, n, th, maxlevel, ess_uniform=4096.0, time_index=ti)
    assert st.level == ref.level
    assert np.array_equal(eng.adjacency(), ref.G)
    assert st.canonical_tests[1] == 0  # with a time index the canonical schedule is not counted on the device
    # without a time index the level-1 finaliser counts the tests of the canonical (sequential) schedule: the oracle's
    ref0 = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), 4096, np.float32), th, maxlevel, np.zeros(n, np.int32))
    st0 = eng.run_hetcor(Cd.ptr, n, th, maxlevel, ess_uniform=4096.0)
    assert st0.level == ref0.level and np.array_equal(eng.adjacency(), ref0.G)
    assert list(st0.canonical_tests[:2]) == [int(ref0.tests[0]), int(ref0.tests[1])] and st0.tests[1] >= ref0.tests[1]
    Cd.free()


def test_hetcor_heterogeneous_ess_time_index_and_ginit(cg, eng, oracle, synth):
    rng = np.random.default_rng(5)
    m, p = 300, 8
    n = m + p
    Cm = synth.synth_corr_block(m, p, N=4096, block_index=77)
    N = np.full((n, n), 4096, np.float32)
    e = rng.uniform(0.5, 1.0, (n, p)).astype(np.float32) * np.float32(4096)
    N[:, m:] = e
    N[m:, :] = e.T
    N[m:, m:] = np.maximum(N[m:, m:], N[m:, m:].T)
    N[3, m + 1] = N[m + 1, 3] = np.nan  # NA correlation -> r = 0, ESS NaN
    Cm[3, m + 1] = Cm[m + 1, 3] = 0.0
    ti = np.zeros(n, np.int32)
    ti[m:] = rng.integers(1, 4, p)
    G0 = np.ones((n, n), np.int32)
    G0[5, :] = 0
    G0[:, 5] = 0  # a variable removed by an earlier stage
    th = cg.hetcor_threshold(1e-4)
    ref = oracle.hetcor_skeleton(Cm, G0, N, th, 4, ti)
    G, level = cg.hetcor_skeleton(Cm, G0, N, th, 4, ti)  # compat ABI (host buffers)
    assert level == ref.level and np.array_equal(G, ref.G)
    Cd, Nd, Gd = cg.DeviceArray(Cm), cg.DeviceArray(N), cg.DeviceArray(G0)
    st = eng.run_hetcor(Cd.ptr, n, th, 4, N_dev=Nd.ptr, G_init_dev=Gd.ptr, time_index=ti)
    assert st.level == ref.level and np.array_equal(eng.adjacency(), ref.G)
    for d in (Cd, Nd, Gd):
        d.free()


def test_hub_row_beyond_lds(cg, eng, oracle):
    """one trait correlated with 260 markers: degree > 191 takes the unstaged path"""
    rng = np.random.default_rng(9)
    k, N = 260, 20000
    X = rng.standard_normal((k, N))
    y = 0.12 * X.sum(0) + rng.standard_normal(N)
    z = 0.5 * y + rng.standard_normal(N)
    Cm = np.corrcoef(np.vstack([X, y, z])).astype(np.float32)
    Cm = np.triu(Cm, 1) + np.triu(Cm, 1).T
    np.fill_diagonal(Cm, 1)
    Cm = np.ascontiguousarray(Cm, np.float32)
    Th = cg.threshold_array(N, 0.01)
    st, ref = _check_skeleton(cg, eng, oracle, Cm, Th, 2)
    assert st.max_degree[1] > 191
    n = Cm.shape[0]
    th = cg.hetcor_threshold(0.01)
    ref2 = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), N, np.float32), th, 2, np.zeros(n, np.int32))
    Cd = cg.DeviceArray(Cm)
    eng.run_hetcor(Cd.ptr, n, th, 2, ess_uniform=float(N))
    assert np.array_equal(eng.adjacency(), ref2.G)
    Cd.free()


def test_edge_cases(cg, eng, oracle):
    # n = 2, no edge survives level 0; and an identity matrix
    for Cm in [np.array([[1, 0.001], [0.001, 1]], np.float32), np.eye(5, dtype=np.float32)]:
        Th = cg.threshold_array(1000, 0.01)
        G, level, pmax, sep = cg.Skeleton(Cm, Th, 3)
        ref = oracle.skeleton(Cm, Th, 3)
        assert np.array_equal(G, ref.G) and level == ref.level
        assert np.allclose(pmax, ref.pmax, atol=1e-6) and np.array_equal(sep, ref.sepset)
    # perfectly collinear pair: singular conditioning sets give NaN z and keep the edge
    rng = np.random.default_rng(2)
    X = rng.standard_normal((6, 200))
    X[1] = X[0]
    Cm = np.corrcoef(X).astype(np.float32)
    np.fill_diagonal(Cm, 1)
    Cm = np.ascontiguousarray((Cm + Cm.T) / 2, np.float32)
    Th = cg.threshold_array(200, 0.3)
    G, level, pmax, sep = cg.Skeleton(Cm, Th, 4)
    ref = oracle.skeleton(Cm, Th, 4)
    assert np.array_equal(G, ref.G) and level == ref.level and np.array_equal(sep, ref.sepset)
    # maxlevel 0
    G, level, _, _ = cg.Skeleton(Cm, Th, 0)
    ref = oracle.skeleton(Cm, Th, 0)
    assert np.array_equal(G, ref.G) and level == ref.level


@pytest.mark.parametrize("n", [257, 700, 1100])
def test_level0_guard_band_and_unusual_values(cg, eng, oracle, n):
    """Level 0 alone (cal_Indepl0, cuPC-S.cu:458-484) on a matrix whose elements crowd the decision boundary: the kernel
    settles |c| against tanh(th) outside a guard band and leaves the band, |c| > 1 (where 0.5 |log |(1+c)/(1-c)|| comes
    back down: c = 100 is "independent") and NaN to the reference's arithmetic.  n covers border-only tilings (257) and
    tilings with interior tiles (700, 1100), whose code path carries no index clamps."""
    rng = np.random.default_rng(n)
    Th = cg.threshold_array(1000, 0.01)
    t = float(np.tanh(Th[0]))
    Cm = rng.uniform(-0.3, 0.3, (n, n)).astype(np.float32)
    iu = np.triu_indices(n, 1)
    k = len(iu[0])
    v = Cm[iu]
    # a third of the pairs within 1e-3 (relative) of the boundary, a few hundred within a few ulps of it
    near = rng.random(k) < 0.33
    v[near] = (t * (1 + rng.uniform(-1e-3, 1e-3, near.sum())) * rng.choice([-1, 1], near.sum())).astype(np.float32)
    ulp = rng.choice(k, 400, replace=False)
    steps = rng.integers(-6, 7, 400)
    base = np.full(400, t, np.float32)
    for _ in range(6):
        base = np.where(steps > 0, np.nextafter(base, np.float32(1)), np.where(steps < 0, np.nextafter(base, np.float32(0)), base))
        steps = steps - np.sign(steps)
    v[ulp] = base * rng.choice([-1, 1], 400).astype(np.float32)
    odd = rng.choice(k, 300, replace=False)
    v[odd] = rng.choice(np.array([1, -1, 1.0000001, -1.0000001, 1.5, -3, 100, -1e6, np.nan, np.inf, -np.inf, 0, -0.0], np.float32), 300)
    Cm[iu] = v
    Cm.T[iu] = v
    np.fill_diagonal(Cm, 1)
    Cm = np.ascontiguousarray(Cm)
    ref = oracle.skeleton(Cm, Th, 0)
    Cd = cg.DeviceArray(Cm)
    st = eng.run_skeleton(Cd.ptr, n, Th, 0)
    G = eng.adjacency()
    Cd.free()
    assert st.level == ref.level
    assert np.array_equal(G, ref.G)
    assert 0 < int(G.sum()) < n * (n - 1)


@pytest.mark.parametrize("mode", ["skeleton", "hetcor", "het"])
def test_fast_filter_never_contradicts_exact_path(cg, oracle, synth, mode):
    """validate mode evaluates BOTH paths for every certified verdict: zero contradictions,
    and the exact-only engine (fast = 0) gives the same answer as the filtered one."""
    e = cg.Engine(0)
    e.set_option("validate", 1)
    rng = np.random.default_rng(3)
    mats = [synth.synth_corr_block(500, 8, N=4096, block_index=31), synth.random_corr(40, seed=21, k=300, strength=1.2)]
    for Cm, N, alpha, ml in [(mats[0], 4096, 1e-4, 5), (mats[1], 300, 0.1, 7)]:
        n = Cm.shape[0]
        Cd = cg.DeviceArray(Cm)
        if mode == "skeleton":
            Th = cg.threshold_array(N, alpha)
            st = e.run_skeleton(Cd.ptr, n, Th, ml)
            ref = oracle.skeleton(Cm, Th, ml)
            G = e.adjacency()
            e.set_option("fast", 0)
            st0 = e.run_skeleton(Cd.ptr, n, Th, ml)
            e.set_option("fast", 1)
            assert np.array_equal(e.adjacency(), G)
        else:
            th = cg.hetcor_threshold(alpha)
            Nm = np.full((n, n), N, np.float32)
            if mode == "het":
                Nm = (Nm * rng.uniform(0.6, 1.0, (n, n))).astype(np.float32)
                Nm = np.maximum(Nm, Nm.T)
            Nd = cg.DeviceArray(Nm)
            st = e.run_hetcor(Cd.ptr, n, th, ml, N_dev=Nd.ptr if mode == "het" else None, ess_uniform=float(N))
            ref = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), Nm, th, ml, np.zeros(n, np.int32))
            G = e.adjacency()
            Nd.free()
        assert st.violations == 0 and st.exact_fallbacks == 0
        assert np.array_equal(G, ref.G) and st.level == ref.level
        assert sum(st.rechecks) < 0.05 * max(1, sum(st.tests[2:])) + 1000
        Cd.free()
    e.close()


def test_asymmetric_matrix_and_generic_level1(cg, eng, oracle, synth):
    """the reference reads C[X,Y], C[X,S], C[Y,S], C[S_a,S_b] (a<b) at exactly those positions;
    an asymmetric matrix must take the generic level-1 kernel and still match the oracle"""
    Cm = synth.synth_corr_block(300, 6, N=4096, block_index=5).copy()
    rng = np.random.default_rng(1)
    il = np.tril_indices(Cm.shape[0], -1)
    Cm[il] = (Cm[il] + rng.normal(0, 2e-3, len(il[0]))).astype(np.float32)
    Th = cg.threshold_array(4096, 1e-4)
    _check_skeleton(cg, eng, oracle, Cm, Th, 4)
    # symmetric matrix through the generic level-1 kernel (pair kernel disabled)
    Cs = synth.synth_corr_block(300, 6, N=4096, block_index=6)
    eng.set_option("pair", 0)
    try:
        _check_skeleton(cg, eng, oracle, Cs, Th, 3)
    finally:
        eng.set_option("pair", 1)


def test_recheck_queue_overflow_falls_back_to_exact(cg, oracle, synth):
    e = cg.Engine(0)
    e.set_option("queue_capacity", 1)
    rng = np.random.default_rng(8)
    X = rng.standard_normal((30, 60))
    X[1] = X[0] + 0.05 * rng.standard_normal(60)  # nearly collinear -> ill-conditioned sets -> many rechecks
    Cm = np.corrcoef(X).astype(np.float32)
    Cm = np.ascontiguousarray(np.triu(Cm, 1) + np.triu(Cm, 1).T + np.eye(30, dtype=np.float32))
    Th = cg.threshold_array(60, 0.3)
    Cd = cg.DeviceArray(Cm)
    st = e.run_skeleton(Cd.ptr, 30, Th, 5)
    ref = oracle.skeleton(Cm, Th, 5)
    assert st.exact_fallbacks > 0
    assert np.array_equal(e.adjacency(), ref.G) and st.level == ref.level
    x, y, lv, z, S = e.sepsets()
    assert np.array_equal(_dense_sepsets(30, x, y, S), ref.sepset)
    Cd.free()
    e.close()


# ---------------------------------------------------------------- correlation build
def test_corr_kat_bmt(cg, kat):
    for key, mm, mp, pp in [("bmt", "exp_mxm", "exp_mxp", "exp_pxp"), ("bmt2", "exp_mxm_npn", "exp_mxp_pearson", "exp_pxp")]:
        b = kat[key]
        mxm, mxp, pxp = cg.cu_corr_pearson_npn(np.array(b["marker_vals"], np.uint8), np.array(b["phen_vals"], np.float32),
                                               b["num_markers"], b["num_individuals"], b["num_phen"], b["marker_mean"],
                                               b["marker_std"])
        assert np.allclose(mxm, b[mm], atol=b["tol"], rtol=0)
        assert np.allclose(mxp, b[mp], atol=b["tol"], rtol=0)
        assert np.allclose(pxp, b[pp], atol=b["tol"], rtol=0)
        mxp2 = cg.cu_marker_phen_corr_pearson(np.array(b["marker_vals"], np.uint8), np.array(b["phen_vals"], np.float32),
                                              b["num_markers"], b["num_individuals"], b["num_phen"], b["marker_mean"],
                                              b["marker_std"])
        assert np.allclose(mxp2, b[mp], atol=b["tol"], rtol=0)


@pytest.mark.parametrize("m,N,p", [(97, 1001, 3), (300, 2048, 7), (64, 130, 1), (150, 640, 45), (33, 64, 21), (70, 4096, 22)])
def test_corr_build_vs_oracle(cg, eng, oracle, synth, m, N, p):
    bed, phen, means, stds, G = synth.synth_bed_block(m, N, p, block_index=m, miss=0.01)
    phen = phen.copy()
    phen[::97] = np.nan  # missing phenotypes
    o_mxm, o_mxp, o_pxp = oracle.corr_pearson_npn(bed, phen, m, N, p, means, stds)
    mxm, mxp, pxp = cg.cu_corr_pearson_npn(bed, phen, m, N, p, means, stds)
    assert np.array_equal(mxm, o_mxm, equal_nan=True)  # exact counts, same epilogue
    assert np.allclose(mxp, o_mxp, atol=1e-5, rtol=0)
    assert np.allclose(pxp, o_pxp, atol=1e-5, rtol=0)
    n = m + p
    Cd = cg.DeviceArray(nbytes=4 * n * n)
    mxp3 = eng.corr_build(bed, phen, m, N, p, means, stds, Cd.ptr, want_mxp=True)
    sq = Cd.download(np.float32, (n, n))
    want = oracle.square_from_cusk_corrs(mxm, mxp, pxp, m, p)
    assert np.array_equal(sq, want, equal_nan=True) and np.array_equal(mxp3, mxp)
    Cd.free()


@pytest.mark.parametrize("m,N", [(200, 1500), (130, 4096), (70, 777)])
def test_corr_mfma_and_popcount_kernels_agree(cg, oracle, synth, m, N):
    """the FP4 MFMA contingency GEMM (default), the int8 MFMA one and the bit-plane popcount kernel give identical
    SNP x SNP matrices, equal to the oracle's (N with and without a partial last K block / byte)"""
    p = 3
    bed, phen, means, stds, G = synth.synth_bed_block(m, N, p, block_index=4, miss=0.02)
    n = m + p
    outs = []
    for opts in ({"corr_fp4": 1}, {"corr_fp4": 0}, {"corr_popcount": 1}):
        e = cg.Engine(0)
        for k, v in opts.items():
            e.set_option(k, v)
        Cd = cg.DeviceArray(nbytes=4 * n * n)
        e.corr_build(bed, phen, m, N, p, means, stds, Cd.ptr)
        outs.append(Cd.download(np.float32, (n, n)))
        Cd.free()
        e.close()
    assert np.array_equal(outs[0][:m, :m], outs[1][:m, :m], equal_nan=True)
    assert np.array_equal(outs[0][:m, :m], outs[2][:m, :m], equal_nan=True)
    assert np.array_equal(outs[0], outs[1], equal_nan=True)
    assert np.allclose(outs[0], outs[2], atol=1e-6, rtol=0, equal_nan=True)  # SNP x trait: different summation orders
    o_mxm, _, _ = oracle.corr_pearson_npn(bed, phen, m, N, p, means, stds)
    iu = np.triu_indices(m, 1)
    assert np.array_equal(outs[0][:m, :m][iu], o_mxm, equal_nan=True)


@pytest.mark.parametrize("m,N,p", [(200, 2048, 20), (95, 1000, 5), (64, 4160, 43)])
def test_corr_snp_trait_bf16_split_against_f32_chain(cg, oracle, synth, m, N, p):
    """SNP x trait on the bf16 matrix pipe (every trait value split exactly into three bf16 pieces: the default) against the
    f32 matrix-instruction form of rounds 1-2 (option corr_mxp_f32) and the oracle: the products are exact in both, only the
    order of the f32 additions differs.  Traits with NaN; N with and without whole 64-individual requests; more than 21
    traits (three launches)."""
    bed, phen, means, stds, G = synth.synth_bed_block(m, N, p, block_index=m + 1, miss=0.03)
    phen = phen.copy()
    phen[::53] = np.nan
    phen[2::307] *= 1e-3  # small values next to ordinary ones: the low pieces matter
    n = m + p
    outs = []
    for opts in ({}, {"corr_mxp_f32": 1}):
        e = cg.Engine(0)
        for k, v in opts.items():
            e.set_option(k, v)
        Cd = cg.DeviceArray(nbytes=4 * n * n)
        mxp = e.corr_build(bed, phen, m, N, p, means, stds, Cd.ptr, want_mxp=True)
        sq = Cd.download(np.float32, (n, n))
        assert np.array_equal(sq[:m, m:], mxp.reshape(m, p), equal_nan=True) and np.array_equal(sq[m:, :m], sq[:m, m:].T, equal_nan=True)
        outs.append(sq)
        Cd.free()
        e.close()
    assert np.array_equal(outs[0][:m, :m], outs[1][:m, :m], equal_nan=True)
    assert np.allclose(outs[0], outs[1], atol=2e-6, rtol=0, equal_nan=True)
    _, o_mxp, _ = oracle.corr_pearson_npn(bed, phen, m, N, p, means, stds)
    assert np.allclose(outs[0][:m, m:].ravel(), np.asarray(o_mxp).ravel(), atol=1e-5, rtol=0, equal_nan=True)


def test_randomised_stress_against_oracle(cg, eng, oracle, synth):
    """120 small random problems over sample size (down to barely more samples than variables: ill-conditioned
    conditioning sets, NaN z, heavy use of the recheck path), density, alpha and max level; both engines."""
    rng = np.random.default_rng(2025)
    checked_levels = 0
    for it in range(120):
        n = int(rng.integers(4, 42))
        k = int(rng.integers(n + 2, 12 * n))
        alpha = float(rng.choice([0.01, 0.05, 0.2, 0.4]))
        maxlevel = int(rng.integers(0, 9))
        Cm = synth.random_corr(n, seed=1000 + it, k=k, strength=float(rng.uniform(0.6, 1.4)))
        if it % 7 == 0:  # duplicate a variable: exactly singular conditioning sets
            Cm[:, 1] = Cm[:, 0]
            Cm[1, :] = Cm[0, :]
            Cm[0, 1] = Cm[1, 0] = np.float32(0.999999)
            np.fill_diagonal(Cm, 1)
        Th = cg.threshold_array(max(k, 20), alpha)
        ref = oracle.skeleton(Cm, Th, maxlevel)
        Cd = cg.DeviceArray(Cm)
        st = eng.run_skeleton(Cd.ptr, n, Th, maxlevel)
        assert st.level == ref.level, (it, n, k, alpha, maxlevel)
        assert np.array_equal(eng.adjacency(), ref.G), (it, n, k, alpha, maxlevel)
        x, y, lv, z, S = eng.sepsets()
        assert np.array_equal(_dense_sepsets(n, x, y, S), ref.sepset), (it, n, k, alpha, maxlevel)
        th = cg.hetcor_threshold(alpha)
        ti = np.zeros(n, np.int32)
        ti[n // 2:] = rng.integers(0, 3, n - n // 2)
        ess = float(max(k, 20))
        ref2 = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), ess, np.float32), th, maxlevel, ti)
        st2 = eng.run_hetcor(Cd.ptr, n, th, maxlevel, ess_uniform=ess, time_index=ti)
        assert st2.level == ref2.level and np.array_equal(eng.adjacency(), ref2.G), (it, n, k, alpha, maxlevel)
        checked_levels += st.levels_run
        Cd.free()
    assert checked_levels > 300


@pytest.mark.parametrize("opts", [
    {"lookahead": 0}, {"lookahead": 3}, {"timing": 0}, {"timing": 2},
    {"item_capacity": 1024},                      # more work items than the buffers hold: grown, level taken up again
    {"item_capacity": 1024, "queue_capacity": 16, "lookahead": 2},  # both resume paths in one run
    {"item_capacity": 1024, "rows": 0},           # ... with the pair kernel's work items at level 1
])
def test_enqueue_ahead_loop_options_and_resume_paths(cg, oracle, synth, opts):
    """the level loop runs ahead of the device; whatever the distance, the timing mode, or how often a level has to be
    taken up again (work-item buffers too small, recheck queue overflow), the result is the oracle's"""
    Cm = synth.synth_corr_block(1400, 8, N=8192, block_index=21)
    Th = cg.threshold_array(8192, 1e-4)
    e = cg.Engine(0)
    for k, v in opts.items():
        e.set_option(k, v)
    st, ref = _check_skeleton(cg, e, oracle, Cm, Th, 5)
    assert st.levels_run == ref.level and sum(st.tests) > 0
    if "queue_capacity" in opts:
        assert st.exact_fallbacks >= 1
    # a second run on the same engine (buffers grown, gate records of the previous run in place) gives the same
    st2, _ = _check_skeleton(cg, e, oracle, Cm, Th, 5)
    assert st2.canonical_tests == st.canonical_tests
    # the hetcor engine through the same loop
    th = cg.hetcor_threshold(1e-4)
    Cd = cg.DeviceArray(Cm)
    n = Cm.shape[0]
    ti = np.zeros(n, np.int32)
    ti[1400:] = 1
    sth = e.run_hetcor(Cd.ptr, n, th, 4, ess_uniform=8192.0, time_index=ti)
    refh = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), 8192, np.float32), th, 4, ti)
    assert np.array_equal(e.adjacency(), refh.G) and sth.level == refh.level
    Cd.free()
    e.close()
