"""GPU parity of the union-major sweep (csrc/sweep_tmaj.hip: deep levels enumerated by T = S + Y, one inverse per
l + 1 tests) against the CPU oracle.  By default it runs from level 6; option tmaj_min_level = 2 puts every level
>= 2 of these small cases through it, so adjacency, level counter, separating sets, canonical test counts and pMax
are checked bit for bit where the oracle finishes in seconds.  `validate` checks every certified verdict of the kernel
against a double-precision evaluation on the device (cusk_stats.violations must stay 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ML = 14


@pytest.fixture(scope="module")
def cg():
    import cigwas_amd

    return cigwas_amd


def _dense_sepsets(n, x, y, S):
    d = np.full((n, n, ML), -1, np.int32)
    d[x, y] = S
    return d


def _check(cg, e, oracle, Cm, Th, maxlevel):
    n = Cm.shape[0]
    ref = oracle.skeleton(Cm, Th, maxlevel)
    Cd = cg.DeviceArray(Cm)
    st = e.run_skeleton(Cd.ptr, n, Th, maxlevel)
    assert st.level == ref.level
    assert np.array_equal(e.adjacency(), ref.G)
    x, y, lv, z, S = e.sepsets()
    assert np.array_equal(_dense_sepsets(n, x, y, S), ref.sepset)
    assert np.allclose(e.pmax(Cd.ptr), ref.pmax, rtol=0, atol=1e-6)
    assert list(st.canonical_tests[: ref.level + 1]) == [int(v) for v in ref.tests[: ref.level + 1]]
    # every (S, Y) pair of a union-major level is evaluated exactly once: never fewer tests than the sequential schedule
    for l in range(2, ref.level):
        assert st.tests[l] >= ref.tests[l]
    Cd.free()
    return st, ref


@pytest.mark.parametrize("validate", [0, 1])
@pytest.mark.parametrize("m,p,maxlevel", [(400, 6, 4), (800, 10, 5)])
def test_tmaj_ld_block_all_levels(cg, oracle, synth, m, p, maxlevel, validate):
    e = cg.Engine(0)
    e.set_option("tmaj_min_level", 2)
    e.set_option("validate", validate)
    Cm = synth.synth_corr_block(m, p, N=4096, block_index=m + 7)
    Th = cg.threshold_array(4096, 1e-4)
    st, ref = _check(cg, e, oracle, Cm, Th, maxlevel)
    assert sum(st.removed[2:]) > 0 and st.violations == 0 and st.exact_fallbacks == 0
    e.close()


@pytest.mark.parametrize("nleaf,nhub,tmin", [(16, 3, 2), (17, 3, 6), (18, 2, 2)])
def test_tmaj_deepest_levels(cg, oracle, synth, nleaf, nhub, tmin):
    """hubs keep their degree to the end: levels up to 14 through the union-major kernel (from level 2, and from the
    default level 6 with the set-major kernels below it), validated against double precision on the device"""
    Cm = synth.hub_corr(nleaf, nhub, seed=5)
    Th = cg.threshold_array(20000, 0.01)
    e = cg.Engine(0)
    e.set_option("tmaj_min_level", tmin)
    e.set_option("validate", 1)
    st, ref = _check(cg, e, oracle, Cm, Th, 14)
    assert st.levels_run == 15 and all(t > 0 for t in st.tests[9:15]) and st.violations == 0
    e.set_option("validate", 0)
    _check(cg, e, oracle, Cm, Th, 14)
    # the same levels through the set-major kernels give the same graph and sets (cross-check of the two enumerations)
    e.set_option("tmaj_min_level", 99)
    _check(cg, e, oracle, Cm, Th, 14)
    e.close()


def test_tmaj_random_dense(cg, oracle, synth):
    """dense random correlations at a loose threshold: many removals at every level, ill-conditioned sets included
    (validated by sampling: every fourth union of a lane, the form used for full-size runs)"""
    e = cg.Engine(0)
    e.set_option("tmaj_min_level", 2)
    e.set_option("validate", 1)
    e.set_option("tmaj_validate_stride", 4)
    Cm = synth.random_corr(40, seed=23, k=300, strength=1.2)
    Th = cg.threshold_array(300, 0.1)
    st, ref = _check(cg, e, oracle, Cm, Th, 8)
    assert st.violations == 0
    e.close()


def test_tmaj_hetcor_uniform_ess_time_index(cg, oracle, synth):
    e = cg.Engine(0)
    e.set_option("tmaj_min_level", 2)
    m, p, maxlevel = 600, 8, 5
    Cm = synth.synth_corr_block(m, p, N=4096, block_index=91)
    n = m + p
    th = cg.hetcor_threshold(1e-4)
    ti = np.zeros(n, np.int32)
    ti[m:] = 1
    ti[m + p // 2:] = 2
    ref = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), 4096, np.float32), th, maxlevel, ti)
    Cd = cg.DeviceArray(Cm)
    st = e.run_hetcor(Cd.ptr, n, th, maxlevel, ess_uniform=4096.0, time_index=ti)
    assert st.level == ref.level and np.array_equal(e.adjacency(), ref.G)
    Cd.free()
    e.close()


def test_tmaj_recheck_queue_overflow_replans_on_exact_path(cg, oracle):
    """a union-major level whose recheck queue overflows is planned again (its work items count unions) and swept by
    the exact kernels"""
    e = cg.Engine(0)
    e.set_option("tmaj_min_level", 2)
    e.set_option("queue_capacity", 1)
    rng = np.random.default_rng(8)
    X = rng.standard_normal((30, 60))
    X[1] = X[0] + 0.05 * rng.standard_normal(60)  # nearly collinear -> ill-conditioned sets -> many rechecks
    Cm = np.corrcoef(X).astype(np.float32)
    Cm = np.ascontiguousarray(np.triu(Cm, 1) + np.triu(Cm, 1).T + np.eye(30, dtype=np.float32))
    Th = cg.threshold_array(60, 0.3)
    st, ref = _check(cg, e, oracle, Cm, Th, 5)
    assert st.exact_fallbacks > 0
    e.close()


@pytest.mark.parametrize("tmin", [2, 99])
def test_unstaged_kernels_every_row_through_l2(cg, oracle, synth, tmin):
    """max_staged_classes = 0 sends every row to the last degree class, whose kernels read their operands through L2
    instead of an LDS copy (the path of hubs with more than 191 neighbours): union-major and set-major sweeps"""
    e = cg.Engine(0)
    e.set_option("max_staged_classes", 0)
    e.set_option("tmaj_min_level", tmin)
    Cm = synth.synth_corr_block(500, 8, N=4096, block_index=77)
    Th = cg.threshold_array(4096, 1e-4)
    st, ref = _check(cg, e, oracle, Cm, Th, 5)
    assert sum(st.removed[2:]) > 0
    Cm = synth.hub_corr(15, 2, seed=6)
    _check(cg, e, oracle, Cm, cg.threshold_array(20000, 0.01), 14)
    e.close()
