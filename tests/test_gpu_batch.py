"""GPU: batched execution of small LD blocks (the reference runs one block per process,
cli.cpp:507-512).  Engine level: several matrices on the diagonal of one allocation swept in ONE run
(cusk_run_skeleton_batch) must give, block by block, exactly what cusk_run_skeleton gives for the block alone and what
the oracle gives -- adjacency, separating sets, canonical test counts summed over the blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ML = 14


@pytest.fixture(scope="module")
def cg():
    import cigwas_amd

    return cigwas_amd


def _layout(sizes):
    lo, hi, base = [], [], 0
    for k in sizes:
        lo.append(base)
        hi.append(base + k)
        base += (k + 63) // 64 * 64
    return np.array(lo, np.int32), np.array(hi, np.int32), base


def _batch_matrix(mats, lo, n, fill):
    """blocks on the diagonal; everything else `fill` (NaN / garbage: must never be read)"""
    C = np.full((n, n), fill, np.float32)
    for Cm, b in zip(mats, lo):
        k = Cm.shape[0]
        C[b:b + k, b:b + k] = Cm
    return C


@pytest.mark.parametrize("maxlevel,fill", [(3, np.nan), (14, 7.5)])
def test_batched_run_equals_per_block_runs_and_oracle(cg, oracle, synth, maxlevel, fill):
    # unequal blocks: LD blocks, a dense random SEM, a hub graph that keeps its degree to level 14, a tiny one, one whose
    # size is an exact multiple of 64 (no padding rows behind it)
    mats = [synth.synth_corr_block(150, 5, N=2048, block_index=11), synth.random_corr(40, seed=3, k=240),
            synth.hub_corr(16, 3, seed=4), synth.synth_corr_block(123, 5, N=2048, block_index=12),
            synth.random_corr(6, seed=9, k=60), synth.synth_corr_block(300, 6, N=2048, block_index=13)]
    assert mats[3].shape[0] == 128
    Th = cg.threshold_array(2048, 1e-3)
    lo, hi, n = _layout([m.shape[0] for m in mats])
    e = cg.Engine(0)
    Cd = cg.DeviceArray(_batch_matrix(mats, lo, n, fill))
    st = e.run_skeleton_batch(Cd.ptr, n, lo, hi, Th, maxlevel)
    Gs = e.adjacency_blocks()
    x, y, lv, z, S = e.sepsets()
    canon = np.zeros(ML + 1, np.int64)
    tests = np.zeros(ML + 1, np.int64)
    e1 = cg.Engine(0)
    level_max = 0
    for b, Cm in enumerate(mats):
        k = Cm.shape[0]
        ref = oracle.skeleton(Cm, Th, maxlevel)
        assert np.array_equal(Gs[b], ref.G), b
        sel = (x >= lo[b]) & (x < hi[b])
        assert np.all((y[sel] >= lo[b]) & (y[sel] < hi[b]))
        dense = np.full((k, k, ML), -1, np.int32)
        Sb = S[sel].copy()
        Sb[Sb >= 0] -= lo[b]
        dense[x[sel] - lo[b], y[sel] - lo[b]] = Sb
        assert np.array_equal(dense, ref.sepset), b
        canon += ref.tests
        level_max = max(level_max, ref.level)
        # and the single-block engine run on the block alone
        C1 = cg.DeviceArray(Cm)
        s1 = e1.run_skeleton(C1.ptr, k, Th, maxlevel)
        assert np.array_equal(e1.adjacency(), Gs[b])
        tests[: len(s1.tests)] += np.array(s1.tests)
        C1.free()
    e1.close()
    assert st.level == level_max
    assert list(st.canonical_tests[: level_max + 1]) == [int(v) for v in canon[: level_max + 1]]
    assert st.tests[0] == canon[0]  # level 0: the pairs inside the blocks only
    if maxlevel == 14:
        assert st.levels_run == 15
    Cd.free()
    e.close()


def test_batch_of_one_and_bad_layouts(cg, oracle, synth):
    Cm = synth.synth_corr_block(200, 4, N=2048, block_index=5)
    k = Cm.shape[0]
    Th = cg.threshold_array(2048, 1e-3)
    e = cg.Engine(0)
    n = (k + 63) // 64 * 64
    Cd = cg.DeviceArray(_batch_matrix([Cm], [0], n, np.nan))
    e.run_skeleton_batch(Cd.ptr, n, [0], [k], Th, 3)
    assert np.array_equal(e.adjacency_blocks()[0], oracle.skeleton(Cm, Th, 3).G)
    with pytest.raises(RuntimeError, match="multiples of 64"):
        e.run_skeleton_batch(Cd.ptr, n, [3], [k], Th, 3)
    with pytest.raises(RuntimeError, match="multiples of 64"):
        e.run_skeleton_batch(Cd.ptr, n, [0, 64], [100, 128], Th, 3)
    Cd.free()
    e.close()


def test_gather_rows_places_submatrices(cg):
    rng = np.random.default_rng(3)
    n = 90
    M = rng.standard_normal((n, n)).astype(np.float32)
    Md = cg.DeviceArray(M)
    lists = [np.array([3, 7, 8, 40]), np.array([50, 51, 89]), np.array([0])]
    idx = np.concatenate(lists)
    row_src, row_k, row_first, row_out = [], [], [], []
    first, out = 0, 0
    for L in lists:
        for r in range(len(L)):
            row_src.append(L[r])
            row_k.append(len(L))
            row_first.append(first)
            row_out.append(out + r * len(L))
        first += len(L)
        out += len(L) ** 2
    e = cg.Engine(0)
    got = e.gather_rows(Md.ptr, n, idx, row_src, row_k, row_first, row_out, out_count=out)
    want = np.concatenate([M[np.ix_(L, L)].reshape(-1) for L in lists])
    assert np.array_equal(got, want)
    Md.free()
    e.close()
