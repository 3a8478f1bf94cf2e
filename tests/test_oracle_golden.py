"""CPU: the oracle reproduces every known-answer vector the reference's own tests hold."""
import numpy as np
import pytest


def test_threshold_kat(oracle, kat):
    k = kat["threshold"]
    th = oracle.threshold_array(k["n"], k["alpha"])
    assert abs(th[0] - k["th0"]) < k["tol"]
    # strictly decreasing degrees of freedom -> increasing thresholds
    assert np.all(np.diff(th) > 0)


def test_qnorm_against_scipy(oracle):
    from scipy.stats import norm

    # the path only ever asks for p = alpha/2 (lower tail); the upper tail loses bits in 1-p
    for p in [1e-12, 5e-9, 5e-5, 0.001, 0.02, 0.3, 0.5]:
        assert abs(oracle.qnorm(p) - norm.ppf(p)) <= 2e-15 * max(1.0, abs(norm.ppf(p))) + 1e-15
    for p in [0.9, 0.999]:
        assert abs(oracle.qnorm(p) - norm.ppf(p)) <= 1e-13


def test_skeleton_n10(oracle, kat):
    c = kat["cupc_n10"]
    Cm = np.array(c["C"], np.float32).reshape(c["n"], c["n"])
    Th = oracle.threshold_array(c["sample_size"], c["alpha"])
    r = oracle.skeleton(Cm, Th, c["max_level"])
    assert np.array_equal(r.G.ravel(), np.array(c["A"], np.int32))
    # pMax post-processing (cuPC-S.cu:424-442)
    assert np.all(np.diag(r.pmax) == 1)
    assert np.all(r.pmax[r.G == 1] == -100000)
    off = (r.G == 0) & ~np.eye(c["n"], dtype=bool)
    assert np.all(r.pmax[off] >= 0) and np.array_equal(r.pmax, r.pmax.T)


def test_hetcor_skeleton_n10(oracle, kat):
    c = kat["cupc_n10"]
    n = c["n"]
    Cm = np.array(c["C"], np.float32).reshape(n, n)
    N = np.full((n, n), c["sample_size"], np.float32)
    r = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), N, oracle.hetcor_threshold(c["alpha"]), c["max_level"],
                               np.zeros(n, np.int32))
    assert np.array_equal(r.G.ravel(), np.array(c["A"], np.int32))


def test_corr_bmt(oracle, kat):
    b = kat["bmt"]
    mxm, mxp, pxp = oracle.corr_pearson_npn(
        np.array(b["marker_vals"], np.uint8), np.array(b["phen_vals"], np.float32), b["num_markers"],
        b["num_individuals"], b["num_phen"], b["marker_mean"], b["marker_std"])
    assert np.allclose(mxm, b["exp_mxm"], atol=b["tol"], rtol=0)
    assert np.allclose(mxp, b["exp_mxp"], atol=b["tol"], rtol=0)
    assert np.allclose(pxp, b["exp_pxp"], atol=b["tol"], rtol=0)


def test_corr_bmt2(oracle, kat):
    b = kat["bmt2"]
    mxm, mxp, pxp = oracle.corr_pearson_npn(
        np.array(b["marker_vals"], np.uint8), np.array(b["phen_vals"], np.float32), b["num_markers"],
        b["num_individuals"], b["num_phen"], b["marker_mean"], b["marker_std"])
    assert np.allclose(mxm, b["exp_mxm_npn"], atol=b["tol"], rtol=0)
    assert np.allclose(mxp, b["exp_mxp_pearson"], atol=b["tol"], rtol=0)
    assert np.allclose(pxp, b["exp_pxp"], atol=b["tol"], rtol=0)


def test_phen_corr_with_nan(oracle, kat, golden_dir):
    import os

    k = kat["with_nan_phen"]
    ns, npn, data = oracle.load_phen(os.path.join(golden_dir, k["file"]))
    out = np.zeros(npn * (npn - 1) // 2, np.float32)
    oracle.lib().orc_phen_corr_pearson(data, ns, npn, out)
    assert np.allclose(out, k["exp_pxp"], atol=k["tol"], rtol=0)


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_subset_variables(oracle, kat, depth):
    p = kat["parent_set"]
    n = p["num_markers"] + p["num_phen"]
    got = oracle.subset_variables(np.array(p["adj"]), n, p["num_markers"], depth)
    assert list(got) == p["d%d" % depth]


def _run_cuskss_case(oracle, k, gd):
    import os

    _, pxp, ess_p = oracle.load_pxp(os.path.join(gd, k["pxp"]), sample_size=k["num_samples"])
    if k["trait_only"]:
        return oracle.cuskss_from_square(pxp, ess_p, pxp.shape[0], k["alpha"], k["max_level_one"], 0, k["depth"])
    mxm = oracle.load_mxm(os.path.join(gd, k["mxm"]))
    if k["merged"]:
        rows = np.fromfile(os.path.join(gd, k["marker_ixs"]), np.int32)
    else:
        b = oracle.read_blocks(os.path.join(gd, k["blocks"]))[k["block_index"]]
        rows = range(b[1] + b[3], b[2] + b[3] + 1)
    mxp, _ = oracle.load_mxp(os.path.join(gd, k["mxp"]), rows)
    sq, es = oracle.make_square_cuskss_inputs(mxm, mxp, pxp, k["num_samples"])
    return oracle.cuskss_from_square(sq, es, pxp.shape[0], k["alpha"], k["max_level_one"], k["max_level_two"], k["depth"])


@pytest.mark.parametrize("case", ["cuskss_trait_only", "cuskss_two_stage_merged", "cuskss_two_stage_block"])
def test_cuskss_end_to_end(oracle, kat, golden_dir, case):
    k = kat[case]
    r = _run_cuskss_case(oracle, k, golden_dir)
    assert list(r.G.ravel()) == k["exp_adj"]
    assert np.allclose(r.C.ravel(), k["exp_corr"], atol=k["tol"], rtol=0)
    if "exp_ixs" in k:
        assert list(r.new_to_old) == k["exp_ixs"]
    assert r.max_level == 14  # cli.cpp:58 passes ML


def test_ith_combination_is_lexicographic(oracle):
    from itertools import combinations

    for n, p in [(5, 3), (7, 2), (9, 5), (6, 1)]:
        combs = list(combinations(range(1, n + 1), p))
        assert oracle.binom(n, p) == len(combs)
        for r, c in enumerate(combs, start=1):
            assert tuple(oracle.ith_combination(n, p, r)) == c


def test_pinv_matches_float64(oracle):
    rng = np.random.default_rng(7)
    for l in range(2, 15):
        for _ in range(20):
            X = rng.standard_normal((l, 4 * l + 8))
            M = np.corrcoef(X).astype(np.float32)
            np.fill_diagonal(M, 1)
            got = oracle.pinv(M)
            want = np.linalg.inv(M.astype(np.float64))
            assert np.allclose(got, want, rtol=2e-3, atol=2e-4), l


def test_ci_test_matches_float64_partial_correlation(oracle, synth):
    Cm = synth.random_corr(24, seed=3, k=400)
    rng = np.random.default_rng(0)
    for l in range(0, 8):
        for _ in range(25):
            idx = rng.choice(24, size=l + 2, replace=False)
            x, y, S = int(idx[0]), int(idx[1]), np.sort(idx[2:]).astype(np.int32)
            rho, z = oracle.ci_test(Cm, x, y, S)
            sub = Cm[np.ix_(idx, idx)].astype(np.float64)
            P = np.linalg.inv(sub)
            want = -P[0, 1] / np.sqrt(P[0, 0] * P[1, 1])
            assert abs(rho - want) < 1e-5
            assert abs(z - abs(np.arctanh(want))) < 1e-5


def test_blocking_kat(oracle, golden_dir):
    """blocking_tests.cpp: block_chr(TEST_V, "1", 500) -> 17 exact blocks; hanning_smoothing of the first 1000 entries"""
    import json
    import os

    k = json.load(open(os.path.join(golden_dir, "..", "blocking_kat.json")))
    v = np.array(k["test_v"], np.float32)
    assert oracle.block_chr(v, k["max_block_size"]) == [tuple(b) for b in k["exp_blocks"]]
    sm = oracle.hanning_smoothing(v[:1000], k["smooth_window"])
    assert np.allclose(sm, np.array(k["test_v_smooth"]), rtol=0, atol=k["smooth_tol"])

