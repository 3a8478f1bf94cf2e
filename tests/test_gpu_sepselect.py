"""GPU: the sepselect path through the C ABI (`cusk_sepselect_greedy`) and its Python host
mirror, against (1) the files the reference itself wrote for the golden merged skeletons and (2) the numpy
oracle on larger seeded inputs.  Sets, triples, PAG marks and every text file must be identical."""
import os

import numpy as np
import pytest

from test_sepselect_oracle import check_outputs, load_cases, materialise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["small", "prior", "collinear", "wide", "dense_traits"])
def test_orient_v_structs_files_equal_reference(name, tmp_path):
    from cigwas_amd import sepselect as SS

    case = load_cases()[name]
    stem, prior = materialise(case, str(tmp_path))
    res = SS.orient_v_structures_merged(stem, case["alpha"], case["num_samples"], orientation_prior_file=prior)
    assert len(res.min_sepsets) == case["pairs_with_minimum"]
    ostem = os.path.join(str(tmp_path), "max_sep_min_pc")
    res.to_file(ostem)
    check_outputs(case, ostem)


def test_sepselect_merged_writes_all_but_pag(tmp_path):
    from cigwas_amd import sepselect as SS

    case = load_cases()["small"]
    stem, _ = materialise(case, str(tmp_path))
    res = SS.sepselect_merged(stem, case["alpha"], case["num_samples"])
    ostem = os.path.join(str(tmp_path), "max_sep_min_pc")
    res.to_file(ostem)
    exp = case["output"]
    assert open(ostem + ".ssm").read() == exp["ssm"] and open(ostem + ".mdim").read() == exp["mdim"]
    assert np.fromfile(ostem + ".ut", dtype=np.int32).tolist() == exp["ut"]
    assert np.fromfile(ostem + ".atr", dtype=np.int32).tolist() == exp["atr"]
    assert not os.path.exists(ostem + "_spm.mtx")


@pytest.mark.parametrize("seed,p,m,alpha,N", [(101, 20, 150, 1e-4, 30000), (102, 36, 60, 1e-3, 8000)])
def test_sets_equal_oracle_on_larger_graphs(seed, p, m, alpha, N, tmp_path, synth):
    import scipy.sparse as sp
    from scipy.io import mmwrite

    from cigwas_amd import sepselect as SS
    from oracle import sepselect_oracle as SO

    adj, corr, ixs, _ = synth.merged_skeleton(seed, p, m)
    stem = os.path.join(str(tmp_path), "all_merged")
    mmwrite(stem + "_sam.mtx", sp.coo_matrix(adj.astype(np.int32)))
    mmwrite(stem + "_scm.mtx", sp.coo_matrix(corr))
    open(stem + ".mdim", "w").write(f"{p + m}\t{p}\t3\n")
    ixs.tofile(stem + ".ixs")
    exp = SO.run(stem, alpha, N)
    res = SS.orient_v_structures_merged(stem, alpha, N)
    assert {k: [int(v) for v in s] for k, s in exp["max_sepsets"].items()} == res.max_sepsets
    assert set(exp["min_sepsets"]) == set(res.min_sepsets)
    assert np.array_equal(exp["rel"], res.get_rfci_relevant_unshielded_triples())
    assert np.array_equal(exp["ambiguous"], res.ambiguous_triples)
    assert np.array_equal(exp["pag"], res.pag)


def test_long_candidate_lists_every_storage_class(oracle):
    """candidate counts straddling the LDS classes (8/16/32/64/84) and the global-work-space class (> 84)"""
    import cigwas_amd as cg
    from cigwas_amd.sepselect import alpha_thr
    from oracle import sepselect_oracle as SO

    rng = np.random.default_rng(7)
    p, extra = 140, 6
    n = p + extra
    F = rng.normal(size=(n, 12)) * 0.45
    cov = F @ F.T + np.eye(n)
    sd = np.sqrt(np.diag(cov))
    corr = cov / np.outer(sd, sd)
    corr = 0.5 * (corr + corr.T)
    np.fill_diagonal(corr, 1.0)
    alpha, N = 1e-3, 400
    sizes = [0, 1, 8, 9, 17, 33, 65, 84, 85, 100, 129, 140]
    pair_i, pair_j, cands = [], [], []
    for k, t in enumerate(sizes):
        i, j = p + (k % extra), p + ((k + 1) % extra)
        pool = set(rng.choice(p, size=t, replace=False).tolist())
        pair_i.append(i)
        pair_j.append(j)
        cands.append(list(pool))
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.int64)
    cand = np.array([v for c in cands for v in c], dtype=np.int32)
    thr = alpha_thr(alpha, N, np.arange(p + 1, dtype=np.float64))
    eng = cg.Engine(0)
    sel, sel_len, flags, ms = eng.sepselect_greedy(corr[:, :p], pair_i, pair_j, corr[pair_i, pair_j], off, cand, thr)
    assert np.all(flags >> 8 == 0) and ms > 0
    for k in range(len(sizes)):
        chosen, seen = SO.greedy_pair(corr, pair_i[k], pair_j[k], set(cands[k]), alpha, N)
        assert sel[off[k]:off[k] + sel_len[k]].tolist() == [int(v) for v in chosen], sizes[k]
        assert bool(flags[k] & 1) == seen


def test_many_long_lists_run_in_work_space_batches():
    """pairs beyond the LDS classes share a bounded HBM work space: many of them give the same answer as one"""
    import cigwas_amd as cg
    from cigwas_amd.sepselect import alpha_thr

    rng = np.random.default_rng(3)
    p = 96
    n = p + 2
    F = rng.normal(size=(n, 10)) * 0.4
    cov = F @ F.T + np.eye(n)
    sd = np.sqrt(np.diag(cov))
    corr = 0.5 * (cov / np.outer(sd, sd) + (cov / np.outer(sd, sd)).T)
    np.fill_diagonal(corr, 1.0)
    thr = alpha_thr(1e-3, 300, np.arange(p + 1, dtype=np.float64))
    cand1 = np.arange(p, dtype=np.int32)
    eng = cg.Engine(0)
    one = eng.sepselect_greedy(corr[:, :p], [p], [p + 1], corr[[p], [p + 1]], [0, p], cand1, thr)
    k = 300
    eng.set_option("sepselect_ws_bytes", 1 << 20)  # ~13 pairs per batch
    many = eng.sepselect_greedy(corr[:, :p], [p] * k, [p + 1] * k, np.repeat(corr[p, p + 1], k),
                                np.arange(k + 1, dtype=np.int64) * p, np.tile(cand1, k), thr)
    assert np.all(many[1] == one[1][0]) and np.all(many[2] == one[2][0])
    assert np.array_equal(many[0].reshape(k, p), np.tile(one[0], (k, 1)))


def test_bad_arguments_are_refused():
    import cigwas_amd as cg

    eng = cg.Engine(0)
    corr = np.eye(4)
    with pytest.raises(Exception):  # candidate 7 is not a trait index
        eng.sepselect_greedy(corr[:, :2], [2], [3], [0.0], [0, 1], [7], [1.0, 1.0])
    with pytest.raises(Exception):  # threshold table too short
        eng.sepselect_greedy(corr[:, :2], [2], [3], [0.0], [0, 2], [0, 1], [1.0])


def test_singular_submatrix_exits_like_reference(tmp_path):
    import cigwas_amd as cg

    eng = cg.Engine(0)
    corr = np.eye(5)
    corr[0, 1] = corr[1, 0] = 1.0  # traits 0 and 1 identical: the first round's pivot is fine, the second is 0
    _, _, flags, _ = eng.sepselect_greedy(corr[:, :2], [3], [4], [0.0], [0, 2], [0, 1], [9.0, 9.0, 9.0])
    assert flags[0] >> 8 == 2


def test_host_mirror_exits_on_singular_like_the_reference(tmp_path):
    """sepselect.py:8-18: a singular sub-matrix ends the reference with sys.exit(); two identical traits do that"""
    import scipy.sparse as sp
    from scipy.io import mmwrite

    from cigwas_amd import sepselect as SS

    p, m = 3, 2
    n = p + m
    corr = np.eye(n)
    corr[0, 1] = corr[1, 0] = 1.0  # traits 0 and 1 are the same variable
    for t in (0, 1, 2):
        for v in (3, 4):
            corr[t, v] = corr[v, t] = 0.3 if t < 2 else 0.2
    corr[0, 2] = corr[2, 0] = corr[1, 2] = corr[2, 1] = 0.25
    adj = corr != 0
    np.fill_diagonal(adj, False)
    adj[3, 4] = adj[4, 3] = False  # markers 3 and 4: unshielded through the traits
    stem = os.path.join(str(tmp_path), "all_merged")
    mmwrite(stem + "_sam.mtx", sp.coo_matrix(adj.astype(np.int32)))
    mmwrite(stem + "_scm.mtx", sp.coo_matrix(corr))
    open(stem + ".mdim", "w").write(f"{n}\t{p}\t3\n")
    np.arange(m, dtype=np.int32).tofile(stem + ".ixs")
    with pytest.raises(SystemExit):
        SS.sepselect_merged(stem, 1e-4, 1000)
