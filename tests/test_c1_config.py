"""BASELINE config 1 (CPU part): the 500-SNP x 5-trait matrix of the reference's random-DAG simulator
(/root/reference/simulation/simulate_dag.R:3-98,106-115, restated in ci-gwas_amd/synth.py: rand_dag_corr), the
oracle's skeleton at l <= 1 on it, and a plain double-precision PC-stable (pcalg::skeleton's algorithm, written out
here for l <= 1) beside the oracle -- the config's "plumbing + parity ref"."""
import numpy as np
import pytest

SNP, TR, NL, N = 500, 5, 2, 16000


@pytest.fixture(scope="module")
def c1(synth):
    return synth.rand_dag_corr(SNP, TR, NL, N, seed=1, return_dag=True)


def test_generator_follows_gen_rand_dag(c1, synth):
    Cm, G, A = c1
    pq = SNP + NL + TR
    assert Cm.shape == (SNP + TR, SNP + TR) and Cm.dtype == np.float32
    assert np.array_equal(Cm, Cm.T) and np.all(np.diag(Cm) == 1.0) and np.abs(Cm).max() <= 1.0
    assert G.shape == (pq, pq) and not np.tril(G).any()  # simulate_dag.R:22-28,42-48: edges point forward only
    # :15-16: a marker has deg / SNP * (pq - i) children on average, a latent / trait min(deg / Tr, 1) per later variable
    assert 0.7 * 3 * SNP * 0.5 < G[:SNP].sum() < 1.4 * 3 * SNP * 0.55
    # effect ranges and placement (:54-78)
    mm = A[:SNP, :SNP][G[:SNP, :SNP] == 1]
    mt = A[:SNP, SNP:][G[:SNP, SNP:] == 1]
    tt = A[SNP:][G[SNP:] == 1]
    assert mm.size and np.all((np.abs(mm) >= 0.001) & (np.abs(mm) <= 0.2))
    assert mt.size and np.all((np.abs(mt) >= 0.001) & (np.abs(mt) <= 0.05))
    assert np.all((np.abs(tt) >= 0.001) & (np.abs(tt) <= 0.2))
    assert np.all(A[G == 0] == 0) and (mm < 0).any() and (mm > 0).any()
    # the sample correlation of a marker-marker edge is its effect up to the estimation error 1 / sqrt(n) (parents
    # of one child are independent roots or nearly so)
    i, j = np.nonzero(G[:SNP, :SNP])
    assert np.abs(Cm[i, j] - A[i, j]).max() < 6.0 / np.sqrt(N) + 0.05
    # deterministic, and the seed matters
    assert np.array_equal(Cm, synth.rand_dag_corr(SNP, TR, NL, N, seed=1))
    assert not np.array_equal(Cm, synth.rand_dag_corr(SNP, TR, NL, N, seed=2))


def _pcstable_l1_f64(Cm, n_samples, alpha, maxlevel):
    """PC-stable for l <= 1 in double precision, pcalg's decision rule sqrt(n - |S| - 3) |atanh r| <= qnorm(1 - a/2)
    (pcalg::gaussCItest / skeleton(method = "stable")); returns adjacency and every |z| - threshold margin"""
    from scipy.stats import norm

    C = Cm.astype(np.float64)
    n = C.shape[0]
    q = norm.ppf(1 - alpha / 2)
    G = np.ones((n, n), bool)
    np.fill_diagonal(G, False)
    margins = []
    z0 = np.sqrt(n_samples - 3) * np.abs(np.arctanh(np.clip(C, -0.9999999, 0.9999999)))
    margins.append((np.abs(z0 - q) / np.sqrt(n_samples - 3))[np.triu_indices(n, 1)])
    G &= z0 > q
    if maxlevel >= 1:
        frozen = G.copy()
        for x in range(n):
            nb = np.flatnonzero(frozen[x])
            for y in nb:
                if not G[x, y]:
                    continue
                for s in nb:
                    if s == y:
                        continue
                    r = (C[x, y] - C[x, s] * C[y, s]) / np.sqrt((1 - C[x, s] ** 2) * (1 - C[y, s] ** 2))
                    z = np.sqrt(n_samples - 4) * abs(np.arctanh(r))
                    margins.append(np.array([abs(z - q) / np.sqrt(n_samples - 4)]))
                    if z <= q:
                        G[x, y] = G[y, x] = False
                        break
    return G.astype(np.int32), np.concatenate(margins)


@pytest.mark.parametrize("alpha", [1e-4, 1e-2])
def test_oracle_skeleton_on_c1_matches_double_precision_pcstable(c1, oracle, alpha):
    Cm = c1[0]
    n = Cm.shape[0]
    ref = oracle.skeleton(Cm, oracle.threshold_array(N, alpha), 1)
    G64, margins = _pcstable_l1_f64(Cm, N, alpha, 1)
    assert ref.level == 2 and ref.tests[0] == n * (n - 1) // 2 and ref.tests[1] > 0
    # fp32 reference arithmetic vs double precision: the same graph unless a test sits within fp32 rounding of the threshold
    if margins.min() > 1e-6:
        assert np.array_equal(ref.G, G64)
    else:
        assert (ref.G != G64).sum() <= 2 * int((margins <= 1e-6).sum())
    # what the skeleton of this DAG has to look like: sparse, every trait keeps a neighbour, separating sets of size one
    assert ref.G.sum() // 2 < 4 * n and np.all(ref.G[SNP:].sum(axis=1) > 0)
    has = ref.sepset[:, :, 0] != -1
    assert has.any() and np.all(ref.sepset[:, :, 1:] == -1)
    # the hetcor engine's arithmetic with one sample size removes the same edges (thresholds coincide)
    r2 = oracle.hetcor_skeleton(Cm, np.ones((n, n), np.int32), np.full((n, n), N, np.float32), oracle.hetcor_threshold(alpha), 1,
                                np.zeros(n, np.int32))
    assert np.array_equal(r2.G, ref.G)
