"""GPU: BASELINE config 1 as specified -- the 500-SNP x 5-trait correlation matrix of the reference's random-DAG
simulator (simulate_dag.R, restated in synth.rand_dag_corr), skeleton at l <= 1: both engines through the device API
and through the reference-named host entry points against the oracle, and through `mps cuskss` files."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPS = os.path.join(ROOT, "ci-gwas_amd", "csrc", "mps")
SNP, TR, N, ML = 500, 5, 16000, 14


@pytest.fixture(scope="module")
def cg():
    import cigwas_amd

    return cigwas_amd


@pytest.fixture(scope="module")
def Cm(synth):
    return synth.rand_dag_corr(SNP, TR, 2, N, seed=1)


@pytest.mark.parametrize("alpha", [1e-4, 1e-2])
def test_c1_both_engines_match_the_oracle(cg, oracle, Cm, alpha):
    n = Cm.shape[0]
    Th = cg.threshold_array(N, alpha)
    ref = oracle.skeleton(Cm, oracle.threshold_array(N, alpha), 1)
    e = cg.Engine(0)
    Cd = cg.DeviceArray(Cm)
    st = e.run_skeleton(Cd.ptr, n, Th, 1)
    assert st.level == ref.level == 2 and np.array_equal(e.adjacency(), ref.G)
    x, y, lv, z, S = e.sepsets()
    dense = np.full((n, n, ML), -1, np.int32)
    dense[x, y] = S
    assert np.array_equal(dense, ref.sepset)
    assert np.allclose(e.pmax(Cd.ptr), ref.pmax, rtol=0, atol=1e-6)
    assert list(st.canonical_tests[:2]) == [int(v) for v in ref.tests[:2]]
    # hetcor engine: one sample size, then per-pair effective sample sizes with a time index
    th = cg.hetcor_threshold(alpha)
    ti0 = np.zeros(n, np.int32)
    ones = np.ones((n, n), np.int32)
    r2 = oracle.hetcor_skeleton(Cm, ones, np.full((n, n), N, np.float32), th, 1, ti0)
    st2 = e.run_hetcor(Cd.ptr, n, th, 1, ess_uniform=float(N))
    assert st2.level == r2.level and np.array_equal(e.adjacency(), r2.G)
    rng = np.random.default_rng(5)
    ess = np.full((n, n), N, np.float32)
    blk = rng.uniform(0.5, 1.0, (n, TR)).astype(np.float32) * N
    ess[:, SNP:] = blk
    ess[SNP:, :] = blk.T
    ess[SNP:, SNP:] = np.minimum(ess[SNP:, SNP:], ess[SNP:, SNP:].T)
    ti = np.zeros(n, np.int32)
    ti[SNP:] = [1, 2, 1, 3, 2]
    r3 = oracle.hetcor_skeleton(Cm, ones, ess, th, 1, ti)
    Nd = cg.DeviceArray(ess)
    st3 = e.run_hetcor(Cd.ptr, n, th, 1, N_dev=Nd.ptr, time_index=ti)
    assert st3.level == r3.level and np.array_equal(e.adjacency(), r3.G)
    Nd.free()
    Cd.free()
    e.close()
    # the reference-named entry points (host buffers in and out, cuPC-S.h:196, hetcor-cuPC-S.h:46)
    G, level, pmax, sep = cg.Skeleton(Cm, Th, 1)
    assert level == ref.level and np.array_equal(G, ref.G) and np.array_equal(sep, ref.sepset)
    G3, level3 = cg.hetcor_skeleton(Cm, ones, ess, th, 1, ti)
    assert level3 == r3.level and np.array_equal(G3, r3.G)


def test_c1_through_mps_cuskss_files(oracle, Cm, tmp_path):
    """the same matrix as mxm / mxp / pxp files through `mps cuskss` (l <= 1 in both stages) against the oracle's
    two-stage pipeline on the loaded files"""
    m, p = SNP, TR
    names = [f"Y{k + 1}" for k in range(p)]
    Cm[:m, :m][np.tril_indices(m)].astype(np.float32).tofile(tmp_path / "mxm.bin")
    with open(tmp_path / "mxp.txt", "w") as f:
        f.write("chr snp ref " + " ".join(names) + "\n")
        for i in range(m):
            f.write(f"1 X{i + 1} A " + " ".join(repr(float(Cm[i, m + k])) for k in range(p)) + "\n")
    with open(tmp_path / "pxp.txt", "w") as f:
        f.write(" ".join(names) + "\n")
        for a in range(p):
            f.write(names[a] + " " + " ".join(repr(float(Cm[m + a, m + b])) for b in range(p)) + "\n")
    with open(tmp_path / "blocks.txt", "w") as f:
        f.write(f"1\t0\t{m - 1}\n")
    t = lambda f: str(tmp_path / f)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([MPS, "cuskss", t("mxm.bin"), t("mxp.txt"), "NULL", t("pxp.txt"), "NULL", "NULL", "0", t("blocks.txt"), "NULL",
                        "0.0001", "1", "1", "1", str(N), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    mxm = oracle.load_mxm(t("mxm.bin"))
    _, pxp, _ = oracle.load_pxp(t("pxp.txt"), sample_size=float(N))
    mxp, _ = oracle.load_mxp(t("mxp.txt"), range(m))
    sq, es = oracle.make_square_cuskss_inputs(mxm, mxp, pxp, float(N))
    assert np.array_equal(sq, Cm)  # repr() round-trips fp32: the engine sweeps the simulator's matrix itself
    ref = oracle.cuskss_from_square(sq, es, p, 1e-4, 1, 1, 1)
    oracle.write_reduced(ref, t("ref"), with_sep=False)
    for ext in (".mdim", ".ixs", ".adj", ".corr"):
        assert open(str(out / f"1_0_{m - 1}") + ext, "rb").read() == open(t("ref") + ext, "rb").read(), ext
    assert ref.num_var > p  # some marker is adjacent to a trait
