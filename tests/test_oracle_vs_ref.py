"""CPU: the oracle's host-side restatement vs the REFERENCE's own host C++ compiled in place
(oracle/_ref/libref_host.so, built by `make -C oracle ref` where /root/reference exists).
Skipped when the library is absent (it is git-ignored but travels with gpurun snapshots)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libref_host.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref not built (no reference checkout)")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(SO)
    L.ref_subset_variables.restype = C.c_int
    L.ref_load_mxm.restype = C.c_int
    L.ref_load_pxp.restype = C.c_int
    L.ref_load_pxp.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_void_p, C.c_void_p]
    L.ref_load_mxp.restype = C.c_int
    L.ref_read_blocks.restype = C.c_int
    L.ref_load_phen.restype = C.c_int
    L.ref_read_block_from_bed.restype = C.c_int
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rand_graph(rng, n, dens):
    G = (rng.random((n, n)) < dens).astype(np.int32)
    G = np.triu(G, 1)
    return np.ascontiguousarray(G + G.T)


@pytest.mark.parametrize("seed", range(6))
def test_subset_variables_random(oracle, ref, seed):
    rng = np.random.default_rng(seed)
    n, p = 40, 4
    G = _rand_graph(rng, n, 0.06)
    for depth in range(0, 4):
        out = np.zeros(n, np.int32)
        cnt = ref.ref_subset_variables(_p(G), n, n - p, depth, _p(out))
        assert list(out[:cnt]) == list(oracle.subset_variables(G, n, n - p, depth))


def _read_outputs(base, with_sep):
    d = {"mdim": open(base + ".mdim").read()}
    for ext, dt in [("ixs", np.int32), ("adj", np.int32), ("corr", np.float32)] + ([("sep", np.int32)] if with_sep else []):
        d[ext] = np.fromfile(base + "." + ext, dt)
    return d


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("use_map", [False, True])
def test_reduce_gcs_files(oracle, ref, tmp_path, seed, use_map):
    rng = np.random.default_rng(100 + seed)
    n, p, ml = 18, 3, 3
    G = _rand_graph(rng, n, 0.2)
    Cm = rng.standard_normal((n, n)).astype(np.float32)
    S = np.full((n, n, 14), -1, np.int32)
    for i in range(n):
        for j in range(n):
            k = rng.integers(0, ml + 1)
            S[i, j, :k] = rng.choice(n, size=k, replace=False)
    P = np.sort(rng.choice(n, size=9, replace=False)).astype(np.int32)
    imap = (np.arange(n, dtype=np.int32) * 3 + 1) if use_map else None
    a, b = str(tmp_path / "ref"), str(tmp_path / "orc")
    ref.ref_reduce_gcs_to_file(_p(G), _p(Cm), _p(S), _p(P), len(P), n, p, ml, _p(imap) if use_map else None, a.encode())
    r = oracle.reduce_gcs(G, Cm, S, P, n, p, ml, imap)
    oracle.write_reduced(r, b, with_sep=True)
    ra, rb = _read_outputs(a, True), _read_outputs(b, True)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]) if k != "mdim" else ra[k] == rb[k], k


@pytest.mark.parametrize("use_map", [False, True])
def test_reduce_gc_files(oracle, ref, tmp_path, use_map):
    rng = np.random.default_rng(5)
    n, p = 15, 2
    G = _rand_graph(rng, n, 0.2)
    Cm = rng.standard_normal((n, n)).astype(np.float32)
    N = rng.uniform(1e3, 1e5, (n, n)).astype(np.float32)
    P = np.sort(rng.choice(n, size=7, replace=False)).astype(np.int32)
    imap = (np.arange(n, dtype=np.int32) + 100) if use_map else None
    a, b = str(tmp_path / "ref"), str(tmp_path / "orc")
    ref.ref_reduce_gc_to_file(_p(G), _p(Cm), _p(N), _p(P), len(P), n, p, 14, _p(imap) if use_map else None, a.encode())
    oracle.write_reduced(oracle.reduce_gc(G, Cm, N, P, n, p, 14, imap), b, with_sep=False)
    ra, rb = _read_outputs(a, False), _read_outputs(b, False)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]) if k != "mdim" else ra[k] == rb[k], k


def test_loaders_on_reference_fixtures(oracle, ref, golden_dir):
    g = lambda f: os.path.join(golden_dir, f)
    # mxm
    m = ref.ref_load_mxm(g("small_mxm.bin").encode(), None)
    buf = np.zeros(m * m, np.float32)
    ref.ref_load_mxm(g("small_mxm.bin").encode(), _p(buf))
    assert np.array_equal(buf.reshape(m, m), oracle.load_mxm(g("small_mxm.bin")))
    # pxp with fixed sample size
    corr = np.zeros(9, np.float32)
    ess = np.zeros(9, np.float32)
    p = ref.ref_load_pxp(g("trait_summary_stats.txt").encode(), None, 500000.0, _p(corr), _p(ess))
    _, c2, e2 = oracle.load_pxp(g("trait_summary_stats.txt"), sample_size=500000.0)
    assert p == 3 and np.array_equal(corr.reshape(3, 3), c2) and np.array_equal(ess.reshape(3, 3), e2)
    # mxp by block and by marker indices
    nph = C.c_int()
    corr = np.zeros(9, np.float32)
    nm = ref.ref_load_mxp(g("marker_trait_summary_stats.txt").encode(), None, None, 0, b"1", 0, 2, 0, _p(corr), None, C.byref(nph))
    c2, _ = oracle.load_mxp(g("marker_trait_summary_stats.txt"), range(0, 3))
    assert nm == 3 and nph.value == 3 and np.array_equal(corr.reshape(3, 3), c2)
    ix = np.fromfile(g("marker_indices.bin"), np.int32)
    corr = np.zeros(len(ix) * 3, np.float32)
    nm = ref.ref_load_mxp(g("marker_trait_summary_stats.txt").encode(), None, _p(ix), len(ix), b"", 0, 0, 0, _p(corr), None, C.byref(nph))
    c2, _ = oracle.load_mxp(g("marker_trait_summary_stats.txt"), ix)
    assert nm == len(ix) and np.array_equal(corr.reshape(nm, 3), c2)
    # blocks
    f, l, o = (np.zeros(8, np.int32) for _ in range(3))
    nb = ref.ref_read_blocks(g("test.blocks").encode(), _p(f), _p(l), _p(o), 8)
    bl = oracle.read_blocks(g("test.blocks"))
    assert nb == len(bl) and [(b[1], b[2], b[3]) for b in bl] == list(zip(f[:nb], l[:nb], o[:nb]))
    # phen with NaN
    ns = ref.ref_load_phen(g("with_nan.phen").encode(), None, C.byref(nph))
    data = np.zeros(ns * nph.value, np.float32)
    ref.ref_load_phen(g("with_nan.phen").encode(), _p(data), C.byref(nph))
    n2, p2, d2 = oracle.load_phen(g("with_nan.phen"))
    assert (ns, nph.value) == (n2, p2) and np.array_equal(data, d2, equal_nan=True)


def test_het_loaders_with_se(oracle, ref, tmp_path):
    """hetcor inputs: correlation + standard-error files, with NA entries."""
    rng = np.random.default_rng(11)
    p, m = 4, 6
    names = [f"T{i}" for i in range(p)]
    c = np.triu(rng.uniform(-0.4, 0.4, (p, p)), 1)
    np.fill_diagonal(c, 1.0)
    se = rng.uniform(0.001, 0.01, (p, p))
    with open(tmp_path / "pxp.txt", "w") as f, open(tmp_path / "pxp_se.txt", "w") as g:
        f.write(" ".join(names) + "\n")
        g.write(" ".join(names) + "\n")
        for i in range(p):
            vals = [("NaN" if (i, j) == (0, 2) else repr(float(c[i, j]))) for j in range(p)]
            f.write(names[i] + " " + " ".join(vals) + "\n")
            g.write(names[i] + " " + " ".join(repr(float(se[i, j])) for j in range(p)) + "\n")
    corr = np.zeros(p * p, np.float32)
    ess = np.zeros(p * p, np.float32)
    ref.ref_load_pxp(str(tmp_path / "pxp.txt").encode(), str(tmp_path / "pxp_se.txt").encode(), 0.0, _p(corr), _p(ess))
    _, c2, e2 = oracle.load_pxp(str(tmp_path / "pxp.txt"), se_path=str(tmp_path / "pxp_se.txt"))
    assert np.array_equal(corr.reshape(p, p), c2)
    assert np.array_equal(ess.reshape(p, p), e2, equal_nan=True)
    mc = rng.uniform(-0.05, 0.05, (m, p))
    ms = rng.uniform(0.001, 0.005, (m, p))
    with open(tmp_path / "mxp.txt", "w") as f, open(tmp_path / "mxp_se.txt", "w") as g:
        f.write("chr snp ref " + " ".join(names) + "\n")
        g.write("chr snp ref " + " ".join(names) + "\n")
        for i in range(m):
            vals = [("NaN" if (i, j) == (2, 1) else repr(float(mc[i, j]))) for j in range(p)]
            f.write(f"1 rs{i} A " + " ".join(vals) + "\n")
            g.write(f"1 rs{i} A " + " ".join(repr(float(ms[i, j])) for j in range(p)) + "\n")
    nph = C.c_int()
    corr = np.zeros(4 * p, np.float32)
    ess = np.zeros(4 * p, np.float32)
    nm = ref.ref_load_mxp(str(tmp_path / "mxp.txt").encode(), str(tmp_path / "mxp_se.txt").encode(), None, 0, b"1", 1, 4, 0,
                          _p(corr), _p(ess), C.byref(nph))
    c2, e2 = oracle.load_mxp(str(tmp_path / "mxp.txt"), range(1, 5), se_path=str(tmp_path / "mxp_se.txt"))
    assert nm == 4 and np.array_equal(corr.reshape(4, p), c2) and np.array_equal(ess.reshape(4, p), e2, equal_nan=True)


@pytest.mark.parametrize("seed,n,maxb", [(0, 3000, 300), (1, 5000, 800), (2, 1200, 150), (3, 800, 2000)])
def test_blocking_matches_reference_code(oracle, ref, seed, n, maxb):
    """the oracle's restatement of blocking.cpp against the reference's own blocking.cpp (compiled in place):
    smoothed curve bit-equal, identical blocks, on random LD-like row-sum profiles"""
    rng = np.random.default_rng(100 + seed)
    # forward |r| row sums look like a noisy positive curve with dips at block boundaries
    base = 20.0 + 10.0 * np.sin(np.arange(n) / 37.0) + 5.0 * np.sin(np.arange(n) / 211.0 + seed)
    dips = np.ones(n)
    for c in rng.integers(50, n - 50, size=max(3, n // 250)):
        dips[max(0, c - 15): c + 15] *= rng.uniform(0.2, 0.6)
    v = (base * dips + rng.normal(0.0, 0.8, n)).astype(np.float32)
    ref.ref_hanning_smoothing.restype = None
    for ws in (5, 101, 333):
        r = np.zeros(n, np.float64)
        ref.ref_hanning_smoothing(_p(v), C.c_int(n), C.c_int(ws), _p(r))
        assert np.array_equal(oracle.hanning_smoothing(v, ws), r)
    first = np.zeros(n, np.int64)
    last = np.zeros(n, np.int64)
    ref.ref_block_chr.restype = C.c_int
    k = ref.ref_block_chr(_p(v), C.c_int(n), C.c_int(maxb), _p(first), _p(last), C.c_int(n))
    assert k > 0
    assert oracle.block_chr(v, maxb) == [(int(first[i]), int(last[i])) for i in range(k)]



def _prep_inputs(tmp_path, synth, golden_dir, which):
    """bfiles for `mps prep`: the reference's own small.* or a synthetic set with missing genotypes"""
    import shutil

    d = tmp_path / which
    d.mkdir()
    stem = str(d / "x")
    if which == "small":
        for sfx in (".bed", ".bim", ".fam"):
            shutil.copy(os.path.join(golden_dir, "small" + sfx), stem + sfx)
    else:
        N = 1003  # not a multiple of four: the last byte of every marker is partly padding
        G = synth.make_genotypes(300, N, np.random.default_rng(5), miss=0.02)
        mu, sd = synth.bed_stats(G)
        synth.write_bfiles(stem, synth.pack_bed(G), N, mu, sd)
        for sfx in (".dim", ".means", ".stds"):
            os.remove(stem + sfx)
    return stem


@pytest.mark.parametrize("which", ["small", "synthetic"])
def test_mps_prep_writes_the_reference_files(ref, kat, synth, golden_dir, tmp_path, which):
    """`mps prep` of this build (host code, runs without a GPU) against the reference's prep.cpp compiled in place:
    .dim/.means/.stds/.modes byte for byte; on small.* also the values the reference's prep_tests.cpp expects"""
    import subprocess

    from cigwas_amd.cli import MPS_PATH

    if not os.path.exists(MPS_PATH):
        pytest.skip("mps not built")
    ours = _prep_inputs(tmp_path, synth, golden_dir, which)
    theirs = ours + "_ref"
    for sfx in (".bed", ".bim", ".fam"):
        os.link(ours + sfx, theirs + sfx)
    subprocess.run([MPS_PATH, "prep", ours], check=True, capture_output=True)
    ref.ref_prep.restype = None
    ref.ref_prep(theirs.encode())
    for sfx in (".dim", ".means", ".stds", ".modes"):
        assert open(ours + sfx, "rb").read() == open(theirs + sfx, "rb").read(), sfx
    if which == "small":
        k = kat["prep_small"]
        assert np.allclose(np.loadtxt(ours + ".means"), k["exp_means"], atol=k["tol"], rtol=0)
        assert np.allclose(np.loadtxt(ours + ".stds"), k["exp_stds"], atol=k["tol"], rtol=0)
        assert open(ours + ".dim").read().split() == [str(v) for v in k["exp_dims"]]
