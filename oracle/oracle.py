"""ctypes/numpy front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import
this module.  The arithmetic lives in oracle/cusk_oracle.c (a C restatement of
the reference's CUDA engines); the host-side bookkeeping of the reference
(square assembly, BFS pruning, sub-matrix extraction, file formats, loaders)
is restated here in numpy, each function citing the reference file:line
(paths relative to /root/reference/cusk).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcusk_oracle.so")
ML = 14


def build(force: bool = False) -> None:
    """Compile oracle/cusk_oracle.c (and oracle/_ref when the reference is present)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "cusk_oracle.c")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, _SO])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.orc_qnorm.restype = C.c_double
        L.orc_qnorm.argtypes = [C.c_double]
        L.orc_threshold_array.argtypes = [C.c_int, C.c_float, f32p]
        L.orc_hetcor_threshold.restype = C.c_float
        L.orc_hetcor_threshold.argtypes = [C.c_float]
        L.orc_fisher_z_ratio.restype = C.c_float
        L.orc_fisher_z_ratio.argtypes = [C.c_float]
        L.orc_fisher_z_diff.restype = C.c_float
        L.orc_fisher_z_diff.argtypes = [C.c_float]
        L.orc_binom.restype = C.c_uint64
        L.orc_binom.argtypes = [C.c_int, C.c_int]
        L.orc_ith_combination.argtypes = [i32p, C.c_int, C.c_int, C.c_uint64]
        L.orc_pinv.argtypes = [C.c_int, f32p, f32p]
        L.orc_ci_test.restype = C.c_float
        L.orc_ci_test.argtypes = [f32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.POINTER(C.c_float)]
        L.orc_skeleton.argtypes = [f32p, C.c_int, i32p, f32p, C.POINTER(C.c_int), C.c_int, f32p, i32p, i64p, i64p]
        L.orc_hetcor_skeleton.argtypes = [f32p, C.c_int, i32p, f32p, C.c_float, C.POINTER(C.c_int), C.c_int, i32p, i64p, i64p]
        L.orc_pcstable_f64.restype = C.c_int
        L.orc_pcstable_f64.argtypes = [f32p, C.c_int, i32p, C.c_double, C.c_double, C.c_int, i64p]
        L.orc_marker_phen_corr_pearson.argtypes = [u8p, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, f32p, f32p]
        L.orc_phen_corr_pearson.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p]
        L.orc_marker_corr_npn.argtypes = [u8p, C.c_size_t, C.c_size_t, f32p]
        L.orc_npn_from_counts.restype = C.c_float
        L.orc_npn_from_counts.argtypes = [f32p]
        L.orc_marker_corr_banded.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
        L.orc_banded_row_abs_sums.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p]
        L.orc_hanning_smoothing.argtypes = [f32p, C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
        L.orc_block_chr.argtypes = [f32p, C.c_int, C.c_int, i64p, i64p, C.c_int]
        L.orc_block_chr.restype = C.c_int
        _lib = L
    return _lib


# --------------------------------------------------------------------------
# thresholds (src/cuPC_call_prep.cpp:13-27)
# --------------------------------------------------------------------------
def threshold_array(n: int, alpha: float) -> np.ndarray:
    out = np.zeros(ML + 1, np.float32)
    lib().orc_threshold_array(int(n), float(np.float32(alpha)), out)
    return out


def hetcor_threshold(alpha: float) -> float:
    return float(lib().orc_hetcor_threshold(float(np.float32(alpha))))


def qnorm(p: float) -> float:
    return float(lib().orc_qnorm(p))


def binom(n: int, k: int) -> int:
    return int(lib().orc_binom(n, k))


def ith_combination(n: int, p: int, rank1: int) -> np.ndarray:
    out = np.zeros(p, np.int32)
    lib().orc_ith_combination(out, n, p, rank1)
    return out


def pinv(m2: np.ndarray) -> np.ndarray:
    m2 = np.ascontiguousarray(m2, np.float32)
    out = np.zeros_like(m2)
    lib().orc_pinv(m2.shape[0], m2, out)
    return out


def ci_test(Cm: np.ndarray, x: int, y: int, S) -> tuple[float, float]:
    """(rho, Z) of one test in the reference's fp32 operation order."""
    S = np.ascontiguousarray(S, np.int32)
    rho = C.c_float()
    z = lib().orc_ci_test(np.ascontiguousarray(Cm, np.float32), Cm.shape[0], x, y, S, len(S), C.byref(rho))
    return float(rho.value), float(z)


# --------------------------------------------------------------------------
# engines
# --------------------------------------------------------------------------
@dataclass
class SkeletonResult:
    G: np.ndarray  # n x n int32
    level: int  # value the reference leaves in *l
    pmax: np.ndarray | None = None  # n x n float32
    sepset: np.ndarray | None = None  # n x n x 14 int32, -1 padded
    tests: np.ndarray = field(default_factory=lambda: np.zeros(ML + 1, np.int64))
    subsets: np.ndarray = field(default_factory=lambda: np.zeros(ML + 1, np.int64))


def skeleton(Cm: np.ndarray, Th: np.ndarray, maxlevel: int) -> SkeletonResult:
    """`Skeleton` (src/cuPC-S.cu:61-450) on an n x n fp32 correlation matrix."""
    Cm = np.ascontiguousarray(Cm, np.float32)
    n = Cm.shape[0]
    G = np.ones((n, n), np.int32)
    pmax = np.zeros((n, n), np.float32)
    sep = np.zeros((n, n, ML), np.int32)
    tests = np.zeros(ML + 1, np.int64)
    subsets = np.zeros(ML + 1, np.int64)
    lvl = C.c_int(0)
    Th = np.ascontiguousarray(Th, np.float32)
    assert Th.size >= min(maxlevel, ML) + 1
    lib().orc_skeleton(Cm, n, G, Th, C.byref(lvl), int(maxlevel), pmax, sep, tests, subsets)
    return SkeletonResult(G, lvl.value, pmax, sep, tests, subsets)


def hetcor_skeleton(Cm, G, N, th: float, maxlevel: int, time_index) -> SkeletonResult:
    """`hetcor_skeleton` (src/hetcor-cuPC-S.cu:75-341); G is copied, then updated."""
    Cm = np.ascontiguousarray(Cm, np.float32)
    n = Cm.shape[0]
    G = np.array(G, np.int32).reshape(n, n).copy()
    N = np.ascontiguousarray(N, np.float32).reshape(n, n)
    ti = np.ascontiguousarray(time_index, np.int32)
    tests = np.zeros(ML + 1, np.int64)
    subsets = np.zeros(ML + 1, np.int64)
    lvl = C.c_int(0)
    lib().orc_hetcor_skeleton(Cm, n, G, N, float(np.float32(th)), C.byref(lvl), int(maxlevel), ti, tests, subsets)
    return SkeletonResult(G, lvl.value, None, None, tests, subsets)


def pcstable_f64(Cm: np.ndarray, num_samples: float, alpha: float, maxlevel: int) -> SkeletonResult:
    """CPU baseline variant (not a parity oracle): PC-stable in double precision with pcalg::gaussCItest semantics,
    sqrt(N - |S| - 3) |atanh r| <= qnorm(1 - alpha/2) removes the edge."""
    Cm = np.ascontiguousarray(Cm, np.float32)
    n = Cm.shape[0]
    G = np.zeros((n, n), np.int32)
    tests = np.zeros(ML + 1, np.int64)
    lvl = lib().orc_pcstable_f64(Cm, n, G, float(num_samples), float(alpha), int(maxlevel), tests)
    return SkeletonResult(G, int(lvl), None, None, tests)


# --------------------------------------------------------------------------
# correlation build (src/corr_kernels.cu, src/corr_host.cu:1023-1197)
# --------------------------------------------------------------------------
def marker_phen_corr_pearson(bed, phen, m, N, p, means, stds) -> np.ndarray:
    out = np.zeros(m * p, np.float32)
    lib().orc_marker_phen_corr_pearson(
        np.ascontiguousarray(bed, np.uint8), np.ascontiguousarray(phen, np.float32), m, N, p,
        np.ascontiguousarray(means, np.float32), np.ascontiguousarray(stds, np.float32), out)
    return out


def corr_pearson_npn(bed, phen, m, N, p, means, stds):
    """(mxm upper-tri w/o diag, mxp row-major m x p, pxp upper-tri) like cu_corr_pearson_npn."""
    bed = np.ascontiguousarray(bed, np.uint8)
    phen = np.ascontiguousarray(phen, np.float32)
    mxm = np.zeros(m * (m - 1) // 2, np.float32)
    pxp = np.zeros(p * (p - 1) // 2, np.float32)
    lib().orc_marker_corr_npn(bed, m, N, mxm)
    mxp = marker_phen_corr_pearson(bed, phen, m, N, p, means, stds)
    if p > 1:
        lib().orc_phen_corr_pearson(phen, N, p, pxp)
    return mxm, mxp, pxp


def npn_from_counts(s) -> float:
    return float(lib().orc_npn_from_counts(np.ascontiguousarray(s, np.float32)))


def square_from_cusk_corrs(mxm, mxp, pxp, m: int, p: int) -> np.ndarray:
    """n x n assembly of src/cli.cpp:597-649 (markers first, then traits; diag 1)."""
    n = m + p
    sq = np.ones((n, n), np.float32)
    iu = np.triu_indices(m, 1)
    sq[:m, :m][iu] = mxm
    sq[:m, :m].T[iu] = mxm
    mp = np.asarray(mxp, np.float32).reshape(m, p)
    sq[:m, m:] = mp
    sq[m:, :m] = mp.T
    if p > 1:
        ip = np.triu_indices(p, 1)
        sq[m:, m:][ip] = pxp
        sq[m:, m:].T[ip] = pxp
    return sq


def prefilter_count(mxp, th0: float) -> int:
    """src/cli.cpp:561-565: number of |atanh(r)| >= Th[0] (double arithmetic on float r)."""
    c = np.asarray(mxp, np.float32)
    one = np.float32(1)
    z = np.abs(0.5 * (np.log(np.abs((one + c)).astype(np.float64)) - np.log(np.abs(one - c).astype(np.float64))))
    return int(np.sum(z >= np.float64(np.float32(th0))))


# --------------------------------------------------------------------------
# graph reduction (src/parent_set.cpp)
# --------------------------------------------------------------------------
def subset_variables(G: np.ndarray, num_var: int, num_markers: int, max_depth: int) -> np.ndarray:
    """parent_set.cpp:8-53 -> sorted retained indices (all traits + markers within max_depth)."""
    G = np.asarray(G).reshape(num_var, num_var)
    keep = set(range(num_markers, num_var))
    for start in range(num_markers, num_var):
        cur = set(range(num_markers, num_var))
        q = [start]
        for _ in range(max_depth):
            nq = []
            for node in q:
                for c in np.nonzero(G[node, :num_markers] == 1)[0]:
                    c = int(c)
                    if c not in cur:
                        cur.add(c)
                        nq.append(c)
            q = nq
        keep |= cur
    return np.array(sorted(keep), np.int32)


@dataclass
class Reduced:
    num_var: int
    num_phen: int
    max_level: int
    new_to_old: np.ndarray
    G: np.ndarray
    C: np.ndarray
    S: np.ndarray | None  # sepsets (cusk) or ESS matrix (cuskss)

    def num_markers(self) -> int:
        return self.num_var - self.num_phen


def reduce_gcs(G, Cm, S, P, num_var, num_phen, max_level, index_map=None) -> Reduced:
    """parent_set.cpp:84-175.  S has stride 14; output stride is max_level.

    With index_map the reference keys old_to_new by index_map[P[i]] while the
    sepset entries are still in the P index space; an entry
    that is not a key reads as 0 through unordered_map::operator[].
    """
    P = np.asarray(P, np.int32)
    G = np.asarray(G, np.int32).reshape(num_var, num_var)
    Cm = np.asarray(Cm, np.float32).reshape(num_var, num_var)
    S = np.asarray(S, np.int32).reshape(num_var, num_var, ML)
    k = len(P)
    if index_map is None:
        new_to_old = P.copy()
    else:
        new_to_old = np.asarray(index_map, np.int32)[P]
    old_to_new = {int(v): i for i, v in enumerate(new_to_old)}
    Pset = set(int(v) for v in P)
    Gr = G[np.ix_(P, P)].copy()
    Cr = Cm[np.ix_(P, P)].copy()
    Sr = np.full((k, k, max_level), -1, np.int32)
    for a, i in enumerate(P):
        for b, j in enumerate(P):
            cnt = 0
            for l in range(max_level):
                e = int(S[i, j, l])
                if e != -1 and e in Pset:
                    Sr[a, b, cnt] = old_to_new.get(e, 0)
                    cnt += 1
    return Reduced(k, num_phen, max_level, new_to_old, Gr, Cr, Sr)


def reduce_gc(G, Cm, S, P, num_var, num_phen, max_level, index_map=None) -> Reduced:
    """parent_set.cpp:177-238 (S is the n x n effective-sample-size matrix)."""
    P = np.asarray(P, np.int32)
    G = np.asarray(G, np.int32).reshape(num_var, num_var)
    Cm = np.asarray(Cm, np.float32).reshape(num_var, num_var)
    S = np.asarray(S, np.float32).reshape(num_var, num_var)
    new_to_old = P.copy() if index_map is None else np.asarray(index_map, np.int32)[P]
    ix = np.ix_(P, P)
    return Reduced(len(P), num_phen, max_level, new_to_old, G[ix].copy(), Cm[ix].copy(), S[ix].copy())


def write_reduced(r: Reduced, base: str, with_sep: bool) -> None:
    """ReducedGCS::to_file / ReducedGC::to_file (include/mps/parent_set.h:42-52,99-108)."""
    with open(base + ".mdim", "w") as f:
        f.write(f"{r.num_var}\t{r.num_phen}\t{r.max_level}\n")
    np.asarray(r.new_to_old, np.int32).tofile(base + ".ixs")
    np.asarray(r.G, np.int32).tofile(base + ".adj")
    np.asarray(r.C, np.float32).tofile(base + ".corr")
    if with_sep:
        np.asarray(r.S, np.int32).tofile(base + ".sep")


# --------------------------------------------------------------------------
# two-stage drivers (src/cli.cpp:29-87, :194-346, :660-677)
# --------------------------------------------------------------------------
def cusk_from_corr(sq: np.ndarray, num_phen: int, Th, max_level: int, max_level_two: int, depth: int) -> Reduced:
    """Skeleton -> prune -> second Skeleton on the reduced set (cli.cpp:660-677, :62-87)."""
    n = sq.shape[0]
    m = n - num_phen
    r1 = skeleton(sq, Th, max_level)
    P = subset_variables(r1.G, n, m, depth)
    gcs = reduce_gcs(r1.G, sq, r1.sepset, P, n, num_phen, max_level)
    r2 = skeleton(gcs.C, Th, max_level_two)
    P2 = subset_variables(r2.G, gcs.num_var, gcs.num_markers(), depth)
    return reduce_gcs(r2.G, gcs.C, r2.sepset, P2, gcs.num_var, num_phen, ML, gcs.new_to_old)


def run_cusk(gc: Reduced, th: float, depth: int, max_level: int, time_index_traits) -> Reduced:
    """cli.cpp:29-60."""
    ti = np.zeros(gc.num_var, np.int32)
    ti[gc.num_markers():] = np.asarray(time_index_traits, np.int32)[: gc.num_phen]
    res = hetcor_skeleton(gc.C, gc.G, gc.S, th, max_level, ti)
    P = subset_variables(res.G, gc.num_var, gc.num_markers(), depth)
    return reduce_gc(res.G, gc.C, gc.S, P, gc.num_var, gc.num_phen, ML, gc.new_to_old)


def cuskss_from_square(sq_corr, sq_ess, num_phen, alpha, max_level_one, max_level_two, depth,
                       time_index_traits=None) -> Reduced:
    """cli.cpp:294-338 (non-trait-only branch) and :225-254 (trait-only when m == 0)."""
    n = sq_corr.shape[0]
    if time_index_traits is None:
        time_index_traits = np.ones(num_phen, np.int32)
    gc = Reduced(n, num_phen, max_level_one, np.arange(n, dtype=np.int32), np.ones((n, n), np.int32),
                 np.asarray(sq_corr, np.float32), np.asarray(sq_ess, np.float32))
    th = hetcor_threshold(alpha)
    gc = run_cusk(gc, th, depth, max_level_one, time_index_traits)
    if max_level_two > 0 and n > num_phen:
        gc = run_cusk(gc, th, depth, max_level_two, time_index_traits)
    return gc


# --------------------------------------------------------------------------
# loaders (src/marker_summary_stats.cpp, marker_trait_summary_stats.cpp,
#          trait_summary_stats.cpp, io.cpp, phen.cpp)
# --------------------------------------------------------------------------
def load_mxm(path: str) -> np.ndarray:
    """f32 lower triangle incl. diagonal, row-major -> full m x m, NaN -> 0 (marker_summary_stats.cpp:8-24)."""
    t = np.fromfile(path, np.float32)
    m = int((np.sqrt(8 * t.size + 1) - 1) / 2)
    out = np.ones((m, m), np.float32)
    il = np.tril_indices(m)
    v = np.where(np.isnan(t), np.float32(0), t)
    out[il] = v
    out.T[il] = v
    return out


def _stof(s: str) -> np.float32:
    return np.float32(float(s))


def load_pxp(path: str, sample_size: float | None = None, se_path: str | None = None):
    """trait_summary_stats.cpp:5-169 -> (names, corr p x p, ess p x p or None)."""
    with open(path) as f:
        lines = f.read().split("\n")
    header = lines[0].split()
    p = len(header)
    corr = np.ones((p, p), np.float32)
    ess = None
    se_lines = None
    if se_path is not None:
        with open(se_path) as f:
            se_lines = f.read().split("\n")
        ess = np.zeros((p, p), np.float32)
    elif sample_size is not None:
        ess = np.full((p, p), np.float32(sample_size), np.float32)
    row = 0
    for li, line in enumerate(lines[1:], start=1):
        fields = line.split()
        if not fields:
            break
        sef = se_lines[li].split() if se_lines is not None else None
        for j in range(1, p + 1):
            v = _stof(fields[j])
            if se_lines is None:
                corr[row, j - 1] = np.float32(0) if np.isnan(v) else v
            elif np.isnan(v):
                corr[row, j - 1] = 0
                ess[row, j - 1] = np.nan
            else:
                se = _stof(sef[j])
                with np.errstate(all="ignore"):
                    ss = np.float32((1.0 - float(v * v)) / float(se))
                    ess[row, j - 1] = ss * ss
                corr[row, j - 1] = v
        row += 1
    iu = np.triu_indices(p, 1)
    corr.T[iu] = corr[iu]
    if se_lines is not None:
        ess.T[iu] = ess[iu]
    return header, corr, ess


def load_mxp(path: str, rows, se_path: str | None = None):
    """marker_trait_summary_stats.cpp:40-299.  rows = ascending global line indices
    (a block's global range or the merged marker indices) -> (corr m x p, ess or None)."""
    rows = [int(r) for r in rows]
    want = set(rows)
    out, ess = [], []
    with open(path) as f:
        header = f.readline().split()
        assert header[:3] == ["chr", "snp", "ref"], "marker-trait summary stat file has bad header"
        p = len(header) - 3
        sef = open(se_path) if se_path else None
        if sef:
            sef.readline()
        for ln, line in enumerate(f):
            sel = sef.readline() if sef else None
            if ln not in want:
                continue
            fields = line.split()
            sf = sel.split() if sel else None
            rc, re = [], []
            for j in range(3, p + 3):
                na = fields[j] in ("NA", "NaN", "nan") or (sef is not None and fields[j] == "NAN")
                if na:
                    rc.append(np.float32(0))
                    re.append(np.float32(np.nan))
                else:
                    rho = _stof(fields[j])
                    rc.append(rho)
                    if sef:
                        se = _stof(sf[j])
                        with np.errstate(all="ignore"):
                            ss = np.float32((1.0 - float(rho * rho)) / float(se))
                            re.append(ss * ss)
            out.append(rc)
            ess.append(re)
        if sef:
            sef.close()
    corr = np.array(out, np.float32).reshape(len(out), p)
    return corr, (np.array(ess, np.float32).reshape(len(out), p) if se_path else None)


def make_square_cuskss_inputs(mxm, mxp, pxp, sample_size, mxp_ess=None, pxp_ess=None):
    """cli.cpp:89-173."""
    m, p = mxm.shape[0], pxp.shape[0]
    n = m + p
    sq = np.ones((n, n), np.float32)
    es = np.full((n, n), np.float32(sample_size), np.float32)
    sq[:m, :m] = mxm
    sq[:m, m:] = mxp
    sq[m:, :m] = mxp.T
    sq[m:, m:] = pxp
    if mxp_ess is not None:
        es[:m, m:] = mxp_ess
        es[m:, :m] = mxp_ess.T
        es[m:, m:] = pxp_ess
    return sq, es


def read_blocks(path: str):
    """io.cpp:74-101 -> list of (chr, first, last, global_offset)."""
    blocks = []
    off = 0
    on_chr = 0
    cur = ""
    with open(path) as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] != cur:
                cur = w[0]
                off += on_chr
                on_chr = 0
            b = (w[0], int(w[1]), int(w[2]), off)
            blocks.append(b)
            on_chr += b[2] - b[1] + 1
    return blocks


def load_phen(path: str):
    """phen.cpp:27-74 -> (num_samples, num_phen, column-major float32 data)."""
    rows = []
    with open(path) as f:
        f.readline()
        for line in f:
            w = line.split()
            rows.append([np.float32(np.nan) if x == "NA" else np.float32(float(x)) for x in w[2:]])
    a = np.array(rows, np.float32)
    return a.shape[0], a.shape[1], np.ascontiguousarray(a.T).reshape(-1)


# ---- mps block (cli.cpp:362-411, blocking.cpp, corr_host.cu:65-128) ----------------------------------

def marker_corr_banded(bed, m: int, N: int, width: int) -> np.ndarray:
    """banded Kendall-npn correlations: out[row, col] = npn(row, row + 1 + col), 0 past the last marker"""
    bed = np.ascontiguousarray(bed, np.uint8).reshape(-1)
    out = np.zeros((m, width), np.float32)
    lib().orc_marker_corr_banded(bed, m, N, width, out)
    return out


def banded_row_abs_sums(band: np.ndarray) -> np.ndarray:
    band = np.ascontiguousarray(band, np.float32)
    m, w = band.shape
    out = np.zeros(m, np.float32)
    lib().orc_banded_row_abs_sums(band, m, w, out)
    return out


def hanning_smoothing(v, window_size: int) -> np.ndarray:
    v = np.ascontiguousarray(v, np.float32)
    out = np.zeros(len(v), np.float64)
    lib().orc_hanning_smoothing(v, len(v), window_size, out)
    return out


def block_chr(v, max_block_size: int) -> list[tuple[int, int]]:
    """(first, last) chromosome-local marker indices of the blocks"""
    v = np.ascontiguousarray(v, np.float32)
    cap = max(len(v), 1)
    first = np.zeros(cap, np.int64)
    last = np.zeros(cap, np.int64)
    k = lib().orc_block_chr(v, len(v), max_block_size, first, last, cap)
    assert k >= 0
    return [(int(first[i]), int(last[i])) for i in range(k)]


def make_blocks(bed, chr_ids, N: int, max_block_size: int, corr_width: int) -> list[str]:
    """lines of <bfiles>_m<max_block_size>.blocks for a SNP-major bed matrix (m x ceil(N/4)) and one chromosome id per
    marker (chromosomes in file order, cli.cpp:381-408)"""
    bed = np.ascontiguousarray(bed, np.uint8)
    lines = []
    order = []
    for c in chr_ids:
        if c not in order:
            order.append(c)
    chr_ids = np.asarray(chr_ids)
    for cid in order:
        rows = np.nonzero(chr_ids == cid)[0]
        sub = bed[rows[0]: rows[-1] + 1]
        band = marker_corr_banded(sub, len(rows), N, corr_width)
        sums = banded_row_abs_sums(band)
        for a, b in block_chr(sums, max_block_size):
            lines.append(f"{cid}\t{a}\t{b}")
    return lines

