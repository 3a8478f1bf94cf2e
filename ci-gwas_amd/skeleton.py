"""Host-side mirror of the reference's operator interface for the cusk path.

`Skeleton`, `hetcor_skeleton`, `threshold_array`, `hetcor_threshold`,
`cu_corr_pearson_npn` and `cu_marker_phen_corr_pearson` take and return host
numpy arrays with the reference's argument meaning
(/root/reference/cusk/include/mps/cuPC-S.h:196, hetcor-cuPC-S.h:46,
cuPC_call_prep.h:7-15, corr_host.h:38-47,92-103) and call straight through the
C ABI of libcusk_hip.so.  `Engine` is the device-resident API used by bench.py
and the mps host program (matrix stays in HBM, sparse sepsets).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import ML, CuskStats, lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def threshold_array(n: int, alpha: float) -> np.ndarray:
    out = np.zeros(ML + 1, np.float32)
    lib().cusk_threshold_array(int(n), float(np.float32(alpha)), _ptr(out))
    return out


def hetcor_threshold(alpha: float) -> float:
    return float(lib().cusk_hetcor_threshold(float(np.float32(alpha))))


def Skeleton(Cm: np.ndarray, Th: np.ndarray, maxlevel: int, want_pmax: bool = True, want_sepset: bool = True):
    """-> (G n*n int32, level, pMax n*n float32 | None, SepSet n*n*14 int32 | None)."""
    Cm = np.ascontiguousarray(Cm, np.float32)
    n = Cm.shape[0]
    G = np.ones((n, n), np.int32)
    pmax = np.zeros((n, n), np.float32) if want_pmax else None
    sep = np.zeros((n, n, ML), np.int32) if want_sepset else None
    Th = np.ascontiguousarray(Th, np.float32)
    P, l, ml = C.c_int(n), C.c_int(0), C.c_int(int(maxlevel))
    lib().Skeleton(_ptr(Cm), C.addressof(P), _ptr(G), _ptr(Th), C.addressof(l), C.addressof(ml), _ptr(pmax), _ptr(sep))
    return G, l.value, pmax, sep


def hetcor_skeleton(Cm, G, N, th: float, maxlevel: int, time_index):
    """-> (G n*n int32 (copy, updated), level)."""
    Cm = np.ascontiguousarray(Cm, np.float32)
    n = Cm.shape[0]
    G = np.array(G, np.int32).reshape(n, n).copy()
    N = np.ascontiguousarray(N, np.float32).reshape(n, n)
    ti = np.ascontiguousarray(time_index, np.int32)
    P, l, ml, thv = C.c_int(n), C.c_int(0), C.c_int(int(maxlevel)), C.c_float(float(np.float32(th)))
    lib().hetcor_skeleton(_ptr(Cm), C.addressof(P), _ptr(G), _ptr(N), C.addressof(thv), C.addressof(l), C.addressof(ml), _ptr(ti))
    return G, l.value


def cu_marker_phen_corr_pearson(bed, phen, m, N, p, means, stds) -> np.ndarray:
    bed = np.ascontiguousarray(bed, np.uint8)
    phen = np.ascontiguousarray(phen, np.float32)
    means = np.ascontiguousarray(means, np.float32)
    stds = np.ascontiguousarray(stds, np.float32)
    out = np.zeros(m * p, np.float32)
    lib().cu_marker_phen_corr_pearson(_ptr(bed), _ptr(phen), m, N, p, _ptr(means), _ptr(stds), _ptr(out))
    return out


def cu_corr_pearson_npn(bed, phen, m, N, p, means, stds):
    bed = np.ascontiguousarray(bed, np.uint8)
    phen = np.ascontiguousarray(phen, np.float32)
    means = np.ascontiguousarray(means, np.float32)
    stds = np.ascontiguousarray(stds, np.float32)
    mxm = np.zeros(max(m * (m - 1) // 2, 1), np.float32)
    mxp = np.zeros(max(m * p, 1), np.float32)
    pxp = np.zeros(max(p * (p - 1) // 2, 1), np.float32)
    lib().cu_corr_pearson_npn(_ptr(bed), _ptr(phen), m, N, p, _ptr(means), _ptr(stds), _ptr(mxm), _ptr(mxp), _ptr(pxp))
    return mxm[: m * (m - 1) // 2], mxp[: m * p], pxp[: p * (p - 1) // 2]


@dataclass
class Stats:
    level: int
    levels_run: int
    max_degree: list
    edges: list
    tests: list
    subsets: list
    removed: list
    kernel_ms: list
    level_ms: list
    total_ms: float
    rechecks: list
    violations: int
    exact_fallbacks: int
    main_kernel_ms: list
    canonical_tests: list

    @staticmethod
    def of(s: CuskStats) -> "Stats":
        return Stats(s.level, s.levels_run, list(s.max_degree), list(s.edges), list(s.tests), list(s.subsets),
                     list(s.removed), list(s.kernel_ms), list(s.level_ms), float(s.total_ms), list(s.rechecks),
                     int(s.violations), int(s.exact_fallbacks), list(s.main_kernel_ms), list(s.canonical_tests))


class DeviceArray:
    """n-byte HBM allocation owned by the library (no torch needed)."""

    def __init__(self, host: np.ndarray | None = None, nbytes: int | None = None):
        self.nbytes = int(host.nbytes if host is not None else nbytes)
        self.ptr = lib().cusk_dev_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError(f"cusk_dev_alloc({self.nbytes}) failed")
        if host is not None:
            h = np.ascontiguousarray(host)
            if lib().cusk_dev_upload(self.ptr, _ptr(h), h.nbytes) != 0:
                raise RuntimeError("upload failed")

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        if lib().cusk_dev_download(_ptr(out), self.ptr, out.nbytes) != 0:
            raise RuntimeError("download failed")
        return out

    def free(self):
        if self.ptr:
            lib().cusk_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """Device-resident engine (cusk_engine_* of include/cusk_hip.h)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        rc = lib().cusk_engine_create(C.byref(h), int(device), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise RuntimeError(f"cusk_engine_create failed with code {rc} (no MI355X / HIP device?)")
        self.h = h

    def close(self):
        if self.h:
            lib().cusk_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libcusk_hip error {rc}: {lib().cusk_last_error(self.h).decode()}")

    def set_option(self, key: str, value: int) -> None:
        self._check(lib().cusk_engine_set_option(self.h, key.encode(), int(value)))

    def set_row_shard(self, rank: int, world: int, exchange=None, host_staging: bool = True) -> None:
        """Row-sharded sweep of one block over `world` engines (cusk_engine_set_row_shard).  `exchange(level, buf,
        count, elem_bytes, on_device, stream) -> int` must all-reduce the buffer with an element-wise unsigned MIN
        (see ci-gwas_amd/shard.py: make_min_exchange)."""
        from ._lib import EXCHANGE_FN

        if exchange is None:
            self._exchange_cb = None
            self._check(lib().cusk_engine_set_row_shard(self.h, 0, 1, None, None, 0))
            return

        def trampoline(_user, level, buf, count, elem_bytes, on_device, stream):
            try:
                return int(exchange(int(level), int(buf), int(count), int(elem_bytes), bool(on_device), stream))
            except Exception as exc:  # an exception must not unwind through the C frames
                import traceback

                traceback.print_exc()
                self._exchange_error = exc
                return 1

        self._exchange_cb = EXCHANGE_FN(trampoline)  # keep the thunk alive as long as the engine uses it
        self._check(lib().cusk_engine_set_row_shard(self.h, int(rank), int(world),
                                                    C.cast(self._exchange_cb, C.c_void_p), None, 1 if host_staging else 0))

    @property
    def stream(self) -> int:
        return int(lib().cusk_engine_stream(self.h) or 0)

    def run_skeleton(self, C_dev: int, n: int, Th, maxlevel: int) -> Stats:
        Th = np.ascontiguousarray(Th, np.float32)
        st = CuskStats()
        self._check(lib().cusk_run_skeleton(self.h, C_dev, n, _ptr(Th), int(maxlevel), C.byref(st)))
        return Stats.of(st)

    def run_skeleton_batch(self, C_dev: int, n: int, lo, hi, Th, maxlevel: int) -> Stats:
        """cusk_run_skeleton_batch: blocks [lo[b], hi[b]) on the diagonal of one n x n allocation, swept in one run"""
        Th = np.ascontiguousarray(Th, np.float32)
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        st = CuskStats()
        self._check(lib().cusk_run_skeleton_batch(self.h, C_dev, int(n), len(lo), _ptr(lo), _ptr(hi), _ptr(Th), int(maxlevel),
                                                  C.byref(st)))
        self._batch = (lo.copy(), hi.copy())
        return Stats.of(st)

    def adjacency_blocks(self) -> list:
        """per block of the last batched run its k x k int32 adjacency (cusk_result_adj_bits_blocks)"""
        lo, hi = self._batch
        k = (hi - lo).astype(np.int64)
        wb = (k + 63) // 64
        out = np.zeros(int((k * wb).sum()), np.uint64)
        self._check(lib().cusk_result_adj_bits_blocks(self.h, _ptr(out)))
        res, o = [], 0
        for kb, w in zip(k, wb):
            bits = out[o:o + kb * w].reshape(kb, w)
            o += kb * w
            G = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :kb].astype(np.int32)
            res.append(G)
        return res

    def gather_rows(self, M_dev: int, n: int, idx, row_src, row_k, row_first, row_out, out_dev: int | None = None,
                    out_count: int = 0):
        idx = np.ascontiguousarray(idx, np.int32)
        row_src = np.ascontiguousarray(row_src, np.int32)
        row_k = np.ascontiguousarray(row_k, np.int32)
        row_first = np.ascontiguousarray(row_first, np.int64)
        row_out = np.ascontiguousarray(row_out, np.int64)
        host = None if out_dev is not None else np.zeros(int(out_count), np.float32)
        self._check(lib().cusk_gather_rows(self.h, M_dev, int(n), _ptr(idx), len(idx), _ptr(row_src), _ptr(row_k), _ptr(row_first),
                                           _ptr(row_out), len(row_src), out_dev if out_dev is not None else _ptr(host),
                                           int(out_count), 1 if out_dev is not None else 0))
        return host

    def run_hetcor(self, C_dev: int, n: int, th: float, maxlevel: int, N_dev: int | None = None,
                   ess_uniform: float = 0.0, G_init_dev: int | None = None, time_index=None) -> Stats:
        ti = np.ascontiguousarray(time_index, np.int32) if time_index is not None else None
        st = CuskStats()
        self._check(lib().cusk_run_hetcor(self.h, C_dev, N_dev, float(ess_uniform), G_init_dev, n,
                                          float(np.float32(th)), int(maxlevel), _ptr(ti), C.byref(st)))
        return Stats.of(st)

    def adjacency(self) -> np.ndarray:
        n = lib().cusk_result_n(self.h)
        G = np.zeros((n, n), np.int32)
        self._check(lib().cusk_result_adj_i32(self.h, _ptr(G)))
        return G

    def adjacency_bits(self) -> np.ndarray:
        n, w = lib().cusk_result_n(self.h), lib().cusk_result_words(self.h)
        out = np.zeros((n, w), np.uint64)
        self._check(lib().cusk_dev_download(_ptr(out), lib().cusk_result_adj_bits_dev(self.h), out.nbytes))
        return out

    def pmax(self, C_dev: int) -> np.ndarray:
        n = lib().cusk_result_n(self.h)
        out = np.zeros((n, n), np.float32)
        self._check(lib().cusk_result_pmax(self.h, C_dev, _ptr(out)))
        return out

    def sepsets(self):
        """-> (x, y, level, z, S[count,14]) sparse records, sorted by (x, y)."""
        cnt = lib().cusk_result_sepsets(self.h, None, None, None, None, None)
        if cnt < 0:
            raise RuntimeError("no Skeleton result")
        x, y, lv = (np.zeros(cnt, np.int32) for _ in range(3))
        z = np.zeros(cnt, np.float32)
        S = np.full((cnt, ML), -1, np.int32)
        if cnt:
            lib().cusk_result_sepsets(self.h, _ptr(x), _ptr(y), _ptr(lv), _ptr(z), _ptr(S))
            o = np.lexsort((y, x))
            x, y, lv, z, S = x[o], y[o], lv[o], z[o], S[o]
        return x, y, lv, z, S

    def corr_build(self, bed, phen, m, N, p, means, stds, C_dev: int, want_mxp: bool = False):
        bed = np.ascontiguousarray(bed, np.uint8)
        phen = np.ascontiguousarray(phen, np.float32)
        means = np.ascontiguousarray(means, np.float32)
        stds = np.ascontiguousarray(stds, np.float32)
        mxp = np.zeros(m * p, np.float32) if want_mxp else None
        self._check(lib().cusk_corr_build(self.h, _ptr(bed), _ptr(phen), m, N, p, _ptr(means), _ptr(stds), C_dev, _ptr(mxp)))
        return mxp

    def corr_banded(self, bed, m: int, N: int, width: int, want_band: bool = False):
        """`mps block`'s device part for one chromosome: forward row sums of |banded Kendall-npn correlations|
        (and the band itself, m x width, on request)"""
        bed = np.ascontiguousarray(bed, np.uint8)
        sums = np.zeros(m, np.float32)
        band = np.zeros((m, width), np.float32) if want_band else None
        self._check(lib().cusk_corr_banded(self.h, _ptr(bed), m, N, width, _ptr(sums), _ptr(band)))
        return (sums, band) if want_band else sums

    def sepselect_greedy(self, trait_corr, pair_i, pair_j, pair_corr, cand_off, cand, thr):
        """Batched greedy separating-set selection (sepselect.py:262-329) on the device.  Returns
        (sel, sel_len, flags, kernel_ms); see include/cusk_hip.h for the layouts."""
        trait_corr = np.ascontiguousarray(trait_corr, np.float64)
        n, p = trait_corr.shape
        pair_i = np.ascontiguousarray(pair_i, np.int32)
        pair_j = np.ascontiguousarray(pair_j, np.int32)
        pair_corr = np.ascontiguousarray(pair_corr, np.float64)
        cand_off = np.ascontiguousarray(cand_off, np.int64)
        cand = np.ascontiguousarray(cand, np.int32)
        thr = np.ascontiguousarray(thr, np.float64)
        npairs = pair_i.shape[0]
        assert pair_j.shape[0] == npairs and pair_corr.shape[0] == npairs and cand_off.shape[0] == npairs + 1
        assert cand.shape[0] == int(cand_off[-1])
        sel = np.full(max(cand.shape[0], 1), -1, np.int32)
        sel_len = np.zeros(max(npairs, 1), np.int32)
        flags = np.zeros(max(npairs, 1), np.int32)
        ms = np.zeros(1, np.float32)
        self._check(lib().cusk_sepselect_greedy(self.h, _ptr(trait_corr), n, p, npairs, _ptr(pair_i), _ptr(pair_j),
                                                _ptr(pair_corr), _ptr(cand_off), _ptr(cand), _ptr(thr), thr.shape[0],
                                                _ptr(sel), _ptr(sel_len), _ptr(flags), _ptr(ms)))
        return sel, sel_len[:npairs], flags[:npairs], float(ms[0])

    def corr_timing(self):
        t = np.zeros(4, np.float32)
        lib().cusk_corr_timing(self.h, _ptr(t))
        return t
