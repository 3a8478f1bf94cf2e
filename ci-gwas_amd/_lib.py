"""ctypes loader of libcusk_hip.so (the product).  No fallback: if the HIP
library is missing or cannot be loaded this raises, it never routes anywhere else."""
from __future__ import annotations

import ctypes as C
import os

# The engine overlaps independent kernels on two HIP streams.  HIP multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL has created its own streams in the same process the
# engine's two end up sharing one queue and serialise (measured: +13 % per block).  Read at HIP initialisation, so it
# has to be in the environment before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libcusk_hip.so")
ML = 14


class CuskStats(C.Structure):
    _fields_ = [
        ("level", C.c_int),
        ("levels_run", C.c_int),
        ("max_degree", C.c_int * (ML + 1)),
        ("edges", C.c_longlong * (ML + 1)),
        ("tests", C.c_longlong * (ML + 1)),
        ("subsets", C.c_longlong * (ML + 1)),
        ("removed", C.c_longlong * (ML + 1)),
        ("kernel_ms", C.c_float * (ML + 1)),
        ("level_ms", C.c_float * (ML + 1)),
        ("total_ms", C.c_float),
        ("rechecks", C.c_longlong * (ML + 1)),
        ("violations", C.c_longlong),
        ("exact_fallbacks", C.c_longlong),
        ("main_kernel_ms", C.c_float * (ML + 1)),
        ("canonical_tests", C.c_longlong * (ML + 1)),
    ]


class CuskBatchStats(C.Structure):
    """cusk_batch_stats of include/cusk_hip.h"""
    _fields_ = [
        ("blocks", C.c_int),
        ("skipped", C.c_int),
        ("markers", C.c_longlong),
        ("retained", C.c_longlong),
        ("vars_stage1", C.c_longlong),
        ("vars_stage2", C.c_longlong),
        ("tests", C.c_longlong * 2),
        ("canonical", C.c_longlong * 2),
        ("ms_corr", C.c_double),
        ("ms_stage1", C.c_double),
        ("ms_prune", C.c_double),
        ("ms_stage2", C.c_double),
        ("ms_reduce", C.c_double),
        ("stage", CuskStats * 2),
    ]


class CuskBlockStats(C.Structure):
    """cusk_block_stats of include/cusk_hip.h"""
    _fields_ = [
        ("skipped", C.c_int),
        ("num_sig", C.c_int),
        ("markers", C.c_longlong),
        ("retained", C.c_longlong),
        ("tests", C.c_longlong * 2),
        ("ms_inputs", C.c_double),
        ("ms_corr", C.c_double),
        ("ms_stage1", C.c_double),
        ("ms_prune", C.c_double),
        ("ms_stage2", C.c_double),
        ("ms_reduce", C.c_double),
        ("stage", CuskStats * 2),
    ]


# every symbol include/cusk_hip.h declares: (restype, argtypes)
_vp, _i, _f, _sz, _ll = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong
SYMBOLS = {
    "Skeleton": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hetcor_skeleton": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cusk_threshold_array": (None, [_i, _f, _vp]),
    "cusk_hetcor_threshold": (_f, [_f]),
    "cu_marker_phen_corr_pearson": (None, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp]),
    "cu_corr_pearson_npn": (None, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "cusk_engine_create": (_i, [C.POINTER(_vp), _i, _vp]),
    "cusk_engine_destroy": (None, [_vp]),
    "cusk_last_error": (C.c_char_p, [_vp]),
    "cusk_engine_set_option": (_i, [_vp, C.c_char_p, _ll]),
    "cusk_engine_stream": (_vp, [_vp]),
    "cusk_engine_bind_thread": (_i, [_vp]),
    "cusk_engine_device": (_i, [_vp]),
    "cusk_engine_set_row_shard": (_i, [_vp, _i, _i, _vp, _vp, _i]),
    "cusk_run_skeleton": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(CuskStats)]),
    "cusk_run_hetcor": (_i, [_vp, _vp, _vp, _f, _vp, _i, _f, _i, _vp, C.POINTER(CuskStats)]),
    "cusk_run_skeleton_batch": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, C.POINTER(CuskStats)]),
    "cusk_result_adj_bits_blocks": (_i, [_vp, _vp]),
    "cusk_result_adj_bits_blocks_tail": (_i, [_vp, _i, _vp]),
    "cusk_result_adj_rows": (_i, [_vp, _i, _i, _vp]),
    "cusk_gather_rows": (_i, [_vp, _vp, _i, _vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _i]),
    "cusk_result_n": (_i, [_vp]),
    "cusk_result_adj_bits_dev": (_vp, [_vp]),
    "cusk_result_words": (_i, [_vp]),
    "cusk_result_adj_i32": (_i, [_vp, _vp]),
    "cusk_result_adj_i32_dev": (_i, [_vp, _vp]),
    "cusk_result_pmax": (_i, [_vp, _vp, _vp]),
    "cusk_result_sepset_dense": (_i, [_vp, _vp]),
    "cusk_result_sepsets": (_ll, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "cusk_result_sepsets_view": (_ll, [_vp, _vp, _vp, _vp]),
    "cusk_corr_build": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp]),
    "cusk_corr_build_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "cusk_corr_build_batch_mxp": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "cusk_corr_build_batch_mxm": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "cusk_corr_build_begin": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp]),
    "cusk_corr_build_end": (_i, [_vp, _vp]),
    "cusk_corr_timing": (None, [_vp, _vp]),
    "cusk_corr_banded": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "cusk_hanning_smooth": (_i, [_vp, _vp, _sz, _vp, _i, _vp]),
    "cusk_sepselect_greedy": (_i, [_vp, _vp, C.c_longlong, _i, C.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "cusk_gather_submatrix": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "cusk_gather_submatrix_dev": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "cusk_blockset_open": (_i, [C.POINTER(_vp), C.c_char_p, C.c_char_p, C.c_char_p, _f, _i, _i, _i, C.c_char_p, _sz]),
    "cusk_blockset_close": (None, [_vp]),
    "cusk_blockset_num_blocks": (_i, [_vp]),
    "cusk_blockset_num_samples": (_ll, [_vp]),
    "cusk_blockset_num_phen": (_i, [_vp]),
    "cusk_blockset_block_markers": (_ll, [_vp, _i]),
    "cusk_blockset_block_stem": (_i, [_vp, _i, C.c_char_p, _sz]),
    "cusk_blockset_stage": (_i, [_vp, _vp]),
    "cusk_blockset_run_block": (_i, [_vp, _vp, _i, C.POINTER(_vp), C.POINTER(CuskBlockStats)]),
    "cusk_blockset_run_block_next": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp), C.POINTER(CuskBlockStats)]),
    "cusk_blockset_last_error": (C.c_char_p, []),
    "cusk_blockset_release_engine": (None, [_vp, _vp]),
    "cusk_blockset_run_batch": (_i, [_vp, _vp, _vp, _i, C.POINTER(_vp), _vp]),
    "cusk_batch_result_count": (_i, [_vp]),
    "cusk_batch_result_block_index": (_i, [_vp, _i]),
    "cusk_batch_result_block": (_vp, [_vp, _i]),
    "cusk_batch_result_write": (_i, [_vp, C.c_char_p]),
    "cusk_batch_result_packed_bytes": (_sz, [_vp]),
    "cusk_batch_result_pack": (_i, [_vp, _vp, _sz]),
    "cusk_packed_results_write": (_i, [_vp, _sz, C.c_char_p, _vp]),
    "cusk_batch_result_packed_bytes_ex": (_sz, [_vp, _i]),
    "cusk_batch_result_pack_ex": (_i, [_vp, _vp, _sz, _i]),
    "cusk_merge_packed": (_i, [C.c_char_p, _vp, _sz, C.c_char_p]),
    "cusk_batch_result_free": (None, [_vp]),
    "cusk_corr_build_pending": (_i, [_vp]),
    "cusk_block_result_dims": (None, [_vp, C.POINTER(_ll), C.POINTER(_ll), C.POINTER(_ll)]),
    "cusk_block_result_stem": (C.c_char_p, [_vp]),
    "cusk_block_result_ixs": (_vp, [_vp]),
    "cusk_block_result_adj": (_vp, [_vp]),
    "cusk_block_result_corr": (_vp, [_vp]),
    "cusk_block_result_sep": (_vp, [_vp]),
    "cusk_block_result_write": (_i, [_vp, C.c_char_p]),
    "cusk_block_result_free": (None, [_vp]),
    "cusk_engine_download": (_i, [_vp, _vp, _vp, _sz]),
    "cusk_dev_alloc": (_vp, [_sz]),
    "cusk_dev_free": (None, [_vp]),
    "cusk_dev_upload": (_i, [_vp, _vp, _sz]),
    "cusk_dev_download": (_i, [_vp, _vp, _sz]),
}

# cusk_exchange_fn of include/cusk_hip.h
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C ci-gwas_amd/csrc).  There is no CPU fallback."
            )
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
