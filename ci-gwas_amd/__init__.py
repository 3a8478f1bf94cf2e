"""MI355X-native `cusk` PC-skeleton engine (host-side Python mirror).

The product is the C-ABI library built from csrc/ (libcusk_hip.so) and the
`mps`-compatible executable; this package only mirrors the reference's
operator interface and Python CLI (ci-gwas.py cusk / cuskss) and offers ctypes
bindings for tests and bench.  Importing it does not load the library; the
first call does, and fails loudly when the HIP build is missing.
"""
__version__ = "0.1.0"

from .skeleton import (  # noqa: F401
    DeviceArray,
    Engine,
    Skeleton,
    Stats,
    cu_corr_pearson_npn,
    cu_marker_phen_corr_pearson,
    hetcor_skeleton,
    hetcor_threshold,
    threshold_array,
)
