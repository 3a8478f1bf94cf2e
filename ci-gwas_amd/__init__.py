"""MI355X-native `cusk` PC-skeleton engine (host-side Python mirror).

The product is the C-ABI library built from csrc/ (libcusk_hip.so) and the
`mps`-compatible executable; this package only mirrors the reference's Python
CLI (ci-gwas.py cusk / cuskss) and offers ctypes bindings for tests and bench.
"""
__version__ = "0.1.0"
