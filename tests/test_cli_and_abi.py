"""CPU: the Python CLI shim marshals argv exactly like the reference's ci-gwas.py, and the C-ABI
library loads and exports every symbol include/cusk_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cusk_argv_matches_reference_order():
    from cigwas_amd import cli

    a = cli.build_parser().parse_args(["cusk", "3", "b.blocks", "stem", "y.phen", "0.0001", "3", "14", "1", "out/"])
    assert cli.cusk_argv(a)[1:] == ["cusk", "y.phen", "stem", "b.blocks", "0.0001", "3", "14", "1", "out/", "3"]


def test_block_argv_matches_reference_order():
    """ci-gwas.py:390-401"""
    from cigwas_amd import cli

    args = cli.build_parser().parse_args(["block", "stem", "9000", "40", "1500"])
    assert cli.block_argv(args) == [cli.MPS_PATH, "block", "stem", "9000", "40", "1500"]


def test_cuskss_argv_null_sentinels_and_aliases():
    from cigwas_amd import cli

    p = cli.build_parser()
    # the three scenarios of the reference's cuskss_tests.cpp
    a = p.parse_args(["cuskss", "--pxp", "pxp.txt", "--marker-indices", "ix.bin", "--alpha", "0.0001", "--num-samples",
                      "500000", "--max-level-one", "3", "--max-level-two", "0", "--outdir", "o"])
    assert cli.cuskss_argv(a)[1:] == ["cuskss", "NULL", "NULL", "NULL", "pxp.txt", "NULL", "NULL", "0", "NULL", "ix.bin",
                                      "0.0001", "3", "0", "1", "500000", "o"]
    a = p.parse_args(["cuskss-merged", "--mxm", "m.bin", "--mxp", "mxp.txt", "--pxp", "pxp.txt", "--marker-indices", "ix.bin",
                      "--alpha", "0.0001", "--num-samples", "500000", "--max-level-two", "1", "--outdir", "o"])
    assert cli.cuskss_argv(a)[1:] == ["cuskss", "m.bin", "mxp.txt", "NULL", "pxp.txt", "NULL", "NULL", "0", "NULL", "ix.bin",
                                      "0.0001", "3", "1", "1", "500000", "o"]
    a = p.parse_args(["cuskss", "--mxm", "m.bin", "--mxp", "mxp.txt", "--pxp", "pxp.txt", "--blockfile", "b.txt",
                      "--block-index", "2", "--time-index", "t.txt", "--alpha", "1e-4", "--num-samples", "10"])
    assert cli.cuskss_argv(a)[1:] == ["cuskss", "m.bin", "mxp.txt", "NULL", "pxp.txt", "NULL", "t.txt", "2", "b.txt", "NULL",
                                      "0.0001", "3", "14", "1", "10", "./"]


@pytest.mark.parametrize("argv", [
    ["cuskss", "--pxp", "p", "--alpha", "0.1", "--num-samples", "5"],  # neither blockfile nor marker indices
    ["cuskss", "--pxp", "p", "--blockfile", "b", "--mxp-se", "s", "--alpha", "0.1", "--num-samples", "5"],  # one se file only
    ["cuskss", "--pxp", "p", "--blockfile", "b", "--mxm", "m", "--alpha", "0.1", "--num-samples", "5"],  # mxm without mxp
    ["cuskss-het", "--pxp", "p", "--blockfile", "b", "--alpha", "0.1", "--num-samples", "5"],
    ["cuskss-merged", "--pxp", "p", "--blockfile", "b", "--alpha", "0.1", "--num-samples", "5"],
])
def test_cuskss_validation_exits(argv):
    from cigwas_amd import cli

    a = cli.build_parser().parse_args(argv)
    with pytest.raises(SystemExit):
        cli.cuskss_argv(a)


def test_range_checks():
    from cigwas_amd import cli

    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["cuskss", "--pxp", "p", "--blockfile", "b", "--alpha", "0.1", "--num-samples", "5",
                                       "--max-level-one", "15"])


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cusk_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"\btypedef\b[^;]*;", "", txt)  # function-pointer typedefs are types, not exports
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", txt)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "ci-gwas_amd", "csrc", "libcusk_hip.so")
    if not os.path.exists(so):
        pytest.skip("libcusk_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(so)
    declared = _declared_symbols()
    assert "Skeleton" in declared and "hetcor_skeleton" in declared and "cusk_run_skeleton" in declared
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    from cigwas_amd._lib import SYMBOLS

    assert sorted(SYMBOLS) == declared


def test_thresholds_match_oracle(oracle):
    import numpy as np

    import cigwas_amd as cg

    so = os.path.join(ROOT, "ci-gwas_amd", "csrc", "libcusk_hip.so")
    if not os.path.exists(so):
        pytest.skip("libcusk_hip.so not built")
    for n, a in [(10000, 1e-5), (16384, 1e-4), (458747, 1e-4), (500, 0.05), (50, 1e-2), (500000, 1e-8)]:
        assert np.array_equal(cg.threshold_array(n, a), oracle.threshold_array(n, a))
        assert cg.hetcor_threshold(a) == oracle.hetcor_threshold(a)


def test_sepselect_subcommands_match_reference_arguments():
    """ci-gwas.py:303-358: positional stem, alpha, num-samples; --orientation-prior only on orient-v-structs"""
    from cigwas_amd import cli

    p = cli.build_parser()
    a = p.parse_args(["orient-v-structs", "out/all_merged", "0.0001", "458747", "--orientation-prior", "prior.bin"])
    assert (a.cusk_result_stem, a.alpha, a.num_samples, a.orientation_prior) == ("out/all_merged", 1e-4, 458747, "prior.bin")
    assert a.func is cli.run_v_struct
    a = p.parse_args(["sepselect", "out/all_merged", "0.001", "1000"])
    assert a.func is cli.run_sepselect and not hasattr(a, "orientation_prior")
    with pytest.raises(SystemExit):
        p.parse_args(["sepselect", "out/all_merged", "1.5", "1000"])  # alpha outside (0, 1)


def test_prep_bed_subcommand():
    """ci-gwas.py:54-61, :386-387"""
    from cigwas_amd import cli

    a = cli.build_parser().parse_args(["prep-bed", "data/chr"])
    assert a.bfiles == "data/chr" and a.func is cli.prep_bed


@pytest.mark.skipif(not os.path.isdir("/root/reference/cusk/include/mps"), reason="needs the reference's headers (build container only)")
def test_reference_headers_link_against_the_library_unchanged(tmp_path):
    """A TU that includes the reference's <mps/corr_host.h> and <mps/cuPC_call_prep.h> links against libcusk_hip.so:
    the C++-linkage names cu_corr_pearson_npn, cu_marker_phen_corr_pearson, threshold_array, hetcor_threshold,
    std_normal_qnorm resolve with the reference's signatures, Skeleton / hetcor_skeleton with C linkage."""
    import subprocess

    lib_dir = os.path.join(ROOT, "ci-gwas_amd", "csrc")
    exe = str(tmp_path / "link_ref")
    r = subprocess.run(["g++", "-std=c++20", "-I/root/reference/cusk/include", os.path.join(ROOT, "tests", "link_ref", "link_ref_headers.cpp"),
                        "-L" + lib_dir, "-lcusk_hip", "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    size, t0, het, q = out.stdout.split()[:4]
    assert int(size) == 15 and abs(float(t0) - 0.0081045) < 1e-4  # the reference's own KAT (cupc_tests.cpp:10-15)
    assert abs(float(het) - 3.8905919) < 1e-5 and abs(float(q) + 1.959964) < 1e-5
    assert out.stdout.split()[4] == "4"
