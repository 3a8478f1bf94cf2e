#!/usr/bin/env python3
"""Extract the reference's own known-answer data into small committed fixtures.

Run in the build container only (reads /root/reference; the GPU box has no
reference checkout).  Writes DATA, never source text:

  ref_kat.json   numeric arrays parsed out of the reference's gtest fixtures
                 (cusk/include/test_data/*.h) and the expected values that the
                 reference's tests assert (cusk/tests/*_tests.cpp)
  blocking_kat.json  the row-sum vector, smoothed prefix and expected blocks of the reference's blocking tests
  files/         byte copies of the data files under cusk/tests/test_files
                 that the cuskss / io / phen tests read

Usage: python tests/golden/make_golden.py [/root/reference]
"""
import json
import os
import re
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CUSK = os.path.join(REF, "cusk")


def read(rel):
    with open(os.path.join(CUSK, rel)) as f:
        return f.read()


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"//[^\n]*", "", txt)


def array(txt, name):
    """numbers of the brace initialiser of `name` (first definition after comment removal)."""
    m = re.search(r"\b" + re.escape(name) + r"\b[^;{]*\{(.*?)\}\s*;", txt, flags=re.S)
    if not m:
        raise KeyError(name)
    toks = [t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()]
    out = []
    for t in toks:
        t = t.rstrip("fF") if not t.lower().startswith("0x") else t
        out.append(int(t, 16) if t.lower().startswith("0x") else float(t))
    return out


def define(txt, name):
    return int(re.search(r"#define\s+" + name + r"\s+(\d+)", txt).group(1))


def scalar(txt, name):
    return float(re.search(r"\b" + name + r"\s*=\s*([-0-9.eE+]+)", txt).group(1))


kat = {}

# ---- cupc_test_set.h : C_N10 -> A_N10 (cupc_tests.cpp:17-88) -----------------
t = strip_comments(read("include/test_data/cupc_test_set.h"))
kat["cupc_n10"] = {
    "alpha": scalar(t, "ALPHA_N10"),
    "n": int(scalar(t, "N_N10")),
    "sample_size": int(scalar(t, "SAMPLE_SIZE_N10")),
    "A": [int(v) for v in array(t, "A_N10")],
    "C": array(t, "C_N10"),
    "max_level": 14,
}
# cupc_tests.cpp:10-15
kat["threshold"] = {"n": 500000, "alpha": 0.00000001, "th0": 0.0081045, "tol": 0.0001}

# ---- bed_marker_test_set.h + corr_tests.cpp -----------------------------------
t = strip_comments(read("include/test_data/bed_marker_test_set.h"))
ct = strip_comments(read("tests/corr_tests.cpp"))


def test_body(txt, suite, name):
    m = re.search(r"TEST\(\s*" + suite + r"\s*,\s*" + name + r"\s*\)\s*\{", txt)
    i = m.end()
    depth = 1
    while depth:
        depth += {"{": 1, "}": -1}.get(txt[i], 0)
        i += 1
    return txt[m.end(): i]


b = test_body(ct, "cu_corr_pearson_npn", "expected_results")
kat["bmt"] = {
    "num_individuals": define(t, "BMT_NUM_INDIVIDUALS"),
    "num_markers": define(t, "BMT_NUM_MARKERS"),
    "num_phen": define(t, "BMT_NUM_PHEN"),
    "marker_vals": [int(v) for v in array(t, "bmt_marker_vals")],
    "marker_mean": array(t, "bmt_marker_mean"),
    "marker_std": array(t, "bmt_marker_std"),
    "phen_vals": array(t, "bmt_phen_vals"),
    "exp_mxm": array(b, "marker_corr_expected"),
    "exp_mxp": array(b, "marker_phen_corr_expected"),
    "exp_pxp": array(b, "phen_corr_expected"),
    "tol": 0.00001,
}
kat["bmt2"] = {
    "num_individuals": define(t, "BMT2_NUM_INDIVIDUALS"),
    "num_markers": define(t, "BMT2_NUM_MARKERS"),
    "num_phen": define(t, "BMT2_NUM_PHEN"),
    "marker_vals": [int(v) for v in array(t, "bmt2_marker_vals")],
    "marker_mean": array(t, "bmt2_marker_mean"),
    "marker_std": array(t, "bmt2_marker_std"),
    "phen_vals": array(t, "bmt2_phen_vals"),
    "exp_mxm_npn": array(t, "bmt2_marker_corrs"),
    "exp_mxp_pearson": array(t, "bmt2_marker_phen_corrs_pearson"),
    "exp_pxp": array(t, "bmt2_phen_corrs"),
    "tol": 0.00001,
}
b = test_body(ct, "cu_phen_corr_pearson_npn", "expected_results")
kat["with_nan_phen"] = {"file": "with_nan.phen", "exp_pxp": array(b, "phen_corr_expected"), "tol": 0.00001}

# ---- parent_set_test_set.h (parents_tests.cpp) --------------------------------
t = strip_comments(read("include/test_data/parent_set_test_set.h"))
kat["parent_set"] = {
    "num_markers": int(scalar(t, "TEST_NUM_MARKERS")),
    "num_phen": int(scalar(t, "TEST_NUM_PHEN")),
    "adj": [int(v) for v in array(t, "TEST_ADJ_MAT")],
    "d0": [int(v) for v in array(t, "TEST_PAR_SET_D0")],
    "d1": [int(v) for v in array(t, "TEST_PAR_SET_D1")],
    "d2": [int(v) for v in array(t, "TEST_PAR_SET_D2")],
}

# ---- cuskss_tests.cpp : arguments and expected outputs ------------------------
ck = strip_comments(read("tests/cuskss_tests.cpp"))
common = dict(alpha=0.0001, num_samples=500000, max_level_one=3, depth=1,
              mxm="small_mxm.bin", mxp="marker_trait_summary_stats.txt",
              pxp="trait_summary_stats.txt", marker_ixs="marker_indices.bin")
b = test_body(ck, "cuskss", "trait_only_merged_expected_results")
kat["cuskss_trait_only"] = dict(common, max_level_two=0, merged=True, trait_only=True, stem="trait_only",
                                exp_adj=[int(v) for v in array(b, "exp_adj")], exp_corr=array(b, "exp_corr"), tol=0.001)
b = test_body(ck, "cuskss", "pearson_two_stage_merged_expected_results")
kat["cuskss_two_stage_merged"] = dict(common, max_level_two=1, merged=True, trait_only=False, stem="cuskss_merged",
                                      exp_ixs=[int(v) for v in array(b, "exp_ixs")],
                                      exp_adj=[int(v) for v in array(b, "exp_adj")], exp_corr=array(b, "exp_corr"), tol=0.001)
b = test_body(ck, "cuskss", "pearson_two_stage_block_expected_results")
kat["cuskss_two_stage_block"] = dict(common, max_level_two=1, merged=False, trait_only=False, stem="1_0_2",
                                     blocks="blocks.txt", block_index=0,
                                     exp_ixs=[int(v) for v in array(b, "exp_ixs")],
                                     exp_adj=[int(v) for v in array(b, "exp_adj")], exp_corr=array(b, "exp_corr"), tol=0.001)

# ---- prep_tests.cpp:46-68 (mps prep on tests/test_files/small.*) ---------------------------------------------
pt = strip_comments(read("tests/prep_tests.cpp"))
b = test_body(pt, "ParseBed", "CorrectOutFilesGenerated")
kat["prep_small"] = {"stem": "small", "exp_stds": array(b, "one_stds_exp"), "exp_means": array(b, "one_means_exp"),
                     "exp_dims": [int(v) for v in re.search(r"BedDims\s+exp\(\s*(\d+)\s*,\s*(\d+)\s*\)", b).groups()],
                     "tol": 0.000001}

with open(os.path.join(HERE, "ref_kat.json"), "w") as f:
    json.dump(kat, f, indent=1)

# ---- blocking_test_set.h + blocking_tests.cpp (mps block): own file, it is 5000 numbers -----------
t = strip_comments(read("include/test_data/blocking_test_set.h"))
bt = strip_comments(read("tests/blocking_tests.cpp"))
b = test_body(bt, "block_chr", "expected_results_synthetic_data")
blocking = {
    "test_v": array(t, "TEST_V"),                       # forward correlation row sums (float)
    "test_v_smooth": array(t, "TEST_V_SMOOTH"),         # first 1000 entries smoothed with window 101, tolerance 0.01
    "smooth_window": 101,
    "smooth_tol": 0.01,
    "chr": "1",
    "max_block_size": 500,
    "exp_blocks": [[int(a), int(c)] for a, c in re.findall(r'MarkerBlock\(\s*"1"\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*0\s*\)', b)],
}
with open(os.path.join(HERE, "blocking_kat.json"), "w") as f:
    json.dump(blocking, f)

os.makedirs(os.path.join(HERE, "files"), exist_ok=True)
for fn in ["small_mxm.bin", "marker_indices.bin", "marker_trait_summary_stats.txt", "trait_summary_stats.txt",
           "blocks.txt", "time_index.txt", "with_nan.phen", "small.bed", "small.bim", "small.fam", "small.phen",
           "test.blocks"]:
    dst = os.path.join(HERE, "files", fn)
    shutil.copyfile(os.path.join(CUSK, "tests/test_files", fn), dst)
    os.chmod(dst, 0o644)
print("wrote", os.path.join(HERE, "ref_kat.json"), "and files/")
