"""CPU: the sepselect restatement (oracle/sepselect_oracle.py) against the files the reference itself wrote
for the seeded merged skeletons of tests/golden/sepselect_kat.json (generator: make_sepselect_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def load_cases():
    with open(os.path.join(GOLDEN, "sepselect_kat.json")) as f:
        return json.load(f)["cases"]


def materialise(case, d):
    """write a case's input files under directory d, return (stem, prior file or None)"""
    stem = os.path.join(d, "all_merged")
    inp = case["input"]
    for suffix, key in ((".mdim", "mdim"), ("_sam.mtx", "sam"), ("_scm.mtx", "scm")):
        with open(stem + suffix, "w") as f:
            f.write(inp[key])
    np.array(inp["ixs"], dtype=np.int32).tofile(stem + ".ixs")
    prior = None
    if inp["prior"] is not None:
        prior = os.path.join(d, "prior.bin")
        np.array(inp["prior"], dtype=np.int32).tofile(prior)
    return stem, prior


def check_outputs(case, ostem):
    """every output file equals the reference's: text verbatim, triples as integers in the same row order"""
    exp = case["output"]
    for suffix, key in ((".mdim", "mdim"), ("_sam.mtx", "sam"), ("_scm.mtx", "scm"), ("_spm.mtx", "spm"), (".ssm", "ssm")):
        with open(ostem + suffix) as f:
            assert f.read() == exp[key], suffix
    assert np.fromfile(ostem + ".ut", dtype=np.int32).tolist() == exp["ut"]
    assert np.fromfile(ostem + ".atr", dtype=np.int32).tolist() == exp["atr"]


@pytest.mark.parametrize("name", ["small", "prior", "collinear", "wide", "dense_traits"])
def test_oracle_reproduces_reference_files(name, tmp_path):
    from oracle import sepselect_oracle as SO

    case = load_cases()[name]
    stem, prior = materialise(case, str(tmp_path))
    res = SO.run(stem, case["alpha"], case["num_samples"], prior)
    assert len(res["min_sepsets"]) == case["pairs_with_minimum"]
    ostem = os.path.join(str(tmp_path), "max_sep_min_pc")
    SO.write(res, ostem)
    check_outputs(case, ostem)


@pytest.mark.parametrize("name", ["small", "collinear", "wide"])
def test_host_graph_logic_matches_reference_files(name, tmp_path):
    """the product's host side up to (not including) the device launch: loader, collinear-marker removal,
    unshielded triples in the reference's row order"""
    from cigwas_amd import sepselect as SS

    case = load_cases()[name]
    stem, prior = materialise(case, str(tmp_path))
    cr = SS.MergedCuskResults(stem, orientation_prior_file=prior)
    exp = case["output"]
    assert cr.get_rfci_relevant_unshielded_triples().ravel().tolist() == exp["ut"]
    assert cr.num_var == int(exp["mdim"].split()[0])
    from oracle import sepselect_oracle as SO

    g = SO.load_merged(stem)
    assert np.array_equal(g["adj"], cr.adj) and np.array_equal(g["corr"], cr.corr) and np.array_equal(g["ixs"], cr.ixs)
    assert SO.unshielded_triples(g["adj"]) == cr.get_unshielded_triples()
    assert list(SO.unshielded_triples(g["adj"])) == list(cr.get_unshielded_triples())  # same iteration order


def test_asymmetric_trait_block_is_refused_before_any_device_work(tmp_path):
    """the device kernel assumes corr[v, t] == corr[t, v]; the host mirror checks instead of guessing"""
    from cigwas_amd import sepselect as SS

    case = load_cases()["small"]
    stem, _ = materialise(case, str(tmp_path))
    cr = SS.MergedCuskResults(stem)
    cr.corr[0, 1] += 0.25
    with pytest.raises(ValueError, match="not symmetric"):
        cr.find_maximal_and_min_pcorr_sepsets_incr(case["alpha"], case["num_samples"])
