#!/usr/bin/env python3
"""Golden vectors for the sepselect path: seeded merged skeletons and the files the
REFERENCE ITSELF writes for them.

Run in the build container only: imports /root/reference/cusk_postprocessing/sepselect.py (never copied,
never shipped) and calls `orient_v_structures_merged(stem, alpha, n).to_file(out)` -- the working entry of
ci-gwas.py (`orient-v-structs`, ci-gwas.py:473-476; the plain `sepselect` entry runs the same selection but its
writer stops at `_spm.mtx` because no PAG was made).  Writes DATA only:

  sepselect_kat.json  per case: the input files' text / integers, alpha, num_samples, the orientation prior,
                      and every output file (text files verbatim, .ut/.atr as integer lists)

Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_sepselect_golden.py [/root/reference]
"""
import json
import os
import sys
import tempfile

import numpy as np
import scipy.sparse as sp
from scipy.io import mmwrite

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cusk_postprocessing import sepselect as ref  # noqa: E402

from cigwas_amd.synth import merged_skeleton  # noqa: E402  (the same generator the tests use)

CASES = [
    # name, seed, traits, markers, alpha, N, prior?, duplicate markers
    ("small", 11, 6, 10, 1e-4, 10000, False, 0),
    ("prior", 12, 8, 24, 1e-4, 20000, True, 0),
    ("collinear", 13, 7, 18, 1e-3, 5000, False, 2),
    ("wide", 14, 14, 40, 1e-4, 50000, True, 1),
    ("dense_traits", 15, 10, 12, 1e-2, 2000, False, 0),
]


def text(path):
    with open(path) as f:
        return f.read()


out = {"generator": "tests/golden/make_sepselect_golden.py", "cases": {}}
for name, seed, p, m, alpha, N, with_prior, dup in CASES:
    adj, corr, ixs, prior = merged_skeleton(seed, p, m, duplicates=dup, with_prior=with_prior)
    n = p + m
    with tempfile.TemporaryDirectory() as d:
        stem = os.path.join(d, "all_merged")
        mmwrite(stem + "_sam.mtx", sp.coo_matrix(adj.astype(np.int32)))
        mmwrite(stem + "_scm.mtx", sp.coo_matrix(corr))
        with open(stem + ".mdim", "w") as f:
            f.write(f"{n}\t{p}\t3\n")
        ixs.tofile(stem + ".ixs")
        prior_file = None
        if prior is not None:
            prior_file = os.path.join(d, "prior.bin")
            prior.astype(np.int32).tofile(prior_file)
        res = ref.orient_v_structures_merged(stem, alpha, N, orientation_prior_file=prior_file)
        ostem = os.path.join(d, "max_sep_min_pc")
        res.to_file(ostem)
        out["cases"][name] = {
            "alpha": alpha, "num_samples": N,
            "input": {"mdim": text(stem + ".mdim"), "sam": text(stem + "_sam.mtx"), "scm": text(stem + "_scm.mtx"),
                      "ixs": [int(v) for v in ixs], "prior": None if prior is None else [int(v) for v in prior.ravel()]},
            "output": {"mdim": text(ostem + ".mdim"), "sam": text(ostem + "_sam.mtx"), "scm": text(ostem + "_scm.mtx"),
                       "spm": text(ostem + "_spm.mtx"), "ssm": text(ostem + ".ssm"),
                       "ut": [int(v) for v in np.fromfile(ostem + ".ut", dtype=np.int32)],
                       "atr": [int(v) for v in np.fromfile(ostem + ".atr", dtype=np.int32)]},
            "pairs_with_minimum": len(res.min_sepsets),
        }
        print(name, "pairs", len(res.max_sepsets), "with minimum", len(res.min_sepsets), "triples",
              len(out["cases"][name]["output"]["ut"]) // 3, "ambiguous", len(out["cases"][name]["output"]["atr"]) // 3)

with open(os.path.join(HERE, "sepselect_kat.json"), "w") as f:
    json.dump(out, f, indent=0)
