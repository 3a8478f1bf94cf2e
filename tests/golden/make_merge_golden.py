#!/usr/bin/env python3
"""Consumer-acceptance goldens (north_star: "... so that merge-block-outputs, sepselect and srfci consume them
unchanged"): the REFERENCE's own `merge_block_outputs` / `reformat_cuskss_merged_output`
(/root/reference/cusk_postprocessing/merge_blocks.py:361-425, called from ci-gwas.py:452-464) run in the build
container on per-block result files, and the argv lists the reference's ci-gwas.py builds for `mps`
(ci-gwas.py:386-451), committed as DATA under tests/golden/merge/.  The GPU box has no reference checkout: there the
tests check that `mps` writes the committed per-block files byte for byte (so the reference's merge, a deterministic
function of those files, yields the committed merged files) and that this repo's own mirror
(ci-gwas_amd/merge.py) reproduces the reference-written files.

Chain:  inputs/ (summary statistics of a 180-marker x 3-trait toy chromosome in 4 blocks)
          -> blocks/          per-block `cuskss` outputs, written by the oracle pipeline (block 2 left out: a block
                              skipped by cusk has no files, merge_blocks.py:371-386)
          -> merged/          reference merge_block_outputs(...).write_mm(...)
          -> cuskss_merged_raw/   `cuskss --marker-indices merged_blocks.ixs` outputs (oracle pipeline)
          -> cuskss_merged/   reference reformat_cuskss_merged_output(...).write_mm(...)
        argv.json             argv lists captured from the reference's ci-gwas.py handlers

Usage (build container only):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_merge_golden.py [/root/reference]
"""
import importlib.util
import json
import os
import shutil
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import cigwas_amd.synth as synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "merge")
ALPHA, L1, L2, DEPTH, NSAMP = 1e-3, 3, 3, 1, 2000
BLOCKS = [(0, 59), (60, 99), (100, 139), (140, 179)]
MISSING = 2
M, P = 180, 3


def fresh(d):
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    return d


def write_tri(path, A):
    A[np.tril_indices(A.shape[0])].astype(np.float32).tofile(path)


def main():
    fresh(OUT)
    inp = fresh(os.path.join(OUT, "inputs"))
    # toy chromosome with marker -> trait effects strong enough to survive at N = 2000 (two causal markers per
    # block and trait) and a chain of trait -> trait effects
    rng = synth.rng_for(77)
    G = synth.make_genotypes(M, NSAMP, rng, window=30, rho=0.8, miss=0.0)
    gs = (G - G.mean(1, keepdims=True)) / G.std(1, keepdims=True)
    Y = np.zeros((P, NSAMP))
    for k in range(P):
        for a, b in BLOCKS:
            idx = a + rng.choice(b - a + 1, size=2, replace=False)
            Y[k] += (rng.uniform(0.2, 0.35, 2) * rng.choice([-1.0, 1.0], 2)) @ gs[idx]
        if k:
            Y[k] += 0.3 * Y[k - 1]
        Y[k] += rng.standard_normal(NSAMP)
    Cm = synth.corr_from_samples(G, Y.astype(np.float32))
    names = [f"T{k}" for k in range(P)]
    with open(os.path.join(inp, "mxp.txt"), "w") as f:
        f.write("chr snp ref " + " ".join(names) + "\n")
        for i in range(M):
            f.write(f"1 rs{i} A " + " ".join(repr(float(Cm[i, M + k])) for k in range(P)) + "\n")
    with open(os.path.join(inp, "pxp.txt"), "w") as f:
        f.write(" ".join(names) + "\n")
        for a in range(P):
            f.write(names[a] + " " + " ".join(repr(float(Cm[M + a, M + b])) for b in range(P)) + "\n")
    with open(os.path.join(inp, "blocks.txt"), "w") as f:
        for a, b in BLOCKS:
            f.write(f"1\t{a}\t{b}\n")
    for bi, (a, b) in enumerate(BLOCKS):
        write_tri(os.path.join(inp, f"mxm_{bi}.bin"), Cm[a:b + 1, a:b + 1])

    # ---- per-block cuskss outputs through the oracle pipeline (from the FILES, as mps reads them) ----
    blk = fresh(os.path.join(OUT, "blocks"))
    _, pxp, _ = O.load_pxp(os.path.join(inp, "pxp.txt"), sample_size=float(NSAMP))
    for bi, (a, b) in enumerate(BLOCKS):
        if bi == MISSING:
            continue
        mxm = O.load_mxm(os.path.join(inp, f"mxm_{bi}.bin"))
        mxp, _ = O.load_mxp(os.path.join(inp, "mxp.txt"), range(a, b + 1))
        sq, es = O.make_square_cuskss_inputs(mxm, mxp, pxp, float(NSAMP))
        red = O.cuskss_from_square(sq, es, P, ALPHA, L1, L2, DEPTH)
        O.write_reduced(red, os.path.join(blk, f"1_{a}_{b}"), with_sep=False)

    # ---- the reference's merge on those files ----
    from cusk_postprocessing.merge_blocks import merge_block_outputs, reformat_cuskss_merged_output

    mer = fresh(os.path.join(OUT, "merged"))
    merge_block_outputs(os.path.join(inp, "blocks.txt"), blk + "/").write_mm(os.path.join(mer, "merged_blocks"))
    ixs = np.fromfile(os.path.join(mer, "merged_blocks.ixs"), np.int32)
    assert ixs.size > 3, "toy chromosome retains too few markers for a useful golden"

    # ---- cuskss-merged on the union of the selected markers (oracle pipeline), then the reference's reformat ----
    write_tri(os.path.join(inp, "mxm_merged.bin"), Cm[np.ix_(ixs, ixs)])
    raw = fresh(os.path.join(OUT, "cuskss_merged_raw"))
    mxm = O.load_mxm(os.path.join(inp, "mxm_merged.bin"))
    mxp, _ = O.load_mxp(os.path.join(inp, "mxp.txt"), ixs)
    sq, es = O.make_square_cuskss_inputs(mxm, mxp, pxp, float(NSAMP))
    red = O.cuskss_from_square(sq, es, P, ALPHA, L1, L2, DEPTH)
    O.write_reduced(red, os.path.join(raw, "cuskss_merged"), with_sep=False)
    fin = fresh(os.path.join(OUT, "cuskss_merged"))
    for f in os.listdir(raw):
        shutil.copy(os.path.join(raw, f), fin)
    shutil.copy(os.path.join(mer, "merged_blocks.ixs"), fin)
    reformat_cuskss_merged_output(cusk_dir=fin).write_mm(basepath=f"{fin}/cuskss_merged")
    for f in ("cuskss_merged.adj", "cuskss_merged.corr", "merged_blocks.ixs"):
        os.remove(os.path.join(fin, f))  # inputs of the reformat step; its outputs stay

    # ---- argv lists of the reference's CLI handlers (subprocess.run captured, nothing executed) ----
    spec = importlib.util.spec_from_file_location("ref_ci_gwas", os.path.join(REF, "ci-gwas.py"))
    cg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cg)
    captured = []

    class _Sub:
        @staticmethod
        def run(argv, check=True):
            captured.append(["<MPS>" if a == cg.MPS_PATH else a for a in argv])

    cg.subprocess = _Sub
    cg.reformat_cuskss_merged_output = lambda cusk_dir: type("R", (), {"write_mm": staticmethod(lambda basepath: None)})()
    cases = {
        "prep-bed": ["prep-bed", "data/geno"],
        "block": ["block", "data/geno", "9000", "12", "1500"],
        "cusk": ["cusk", "7", "data/geno.blocks", "data/geno", "data/y.phen", "0.0001", "3", "14", "1", "out/"],
        "cuskss": ["cuskss", "--mxm", "a.mxm", "--mxp", "a.mxp", "--pxp", "a.pxp", "--block-index", "4", "--blockfile",
                   "a.blocks", "--alpha", "0.001", "--num-samples", "12345", "--outdir", "o/"],
        "cuskss-het": ["cuskss", "--mxm", "a.mxm", "--mxp", "a.mxp", "--pxp", "a.pxp", "--mxp-se", "a.mxp_se", "--pxp-se",
                       "a.pxp_se", "--time-index", "t.txt", "--block-index", "0", "--blockfile", "a.blocks", "--alpha",
                       "0.0001", "--max-level-one", "5", "--max-level-two", "6", "--max-depth", "2", "--num-samples", "458747",
                       "--outdir", "o/"],
        "cuskss-merged": ["cuskss", "--mxm", "m.mxm", "--mxp", "a.mxp", "--pxp", "a.pxp", "--marker-indices",
                          "o/merged_blocks.ixs", "--alpha", "0.0001", "--num-samples", "1000", "--outdir", "o/"],
        "cuskss-trait-only": ["cuskss", "--pxp", "a.pxp", "--blockfile", "a.blocks", "--alpha", "0.05", "--num-samples", "500"],
    }
    argv = {}
    for name, a in cases.items():
        captured.clear()
        sys.argv = ["ci-gwas"] + a
        cg.main()
        argv[name] = {"cli": a, "mps_argv": captured[-1]}
    with open(os.path.join(OUT, "argv.json"), "w") as f:
        json.dump(argv, f, indent=1)
    with open(os.path.join(OUT, "params.json"), "w") as f:
        json.dump({"alpha": ALPHA, "max_level_one": L1, "max_level_two": L2, "depth": DEPTH, "num_samples": NSAMP,
                   "blocks": BLOCKS, "missing_block": MISSING, "num_markers": M, "num_phen": P}, f, indent=1)
    total = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(OUT) for f in fs)
    print(f"wrote {OUT}: {total} bytes; merged markers {ixs.tolist()}")


if __name__ == "__main__":
    main()
