"""GPU: the whole workflow in order, one hop's files feeding the next (/root/reference/README.md:57-66, handlers
/root/reference/ci-gwas.py:386-470), at the scale of one GPU's share of a chromosome job:

    prep-bed -> block -> cusk on every block (the multi-GPU block driver, 2 ranks over gloo on the box's one GPU)
             -> merge-block-outputs -> cuskss-merged -> sepselect / orient-v-structs

Every artefact is compared with the oracle pipeline fed the PREVIOUS artefact of the product (so a deviation cannot
hide behind an earlier one): `.dim/.means/.stds` against the reference's own prep.cpp compiled in place (oracle/_ref,
when present), the `.blocks` file against the oracle's blocking, per-block result files against the oracle's
correlation build + two-stage skeleton on the packed genotypes, the merged files against oracle/merge_oracle.py (pinned
by files the reference wrote), `cuskss_merged.*` against the oracle's hetcor pipeline on the summary files, the
sepselect outputs against oracle/sepselect_oracle.py (pinned by files the reference wrote).
"""
import ctypes as C
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_host.so")
M1, M2, N, P = 1300, 700, 1500, 6
ALPHA, L1, L2, DEPTH = 1e-4, 3, 14, 1
MAX_BLOCK, WIDTH = 400, 200


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _same(a, b):
    assert open(a, "rb").read() == open(b, "rb").read(), (a, b)


@pytest.fixture(scope="module")
def chain(tmp_path_factory, synth):
    """PLINK file set of two chromosomes (no .dim/.means/.stds yet: `prep-bed` writes them) and a .phen whose traits
    depend on 8 markers each plus a trait DAG"""
    d = tmp_path_factory.mktemp("chain")
    m = M1 + M2
    rng = synth.rng_for(4242)
    G = synth.make_genotypes(m, N, rng, miss=0.002)
    g = G.astype(np.float64)
    g[G < 0] = np.nan
    gs = np.nan_to_num((g - np.nanmean(g, 1, keepdims=True)) / np.nanstd(g, 1, keepdims=True))
    Y = np.zeros((P, N))
    for k in range(P):
        idx = rng.choice(m, size=8, replace=False)
        y = (rng.uniform(0.12, 0.25, 8) * rng.choice([-1.0, 1.0], 8)) @ gs[idx]
        for k2 in range(k):
            if rng.random() < 0.5:
                y = y + rng.uniform(0.15, 0.3) * rng.choice([-1.0, 1.0]) * Y[k2]
        y = y + rng.standard_normal(N)
        Y[k] = (y - y.mean()) / y.std()
    stem = str(d / "geno")
    chr_ids = ["1"] * M1 + ["2"] * M2
    bed = synth.pack_bed(G)
    synth.write_bfiles(stem, bed, N, np.zeros(m), np.zeros(m), chr_ids)
    for sfx in (".dim", ".means", ".stds"):
        os.remove(stem + sfx)
    phen = np.ascontiguousarray(Y.astype(np.float32)).reshape(-1)
    synth.write_phen(str(d / "y.phen"), phen, N, P)
    return dict(dir=d, stem=stem, phen=str(d / "y.phen"), bed=bed, chr_ids=chr_ids, m=m)


@pytest.mark.timeout(1500)
def test_workflow_chain(chain, oracle, tmp_path):
    from cigwas_amd import cli
    from oracle import merge_oracle as MO
    from oracle import sepselect_oracle as SO

    stem, m, bed = chain["stem"], chain["m"], chain["bed"]

    # ---- 1. prep-bed (ci-gwas.py:386-387 -> `mps prep`) ----------------------------------------------------------------
    cli.main(["prep-bed", stem])
    assert open(stem + ".dim").read().split() == [str(N), str(m)]
    if os.path.exists(REF_SO):
        theirs = stem + "_ref"
        for sfx in (".bed", ".bim", ".fam"):
            os.link(stem + sfx, theirs + sfx)
        L = C.CDLL(REF_SO)
        L.ref_prep.restype = None
        L.ref_prep(theirs.encode())
        for sfx in (".dim", ".means", ".stds", ".modes"):
            _same(stem + sfx, theirs + sfx)
    means = np.loadtxt(stem + ".means", dtype=np.float32)
    stds = np.loadtxt(stem + ".stds", dtype=np.float32)
    assert means.shape == (m,) and np.all(stds > 0)

    # ---- 2. block (ci-gwas.py:390-401 -> `mps block`) -------------------------------------------------------------------
    cli.main(["block", stem, str(MAX_BLOCK), "1", str(WIDTH)])
    blocks = f"{stem}_m{MAX_BLOCK}.blocks"
    got = open(blocks).read().splitlines()
    assert got == oracle.make_blocks(bed, chain["chr_ids"], N, MAX_BLOCK, WIDTH)
    bounds, start = [], {"1": 0, "2": M1}
    for ln in got:
        cid, a, b = ln.split("\t")
        bounds.append((cid, int(a), int(b), start[cid] + int(a)))
    assert len(bounds) >= 8 and {c for c, *_ in bounds} == {"1", "2"}

    # ---- 3. cusk on every block: the block driver, two ranks (gloo) sharing the GPU ---------------------------------------
    out = tmp_path / "cusk"
    out.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "ci-gwas_amd", "run_blocks.py"), chain["phen"], stem, blocks,
           str(ALPHA), str(L1), str(L2), str(DEPTH), str(out), "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    phen = oracle.load_phen(chain["phen"])[2]
    Th = oracle.threshold_array(N, ALPHA)
    written, skipped = 0, 0
    for cid, a, b, g0 in bounds:
        mb = b - a + 1
        sel = slice(g0, g0 + mb)
        o_mxm, o_mxp, o_pxp = oracle.corr_pearson_npn(bed[sel], phen, mb, N, P, means[sel], stds[sel])
        base = str(out / f"{cid}_{a}_{b}")
        if oracle.prefilter_count(o_mxp, Th[0]) == 0:  # cli.cpp:561-576: no files for such a block
            assert not os.path.exists(base + ".mdim")
            skipped += 1
            continue
        ref = oracle.cusk_from_corr(oracle.square_from_cusk_corrs(o_mxm, o_mxp, o_pxp, mb, P), P, Th, L1, L2, DEPTH)
        assert open(base + ".mdim").read() == f"{ref.num_var}\t{P}\t14\n"
        assert list(np.fromfile(base + ".ixs", np.int32)) == list(ref.new_to_old)
        assert np.array_equal(np.fromfile(base + ".adj", np.int32), np.asarray(ref.G, np.int32).reshape(-1))
        assert np.array_equal(np.fromfile(base + ".sep", np.int32), np.asarray(ref.S, np.int32).reshape(-1))
        # SNP x trait / trait x trait correlations: 1e-5 as the reference's own corr tests; SNP x SNP bit-exact
        assert np.allclose(np.fromfile(base + ".corr", np.float32), np.asarray(ref.C, np.float32).reshape(-1), atol=1e-5, rtol=0)
        written += 1
    assert written >= 6 and skipped >= 1
    assert len(os.listdir(out)) == 5 * written

    # ---- 4. merge-block-outputs (ci-gwas.py:459-464) on the files of step 3 ----------------------------------------------
    cli.main(["merge-block-outputs", str(out), blocks])
    exp = tmp_path / "merge_oracle"
    exp.mkdir()
    MO.write_mm(MO.merge(blocks, str(out) + "/"), str(exp / "merged_blocks"))
    for f in ("merged_blocks_sam.mtx", "merged_blocks_scm.mtx", "merged_blocks.mdim", "merged_blocks.ixs"):
        _same(str(out / f), str(exp / f))
    ixs = np.fromfile(str(out / "merged_blocks.ixs"), np.int32)
    k = len(ixs)
    assert k >= 10 and np.all(np.diff(ixs) > 0) and ixs[-1] < m and (ixs >= M1).any()

    # ---- 5. cuskss-merged (ci-gwas.py:423-456) on the union of the selected markers ---------------------------------------
    # the summary statistics a user would bring (README.md:65): LD of the selected markers, marker-trait correlations
    # of ALL markers (rows picked by --marker-indices), trait-trait correlations
    mxm_sel, _, pxp = oracle.corr_pearson_npn(bed[ixs], phen, k, N, P, means[ixs], stds[ixs])
    mxp_all = oracle.marker_phen_corr_pearson(bed, phen, m, N, P, means, stds).reshape(m, P)
    sqm = np.ones((k, k), np.float32)
    iu = np.triu_indices(k, 1)
    sqm[iu] = mxm_sel
    sqm.T[iu] = mxm_sel
    ss = tmp_path / "sumstats"
    ss.mkdir()
    sqm[np.tril_indices(k)].astype(np.float32).tofile(ss / "mxm.bin")
    names = [f"T{t}" for t in range(P)]
    with open(ss / "mxp.txt", "w") as f:
        f.write("chr snp ref " + " ".join(names) + "\n")
        for i in range(m):
            f.write(f"{chain['chr_ids'][i]} rs{i} A " + " ".join(repr(float(v)) for v in mxp_all[i]) + "\n")
    pp = np.ones((P, P), np.float32)
    ip = np.triu_indices(P, 1)
    pp[ip] = pxp
    pp.T[ip] = pxp
    with open(ss / "pxp.txt", "w") as f:
        f.write(" ".join(names) + "\n")
        for a in range(P):
            f.write(names[a] + " " + " ".join(repr(float(v)) for v in pp[a]) + "\n")
    cm = tmp_path / "cuskss_merged"
    cm.mkdir()
    shutil.copy(str(out / "merged_blocks.ixs"), cm)
    cli.main(["cuskss-merged", "--mxm", str(ss / "mxm.bin"), "--mxp", str(ss / "mxp.txt"), "--pxp", str(ss / "pxp.txt"),
              "--marker-indices", str(cm / "merged_blocks.ixs"), "--alpha", str(ALPHA), "--max-level-one", "3", "--max-level-two", "3",
              "--max-depth", "1", "--num-samples", str(N), "--outdir", str(cm)])
    mxm_l = oracle.load_mxm(str(ss / "mxm.bin"))
    _, pxp_l, _ = oracle.load_pxp(str(ss / "pxp.txt"), sample_size=float(N))
    mxp_l, _ = oracle.load_mxp(str(ss / "mxp.txt"), [int(v) for v in ixs])
    sq, es = oracle.make_square_cuskss_inputs(mxm_l, mxp_l, pxp_l, float(N))
    red = oracle.cuskss_from_square(sq, es, P, ALPHA, 3, 3, 1)
    expc = tmp_path / "cuskss_oracle"
    expc.mkdir()
    oracle.write_reduced(red, str(expc / "cuskss_merged"), with_sep=False)
    # (the CLI's post-step has already replaced the raw .ixs -- positions in the selected-marker list -- by global marker
    # indices, merge_blocks.py:322-324; the raw one is what the rewritten one is computed from, checked below)
    for ext in (".mdim", ".adj", ".corr"):
        _same(str(cm / "cuskss_merged") + ext, str(expc / "cuskss_merged") + ext)
    # ... and the post-step that rewrites it in the merged sparse format (ci-gwas.py:452-456), from the oracle's dense
    # files (identical to the product's, just compared)
    raw = tmp_path / "cuskss_raw"
    raw.mkdir()
    for ext in (".mdim", ".adj", ".corr"):
        shutil.copy(str(expc / "cuskss_merged") + ext, raw)
    shutil.copy(str(expc / "cuskss_merged.ixs"), raw)
    shutil.copy(str(out / "merged_blocks.ixs"), raw)
    MO.write_mm(MO.reformat_cuskss_merged(str(raw)), str(raw / "cuskss_merged"))
    for f in ("cuskss_merged_sam.mtx", "cuskss_merged_scm.mtx", "cuskss_merged.mdim", "cuskss_merged.ixs"):
        _same(str(cm / f), str(raw / f))
    assert red.num_var > P + 5

    # ---- 6. sepselect / orient-v-structs (ci-gwas.py:467-476) on the merged skeleton of step 5 ----------------------------
    res = SO.run(str(cm / "cuskss_merged"), ALPHA, N)
    exps = tmp_path / "sepselect_oracle"
    exps.mkdir()
    SO.write(res, str(exps / "max_sep_min_pc"))
    cli.main(["sepselect", str(cm / "cuskss_merged"), str(ALPHA), str(N)])
    for sfx in (".mdim", ".ssm", ".ut", ".atr", "_sam.mtx", "_scm.mtx"):
        _same(str(cm / "max_sep_min_pc") + sfx, str(exps / "max_sep_min_pc") + sfx)
    assert not os.path.exists(str(cm / "max_sep_min_pc_spm.mtx"))
    cli.main(["orient-v-structs", str(cm / "cuskss_merged"), str(ALPHA), str(N)])
    for sfx in (".mdim", ".ssm", ".ut", ".atr", "_sam.mtx", "_scm.mtx", "_spm.mtx"):
        _same(str(cm / "max_sep_min_pc") + sfx, str(exps / "max_sep_min_pc") + sfx)
    assert res["rel"].shape[0] > 10 and len(res["max_sepsets"]) > 10
