"""Consumer acceptance (north_star: merge-block-outputs / cuskss-merged consume the result files unchanged).

tests/golden/merge/ holds files written by the REFERENCE's own merge_block_outputs / reformat_cuskss_merged_output
and argv lists captured from the reference's ci-gwas.py (tests/golden/make_merge_golden.py, run in the build
container).  CPU part: the oracle pipeline reproduces the committed per-block files from the committed inputs, this
package's mirror (ci-gwas_amd/merge.py) reproduces the reference-written merged files byte for byte, and the CLI
shim builds the reference's argv lists.  GPU part: `mps cuskss` writes the committed per-block files byte for byte,
so the reference's merge (a deterministic function of those files) yields the committed merged files.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "merge")
MPS = os.path.join(ROOT, "ci-gwas_amd", "csrc", "mps")
PAR = json.load(open(os.path.join(G, "params.json")))


def _same_dir(a, b, names=None):
    names = names or sorted(os.listdir(a))
    for f in names:
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f


def test_oracle_pipeline_reproduces_the_block_files(oracle, tmp_path):
    inp = os.path.join(G, "inputs")
    _, pxp, _ = oracle.load_pxp(os.path.join(inp, "pxp.txt"), sample_size=float(PAR["num_samples"]))
    for bi, (a, b) in enumerate(PAR["blocks"]):
        if bi == PAR["missing_block"]:
            continue
        mxm = oracle.load_mxm(os.path.join(inp, f"mxm_{bi}.bin"))
        mxp, _ = oracle.load_mxp(os.path.join(inp, "mxp.txt"), range(a, b + 1))
        sq, es = oracle.make_square_cuskss_inputs(mxm, mxp, pxp, float(PAR["num_samples"]))
        red = oracle.cuskss_from_square(sq, es, PAR["num_phen"], PAR["alpha"], PAR["max_level_one"], PAR["max_level_two"], PAR["depth"])
        oracle.write_reduced(red, str(tmp_path / f"1_{a}_{b}"), with_sep=False)
    _same_dir(os.path.join(G, "blocks"), str(tmp_path))


def test_merge_mirror_writes_the_files_the_reference_wrote(tmp_path, capsys):
    from cigwas_amd import merge

    res = merge.merge_block_outputs(os.path.join(G, "inputs", "blocks.txt"), os.path.join(G, "blocks") + "/")
    assert "Missing:" in capsys.readouterr().out  # block 2 has no files
    res.write_mm(str(tmp_path / "merged_blocks"))
    _same_dir(os.path.join(G, "merged"), str(tmp_path))
    assert res.num_phen == PAR["num_phen"] and res.num_var == PAR["num_phen"] + len(res.gmi)
    # the cuskss-merged post-step (ci-gwas.py:452-456)
    d = tmp_path / "cm"
    shutil.copytree(os.path.join(G, "cuskss_merged_raw"), d)
    shutil.copy(os.path.join(G, "merged", "merged_blocks.ixs"), d)
    merge.reformat_cuskss_merged_output(str(d)).write_mm(f"{d}/cuskss_merged")
    _same_dir(os.path.join(G, "cuskss_merged"), str(d), sorted(os.listdir(os.path.join(G, "cuskss_merged"))))
    # merged marker indices are global .bim rows inside the blocks that have files
    ixs = np.fromfile(os.path.join(G, "merged", "merged_blocks.ixs"), np.int32)
    a, b = PAR["blocks"][PAR["missing_block"]]
    assert np.all(np.diff(ixs) > 0) and not np.any((ixs >= a) & (ixs <= b))


def test_merge_oracle_reproduces_the_reference_files(tmp_path):
    """the checker of the workflow chain test (oracle/merge_oracle.py) is pinned by the files the reference wrote"""
    from oracle import merge_oracle as MO

    MO.write_mm(MO.merge(os.path.join(G, "inputs", "blocks.txt"), os.path.join(G, "blocks") + "/"), str(tmp_path / "merged_blocks"))
    _same_dir(os.path.join(G, "merged"), str(tmp_path))
    d = tmp_path / "cm"
    shutil.copytree(os.path.join(G, "cuskss_merged_raw"), d)
    shutil.copy(os.path.join(G, "merged", "merged_blocks.ixs"), d)
    MO.write_mm(MO.reformat_cuskss_merged(str(d)), f"{d}/cuskss_merged")
    _same_dir(os.path.join(G, "cuskss_merged"), str(d), sorted(os.listdir(os.path.join(G, "cuskss_merged"))))


def _pack_block_files(blockdir, stems_in_order, with_sep=False):
    """per-block result files -> the byte string the multi-GPU job gathers (shard.BlockResult.pack layout)"""
    from cigwas_amd import shard

    parts = []
    for bi, stem in enumerate(stems_in_order):
        base = os.path.join(blockdir, stem)
        if not os.path.exists(base + ".mdim"):
            continue
        nv, nph, ml = (int(v) for v in open(base + ".mdim").read().split())
        r = shard.BlockResult(bi, stem, nph, ml, np.fromfile(base + ".ixs", np.int32), np.fromfile(base + ".adj", np.int32).reshape(nv, nv),
                              np.fromfile(base + ".corr", np.float32).reshape(nv, nv), None)
        parts.append(r.pack())
    return np.concatenate(parts[::-1])  # any order: the merge goes by the block file


def test_library_merge_of_packed_results_writes_the_files_the_reference_wrote(tmp_path):
    """cusk_merge_packed (csrc/host/merge.h: the merged skeleton straight from the gathered results, host code of the library)
    against the files the REFERENCE's merge_block_outputs wrote from the same per-block files"""
    import ctypes as C

    from cigwas_amd import merge
    from cigwas_amd._lib import lib

    blockfile = os.path.join(G, "inputs", "blocks.txt")
    buf = _pack_block_files(os.path.join(G, "blocks"), merge.block_stems(blockfile))
    rc = lib().cusk_merge_packed(blockfile.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, str(tmp_path / "merged_blocks").encode())
    assert rc == 0, lib().cusk_blockset_last_error()
    _same_dir(os.path.join(G, "merged"), str(tmp_path))


def test_library_prints_floats_as_numpy_does():
    """the .mtx values are f"{numpy.float32}" = repr(float(v)): shortest digits of the double, positional for exponents in [-4, 16)"""
    import ctypes as C

    from cigwas_amd import merge, shard
    from cigwas_amd._lib import lib
    import tempfile

    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.uniform(-1, 1, 300), 10.0 ** rng.uniform(-9, -3, 200) * rng.choice([-1, 1], 200),
                           [1.0, -1.0, 0.1, 1e-4, 9.9999e-5, 0.00012345, 1e-5, 0.5, 1e-45, 3.4028235e38, 123456.0, 1e15, 1e16, np.nan]])
    vals = vals.astype(np.float32)
    k = len(vals)
    corr = np.zeros((k + 1, k + 1), np.float32)
    corr[0, 1:] = vals  # one row of non-zeros: entries (k + 1? no: marker 0 -> merged index p + 1 = 2; trait -> 1)
    adj = np.zeros((k + 1, k + 1), np.int32)
    adj[0, 1] = 1
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "b.blocks"), "w").write(f"1\t0\t{k - 1}\n")
        r = shard.BlockResult(0, f"1_0_{k - 1}", 1, 14, np.arange(k + 1, dtype=np.int32), adj, corr, None)
        buf = r.pack()
        rc = lib().cusk_merge_packed(os.path.join(d, "b.blocks").encode(), buf.ctypes.data_as(C.c_void_p), buf.size, os.path.join(d, "m").encode())
        assert rc == 0
        got = [ln.split("\t")[2] for ln in open(os.path.join(d, "m_scm.mtx")).read().splitlines()[2:]]
    want = [f"{v}" for v in vals if v != 0]
    assert got == want


def test_cli_shim_builds_the_argv_of_the_reference_cli():
    """argv lists captured from the reference's ci-gwas.py handlers (subprocess.run intercepted)"""
    from cigwas_amd import cli

    cases = json.load(open(os.path.join(G, "argv.json")))
    builders = {"prep-bed": lambda a: [cli.MPS_PATH, "prep", a.bfiles], "block": cli.block_argv, "cusk": cli.cusk_argv}
    for name, c in cases.items():
        args = cli.build_parser().parse_args(c["cli"])
        argv = builders.get(name, cli.cuskss_argv)(args)
        assert argv[0] == cli.MPS_PATH and argv[1:] == c["mps_argv"][1:], name
    # the aliases select the same mode
    het = cases["cuskss-het"]
    a2 = cli.build_parser().parse_args(["cuskss-het"] + het["cli"][1:])
    assert cli.cuskss_argv(a2)[1:] == het["mps_argv"][1:]
    mer = cases["cuskss-merged"]
    a3 = cli.build_parser().parse_args(["cuskss-merged"] + mer["cli"][1:])
    assert cli.cuskss_argv(a3)[1:] == mer["mps_argv"][1:]


@pytest.mark.gpu
def test_mps_cuskss_writes_the_committed_block_files(tmp_path):
    inp = os.path.join(G, "inputs")
    out = tmp_path / "blocks"
    out.mkdir()
    for bi in range(len(PAR["blocks"])):
        if bi == PAR["missing_block"]:
            continue
        r = subprocess.run([MPS, "cuskss", os.path.join(inp, f"mxm_{bi}.bin"), os.path.join(inp, "mxp.txt"), "NULL",
                            os.path.join(inp, "pxp.txt"), "NULL", "NULL", str(bi), os.path.join(inp, "blocks.txt"), "NULL",
                            str(PAR["alpha"]), str(PAR["max_level_one"]), str(PAR["max_level_two"]), str(PAR["depth"]),
                            str(PAR["num_samples"]), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    _same_dir(os.path.join(G, "blocks"), str(out))
    assert sorted(os.listdir(out)) == sorted(os.listdir(os.path.join(G, "blocks")))
    # the merged run on the union of the selected markers, then this package's CLI post-step
    d = tmp_path / "cm"
    d.mkdir()
    shutil.copy(os.path.join(G, "merged", "merged_blocks.ixs"), d)
    from cigwas_amd import cli

    cli.main(["cuskss-merged", "--mxm", os.path.join(inp, "mxm_merged.bin"), "--mxp", os.path.join(inp, "mxp.txt"), "--pxp",
              os.path.join(inp, "pxp.txt"), "--marker-indices", str(d / "merged_blocks.ixs"), "--alpha", str(PAR["alpha"]),
              "--max-level-one", str(PAR["max_level_one"]), "--max-level-two", str(PAR["max_level_two"]), "--max-depth",
              str(PAR["depth"]), "--num-samples", str(PAR["num_samples"]), "--outdir", str(d)])
    _same_dir(os.path.join(G, "cuskss_merged"), str(d), sorted(os.listdir(os.path.join(G, "cuskss_merged"))))
    for f in ("cuskss_merged.adj", "cuskss_merged.corr"):
        assert open(d / f, "rb").read() == open(os.path.join(G, "cuskss_merged_raw", f), "rb").read()
