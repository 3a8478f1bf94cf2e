"""GPU: batched execution of small LD blocks (the reference runs one block per process,
cli.cpp:507-512).  Engine level: several matrices on the diagonal of one allocation swept in ONE run
(cusk_run_skeleton_batch) must give, block by block, exactly what cusk_run_skeleton gives for the block alone and what
the oracle gives -- adjacency, separating sets, canonical test counts summed over the blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ML = 14


@pytest.fixture(scope="module")
def cg():
    import cigwas_amd

    return cigwas_amd


def _layout(sizes):
    lo, hi, base = [], [], 0
    for k in sizes:
        lo.append(base)
        hi.append(base + k)
        base += (k + 63) // 64 * 64
    return np.array(lo, np.int32), np.array(hi, np.int32), base


def _batch_matrix(mats, lo, n, fill):
    """blocks on the diagonal; everything else `fill` (NaN / garbage: must never be read)"""
    C = np.full((n, n), fill, np.float32)
    for Cm, b in zip(mats, lo):
        k = Cm.shape[0]
        C[b:b + k, b:b + k] = Cm
    return C


@pytest.mark.parametrize("maxlevel,fill", [(3, np.nan), (14, 7.5)])
def test_batched_run_equals_per_block_runs_and_oracle(cg, oracle, synth, maxlevel, fill):
    # unequal blocks: LD blocks, a dense random SEM, a hub graph that keeps its degree to level 14, a tiny one, one whose
    # size is an exact multiple of 64 (no padding rows behind it)
    mats = [synth.synth_corr_block(150, 5, N=2048, block_index=11), synth.random_corr(40, seed=3, k=240),
            synth.hub_corr(16, 3, seed=4), synth.synth_corr_block(123, 5, N=2048, block_index=12),
            synth.random_corr(6, seed=9, k=60), synth.synth_corr_block(300, 6, N=2048, block_index=13)]
    assert mats[3].shape[0] == 128
    Th = cg.threshold_array(2048, 1e-3)
    lo, hi, n = _layout([m.shape[0] for m in mats])
    e = cg.Engine(0)
    Cd = cg.DeviceArray(_batch_matrix(mats, lo, n, fill))
    st = e.run_skeleton_batch(Cd.ptr, n, lo, hi, Th, maxlevel)
    Gs = e.adjacency_blocks()
    x, y, lv, z, S = e.sepsets()
    canon = np.zeros(ML + 1, np.int64)
    tests = np.zeros(ML + 1, np.int64)
    e1 = cg.Engine(0)
    level_max = 0
    for b, Cm in enumerate(mats):
        k = Cm.shape[0]
        ref = oracle.skeleton(Cm, Th, maxlevel)
        assert np.array_equal(Gs[b], ref.G), b
        sel = (x >= lo[b]) & (x < hi[b])
        assert np.all((y[sel] >= lo[b]) & (y[sel] < hi[b]))
        dense = np.full((k, k, ML), -1, np.int32)
        Sb = S[sel].copy()
        Sb[Sb >= 0] -= lo[b]
        dense[x[sel] - lo[b], y[sel] - lo[b]] = Sb
        assert np.array_equal(dense, ref.sepset), b
        canon += ref.tests
        level_max = max(level_max, ref.level)
        # and the single-block engine run on the block alone
        C1 = cg.DeviceArray(Cm)
        s1 = e1.run_skeleton(C1.ptr, k, Th, maxlevel)
        assert np.array_equal(e1.adjacency(), Gs[b])
        tests[: len(s1.tests)] += np.array(s1.tests)
        C1.free()
    e1.close()
    assert st.level == level_max
    assert list(st.canonical_tests[: level_max + 1]) == [int(v) for v in canon[: level_max + 1]]
    assert st.tests[0] == canon[0]  # level 0: the pairs inside the blocks only
    if maxlevel == 14:
        assert st.levels_run == 15
    Cd.free()
    e.close()


def test_batch_of_one_and_bad_layouts(cg, oracle, synth):
    Cm = synth.synth_corr_block(200, 4, N=2048, block_index=5)
    k = Cm.shape[0]
    Th = cg.threshold_array(2048, 1e-3)
    e = cg.Engine(0)
    n = (k + 63) // 64 * 64
    Cd = cg.DeviceArray(_batch_matrix([Cm], [0], n, np.nan))
    e.run_skeleton_batch(Cd.ptr, n, [0], [k], Th, 3)
    assert np.array_equal(e.adjacency_blocks()[0], oracle.skeleton(Cm, Th, 3).G)
    with pytest.raises(RuntimeError, match="multiples of 64"):
        e.run_skeleton_batch(Cd.ptr, n, [3], [k], Th, 3)
    with pytest.raises(RuntimeError, match="multiples of 64"):
        e.run_skeleton_batch(Cd.ptr, n, [0, 64], [100, 128], Th, 3)
    Cd.free()
    e.close()


def test_gather_rows_places_submatrices(cg):
    rng = np.random.default_rng(3)
    n = 90
    M = rng.standard_normal((n, n)).astype(np.float32)
    Md = cg.DeviceArray(M)
    lists = [np.array([3, 7, 8, 40]), np.array([50, 51, 89]), np.array([0])]
    idx = np.concatenate(lists)
    row_src, row_k, row_first, row_out = [], [], [], []
    first, out = 0, 0
    for L in lists:
        for r in range(len(L)):
            row_src.append(L[r])
            row_k.append(len(L))
            row_first.append(first)
            row_out.append(out + r * len(L))
        first += len(L)
        out += len(L) ** 2
    e = cg.Engine(0)
    got = e.gather_rows(Md.ptr, n, idx, row_src, row_k, row_first, row_out, out_count=out)
    want = np.concatenate([M[np.ix_(L, L)].reshape(-1) for L in lists])
    assert np.array_equal(got, want)
    Md.free()
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# block driver level: cusk_blockset_run_batch against per-block `mps cusk` (files byte-identical) and the oracle pipeline
# ---------------------------------------------------------------------------------------------------------------------
import os
import socket
import subprocess
import sys

from test_gpu_run_blocks import ALPHA, DEPTH, L1, L2, ROOT, _same_files, chromosome  # noqa: E402,F401  (the fixture)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("batch_vars,writer", [(100000, "local"), (1024, "rank0"), (600, "local")])
def test_batched_driver_writes_the_files_of_per_block_mps_cusk(chromosome, tmp_path, batch_vars, writer):
    """all 11 blocks in one batch, in several batches, and with batches of one block (every block above the budget)"""
    from cigwas_amd import run_blocks as rb

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), int(DEPTH))
    out = tmp_path / "out"
    out.mkdir()
    tm = {}
    done, stats, owned = rb.run_job(bs, str(out), device=0, writer=writer, batch_vars=batch_vars, options={"timing": 0}, timings=tm)
    files = _same_files(str(chromosome["ref"]), str(out))
    assert len(files) == 5 * len(done) and owned == list(range(bs.num_blocks))
    assert sum(s.blocks for s in stats) == bs.num_blocks and sum(s.skipped for s in stats) == bs.num_blocks - len(done) > 0
    assert all(s.tests[0] > 0 for s in stats if s.blocks > s.skipped) and "compute_s" in tm
    if batch_vars == 100000:
        assert len(stats) == 1 and stats[0].vars_stage1 >= 2900 and stats[0].canonical[0] > 0 and stats[0].canonical[1] > 0
    else:
        assert len(stats) > 3
    bs.close()


def test_batch_result_accessors_and_oracle(chromosome, oracle, tmp_path):
    """a batch's results through the pack / unpack path, two blocks against the oracle pipeline from the packed genotypes"""
    from cigwas_amd import run_blocks as rb
    from cigwas_amd.skeleton import Engine

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), int(DEPTH))
    e = Engine(0)
    sig = chromosome["signal"]
    br, st = bs.run_batch(e, [sig[-1], 1, sig[0]])  # any order, a block without signal in between
    res = {r.block_index: r for r in br.results()}
    assert set(res) == {sig[0], sig[-1]} and st.skipped == 1 and br.block_indices == [sig[-1], sig[0]]
    bed = np.fromfile(chromosome["stem"] + ".bed", np.uint8)[3:].reshape(-1, (chromosome["N"] + 3) // 4)
    means = np.loadtxt(chromosome["stem"] + ".means", dtype=np.float32)
    stds = np.loadtxt(chromosome["stem"] + ".stds", dtype=np.float32)
    phen = oracle.load_phen(chromosome["phen"])[2]
    N, p = chromosome["N"], chromosome["p"]
    Th = oracle.threshold_array(N, float(ALPHA))
    for b in (sig[0], sig[-1]):
        cid, f, l, g0 = chromosome["bounds"][b]
        mb = l - f + 1
        sel = slice(g0, g0 + mb)
        o_mxm, o_mxp, o_pxp = oracle.corr_pearson_npn(bed[sel], phen, mb, N, p, means[sel], stds[sel])
        ref = oracle.cusk_from_corr(oracle.square_from_cusk_corrs(o_mxm, o_mxp, o_pxp, mb, p), p, Th, int(L1), int(L2), int(DEPTH))
        r = res[b]
        assert r.stem == f"{cid}_{f}_{l}" and r.max_level == 14 and r.num_phen == p
        assert list(r.new_to_old) == list(ref.new_to_old)
        assert np.array_equal(r.adj.reshape(-1), np.asarray(ref.G, np.int32).reshape(-1))
        assert np.array_equal(r.sep.reshape(-1), np.asarray(ref.S, np.int32).reshape(-1))
        assert np.allclose(r.corr.reshape(-1), np.asarray(ref.C, np.float32).reshape(-1), atol=1e-5, rtol=0)
    # packed bytes -> files on "rank 0"
    out = tmp_path / "o"
    out.mkdir()
    assert rb.write_packed(br.pack(), str(out)) == 2
    for f in os.listdir(out):
        assert open(out / f, "rb").read() == open(os.path.join(str(chromosome["ref"]), f), "rb").read()
    # the list form of the separating sets (what the rank-0 writer receives): a fraction of the bytes, the same files;
    # and the merge reads past either form
    out2 = tmp_path / "o2"
    out2.mkdir()
    sparse = br.pack(with_sep=2)
    assert sparse.size < br.pack().size // 4 and rb.write_packed(sparse, str(out2)) == 2
    assert sorted(os.listdir(out2)) == sorted(os.listdir(out))
    for f in os.listdir(out2):
        assert open(out2 / f, "rb").read() == open(out / f, "rb").read(), f
    for k, form in enumerate((sparse, br.pack(), br.pack(with_sep=False))):
        rb.merge_packed(chromosome["blocks"], form, str(tmp_path / f"m{k}"))
    for suffix in ("_sam.mtx", "_scm.mtx", ".mdim", ".ixs"):
        assert open(str(tmp_path / "m0") + suffix, "rb").read() == open(str(tmp_path / "m2") + suffix, "rb").read() == open(str(tmp_path / "m1") + suffix, "rb").read()
    br.free()
    lib_release = rb.lib().cusk_blockset_release_engine
    lib_release(bs.h, e.h)
    e.close()
    bs.close()


def test_run_block_to_files_writes_what_mps_cusk_writes(chromosome, tmp_path):
    """BlockSet.run_block_to_files (pipeline + the library's own file writer, what bench.py's end-to-end block figure
    times) leaves, block by block, the five files per-block `mps cusk` runs left; a block without signal writes nothing"""
    from cigwas_amd import run_blocks as rb
    from cigwas_amd.skeleton import Engine

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), int(DEPTH))
    e = Engine(0)
    out = tmp_path / "o"
    out.mkdir()
    sig = chromosome["signal"]
    quiet = [i for i in range(len(chromosome["bounds"])) if i not in sig][0]
    for b in (sig[0], quiet, sig[-1]):
        written, st, secs = bs.run_block_to_files(e, b, str(out))
        assert written == (b in sig) and secs >= 0.0
    assert len(os.listdir(out)) == 10
    for f in os.listdir(out):
        assert open(out / f, "rb").read() == open(os.path.join(str(chromosome["ref"]), f), "rb").read(), f
    rb.lib().cusk_blockset_release_engine(bs.h, e.h)
    e.close()
    bs.close()


@pytest.mark.timeout(900)
def test_two_ranks_batched_over_gloo(chromosome, tmp_path):
    """the job under torchrun, two ranks sharing the GPU, batched execution, results gathered to rank 0"""
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "ci-gwas_amd", "run_blocks.py"), chromosome["phen"],
           chromosome["stem"], chromosome["blocks"], ALPHA, L1, L2, DEPTH, str(out), "--backend", "gloo", "--batch-vars", "2048",
           "--writer", "rank0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    _same_files(str(chromosome["ref"]), str(out))


def test_depth_two_batch_equals_per_block_and_oracle(chromosome, oracle):
    """max_depth 2 takes the general pruning (the whole bitmap of a block travels; depth 1 reads the trait rows only):
    batched and per-block execution agree with each other and, for one block, with the oracle pipeline"""
    from cigwas_amd import run_blocks as rb
    from cigwas_amd.skeleton import Engine

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), 2)
    e = Engine(0)
    sig = chromosome["signal"][:3]
    br, _st = bs.run_batch(e, sig)
    res = {r.block_index: r for r in br.results()}
    for b in sig:
        one, _ = bs.run_block(e, b)
        r = res[b]
        assert np.array_equal(r.new_to_old, one.new_to_old) and np.array_equal(r.adj, one.adj)
        assert np.array_equal(r.corr, one.corr) and np.array_equal(r.sep, one.sep)
    b = sig[0]
    cid, f, l, g0 = chromosome["bounds"][b]
    mb, N, p = l - f + 1, chromosome["N"], chromosome["p"]
    bed = np.fromfile(chromosome["stem"] + ".bed", np.uint8)[3:].reshape(-1, (N + 3) // 4)
    means = np.loadtxt(chromosome["stem"] + ".means", dtype=np.float32)
    stds = np.loadtxt(chromosome["stem"] + ".stds", dtype=np.float32)
    phen = oracle.load_phen(chromosome["phen"])[2]
    sel = slice(g0, g0 + mb)
    o_mxm, o_mxp, o_pxp = oracle.corr_pearson_npn(bed[sel], phen, mb, N, p, means[sel], stds[sel])
    ref = oracle.cusk_from_corr(oracle.square_from_cusk_corrs(o_mxm, o_mxp, o_pxp, mb, p), p, oracle.threshold_array(N, float(ALPHA)),
                                int(L1), int(L2), 2)
    assert list(res[b].new_to_old) == list(ref.new_to_old)
    assert np.array_equal(res[b].adj.reshape(-1), np.asarray(ref.G, np.int32).reshape(-1))
    assert np.array_equal(res[b].sep.reshape(-1), np.asarray(ref.S, np.int32).reshape(-1))
    br.free()
    rb.lib().cusk_blockset_release_engine(bs.h, e.h)
    e.close()
    bs.close()


def test_merge_writer_leaves_block_files_and_the_merged_skeleton(chromosome, tmp_path, capsys):
    """writer = "merge": every block's five files as per-block `mps cusk` writes them, plus the merged skeleton written by
    the library from the gathered results -- equal to what `merge-block-outputs` makes of those files (this package's
    mirror, pinned by reference-written goldens, and the oracle's restatement)"""
    from cigwas_amd import cli
    from cigwas_amd import run_blocks as rb
    from oracle import merge_oracle as MO

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), int(DEPTH))
    out = tmp_path / "out"
    out.mkdir()
    done, stats, owned = rb.run_job(bs, str(out), device=0, writer="merge", batch_vars=2048, options={"timing": 0}, blockfile=chromosome["blocks"])
    bs.close()
    merged = ["merged_blocks_sam.mtx", "merged_blocks_scm.mtx", "merged_blocks.mdim", "merged_blocks.ixs"]
    got = {f: open(out / f, "rb").read() for f in merged}
    for f in merged:
        os.remove(out / f)
    _same_files(str(chromosome["ref"]), str(out))
    cli.main(["merge-block-outputs", str(out), chromosome["blocks"]])
    assert "Missing:" in capsys.readouterr().out  # blocks without signal have no files
    for f in merged:
        assert open(out / f, "rb").read() == got[f], f
    exp = tmp_path / "mo"
    exp.mkdir()
    MO.write_mm(MO.merge(chromosome["blocks"], str(out) + "/"), str(exp / "merged_blocks"))
    for f in merged:
        assert open(exp / f, "rb").read() == got[f], f
