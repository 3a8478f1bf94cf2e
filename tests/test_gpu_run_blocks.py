"""GPU: the multi-GPU block driver (ci-gwas_amd/run_blocks.py) with the real engine.

A synthetic two-chromosome file set with unequal LD blocks goes (a) block by block through `mps cusk`, one process per
block as the reference runs it (README.md:62, cli.cpp:507-512), and (b) through the block driver under torchrun with
world_size 2 (both ranks share the box's one GPU; collectives over gloo), static LPT and dynamic scheduling, and
in-process with several blocks in flight.  Every result file must be byte-identical, skipped blocks must be skipped by
both, and a sample of blocks is checked against the oracle pipeline directly.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPS = os.path.join(ROOT, "ci-gwas_amd", "csrc", "mps")
EXTS = (".mdim", ".ixs", ".adj", ".corr", ".sep")
ALPHA, L1, L2, DEPTH = "0.0001", "3", "6", "1"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def chromosome(tmp_path_factory, synth):
    """2,900 markers on two chromosomes, 1,500 individuals, 5 traits; 11 blocks of 60..520 markers.  Traits depend on
    markers of some blocks only, so that the marginal prefilter (cli.cpp:561-576) skips the others."""
    d = tmp_path_factory.mktemp("chrom")
    N, p = 1500, 5
    sizes1 = [300, 60, 520, 180, 240, 400]   # chromosome 1: 1700 markers
    sizes2 = [200, 350, 90, 410, 150]        # chromosome 2: 1200 markers
    m = sum(sizes1) + sum(sizes2)
    rng = synth.rng_for(404)
    G = synth.make_genotypes(m, N, rng, window=50, rho=0.8, miss=0.002)
    bounds, start = [], 0
    for cid, sizes in (("1", sizes1), ("2", sizes2)):
        first = 0
        for s in sizes:
            bounds.append((cid, first, first + s - 1, start + first))
            first += s
        start += sum(sizes)
    signal_blocks = [0, 2, 3, 5, 7, 9]  # the others carry no causal marker
    Y = np.zeros((p, N))
    g = G.astype(np.float64)
    g[g < 0] = np.nan
    gs = np.nan_to_num((g - np.nanmean(g, 1, keepdims=True)) / np.maximum(np.nanstd(g, 1, keepdims=True), 1e-9))
    for k in range(p):
        for b in signal_blocks:
            _cid, f, l, g0 = bounds[b]
            idx = g0 + rng.choice(l - f + 1, size=2, replace=False)
            Y[k] += (rng.uniform(0.25, 0.4, 2) * rng.choice([-1.0, 1.0], 2)) @ gs[idx]
        if k:
            Y[k] += 0.3 * Y[k - 1]
        Y[k] += rng.standard_normal(N)
        Y[k] = (Y[k] - Y[k].mean()) / Y[k].std()
    phen = np.ascontiguousarray(Y.astype(np.float32)).reshape(-1)
    phen[3] = np.nan
    means, stds = synth.bed_stats(G)
    stem = str(d / "geno")
    synth.write_bfiles(stem, synth.pack_bed(G), N, means, stds, ["1"] * sum(sizes1) + ["2"] * sum(sizes2))
    synth.write_phen(str(d / "y.phen"), phen, N, p)
    with open(d / "c.blocks", "w") as f:
        for cid, first, last, _ in bounds:
            f.write(f"{cid}\t{first}\t{last}\n")
    # (a) one `mps cusk` process per block
    ref = d / "per_block"
    ref.mkdir()
    for b in range(len(bounds)):
        r = subprocess.run([MPS, "cusk", str(d / "y.phen"), stem, str(d / "c.blocks"), ALPHA, L1, L2, DEPTH, str(ref), str(b)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    return dict(dir=d, stem=stem, phen=str(d / "y.phen"), blocks=str(d / "c.blocks"), ref=ref, bounds=bounds,
                signal=signal_blocks, N=N, p=p)


def _same_files(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    assert fa == fb, (fa, fb)
    for f in fa:
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    return fa


def test_per_block_reference_run_is_what_we_think(chromosome):
    files = sorted(os.listdir(chromosome["ref"]))
    stems = sorted({f.rsplit(".", 1)[0] for f in files})
    kept = [f"{c}_{f}_{l}" for i, (c, f, l, _) in enumerate(chromosome["bounds"]) if i in chromosome["signal"]]
    assert set(kept) <= set(stems)           # every block with a causal marker is written ...
    assert len(stems) < len(chromosome["bounds"])  # ... and at least one block without signal is skipped
    assert len(files) == 5 * len(stems)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("schedule", ["lpt", "dynamic"])
def test_two_ranks_write_the_files_of_per_block_mps_cusk(chromosome, tmp_path, schedule):
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "ci-gwas_amd", "run_blocks.py"), chromosome["phen"],
           chromosome["stem"], chromosome["blocks"], ALPHA, L1, L2, DEPTH, str(out), "--backend", "gloo", "--inflight", "2",
           "--schedule", schedule]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "[rank 0/2]" in r.stdout and "[rank 1/2]" in r.stdout
    _same_files(str(chromosome["ref"]), str(out))


def test_in_process_driver_with_blocks_in_flight_and_oracle_check(chromosome, tmp_path, oracle):
    from cigwas_amd import run_blocks as rb

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), int(DEPTH))
    assert bs.num_blocks == len(chromosome["bounds"]) and bs.num_phen == chromosome["p"]
    assert [bs.markers(i) for i in range(bs.num_blocks)] == [l - f + 1 for _, f, l, _ in chromosome["bounds"]]
    out = tmp_path / "out"
    out.mkdir()
    allr, stats, owned = rb.run_job(bs, str(out), device=0, inflight=3)
    assert owned == list(range(bs.num_blocks)) and sorted(stats) == owned
    files = _same_files(str(chromosome["ref"]), str(out))
    skipped = [b for b in owned if stats[b].skipped]
    assert len(files) == 5 * (bs.num_blocks - len(skipped)) and skipped
    assert all(stats[b].tests[0] > 0 and stats[b].tests[1] > 0 for b in owned if not stats[b].skipped)
    # the oracle pipeline on two of the blocks, from the packed genotypes (correlation build + both stages + reduction)
    bed = np.fromfile(chromosome["stem"] + ".bed", np.uint8)[3:].reshape(-1, (chromosome["N"] + 3) // 4)
    means = np.loadtxt(chromosome["stem"] + ".means", dtype=np.float32)
    stds = np.loadtxt(chromosome["stem"] + ".stds", dtype=np.float32)
    phen = oracle.load_phen(chromosome["phen"])[2]
    N, p = chromosome["N"], chromosome["p"]
    Th = oracle.threshold_array(N, float(ALPHA))
    for b in (chromosome["signal"][0], chromosome["signal"][-1]):
        cid, f, l, g0 = chromosome["bounds"][b]
        mb = l - f + 1
        sel = slice(g0, g0 + mb)
        o_mxm, o_mxp, o_pxp = oracle.corr_pearson_npn(bed[sel], phen, mb, N, p, means[sel], stds[sel])
        sq = oracle.square_from_cusk_corrs(o_mxm, o_mxp, o_pxp, mb, p)
        # the SNP x trait block of the device build agrees to 1e-5, not bitwise: take the device's .corr where the
        # two differ in the last bits would change nothing here, the adjacency / indices / sepsets must be identical
        ref = oracle.cusk_from_corr(sq, p, Th, int(L1), int(L2), int(DEPTH))
        base = str(out / f"{cid}_{f}_{l}")
        assert list(np.fromfile(base + ".ixs", np.int32)) == list(ref.new_to_old)
        assert np.array_equal(np.fromfile(base + ".adj", np.int32), np.asarray(ref.G, np.int32).reshape(-1))
        assert np.array_equal(np.fromfile(base + ".sep", np.int32), np.asarray(ref.S, np.int32).reshape(-1))
        assert np.allclose(np.fromfile(base + ".corr", np.float32), np.asarray(ref.C, np.float32).reshape(-1), atol=1e-5, rtol=0)
    bs.close()


@pytest.mark.parametrize("ahead", [1, 0])
def test_correlation_build_of_the_next_block_runs_beside_the_sweeps(chromosome, tmp_path, ahead):
    """one engine per rank: the driver names the block the engine takes next and its correlation matrix is built on a
    third stream while the current block is swept (cusk_corr_build_begin / _end); files as from per-block `mps cusk`.
    Also: naming one block and then asking for another drops the build that is in flight."""
    from cigwas_amd import run_blocks as rb
    from cigwas_amd.skeleton import Engine

    bs = rb.BlockSet(chromosome["phen"], chromosome["stem"], chromosome["blocks"], float(ALPHA), int(L1), int(L2), int(DEPTH))
    out = tmp_path / "out"
    out.mkdir()
    allr, stats, owned = rb.run_job(bs, str(out), device=0, inflight=1, options={"corr_ahead": ahead, "timing": 0})
    _same_files(str(chromosome["ref"]), str(out))
    if ahead:
        e = Engine(0)
        assert bs.stage(e)
        a, b, c = chromosome["signal"][0], chromosome["signal"][1], chromosome["signal"][-1]
        out2 = tmp_path / "out2"
        out2.mkdir()
        r1, _ = bs.run_block(e, a, next_block=b)   # builds b ahead ...
        r3, _ = bs.run_block(e, c, next_block=a)   # ... but c is asked for: b's build is dropped, a is built ahead
        r2, _ = bs.run_block(e, a)                 # a again: taken from the build in flight
        r4, _ = bs.run_block(e, b)
        for r in (r1, r3, r4):
            r.write(str(out2))
        ref = {f for f in os.listdir(str(chromosome["ref"]))}
        for f in os.listdir(str(out2)):
            assert f in ref and open(os.path.join(str(out2), f), "rb").read() == open(os.path.join(str(chromosome["ref"]), f), "rb").read()
        assert np.array_equal(r1.adj, r2.adj) and np.array_equal(r1.sep, r2.sep) and np.array_equal(r1.corr, r2.corr)
        e.close()
    bs.close()


def test_blockset_reports_bad_inputs(tmp_path, chromosome):
    from cigwas_amd import run_blocks as rb

    with pytest.raises(RuntimeError, match="not found"):
        rb.BlockSet(str(tmp_path / "nope.phen"), chromosome["stem"], chromosome["blocks"], 1e-4, 3, 3, 1)
    with open(tmp_path / "bad.blocks", "w") as f:
        f.write("1\t0\t99999\n")
    with pytest.raises(RuntimeError, match="out of bounds"):
        rb.BlockSet(chromosome["phen"], chromosome["stem"], str(tmp_path / "bad.blocks"), 1e-4, 3, 3, 1)
