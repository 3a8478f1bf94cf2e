"""CPU, world_size 2 over gloo: the block driver's scheduling (static LPT and the shared-counter dynamic queue), its
worker threads and the one gather, with a stand-in block set whose blocks are computed by the oracle (the kernels are
not under test here; tests/test_gpu_run_blocks.py runs the same driver on the real engine)."""
import os
import socket

import numpy as np
import pytest

NBLOCKS = 7
SIZES = [60 + 23 * ((5 * b) % NBLOCKS) for b in range(NBLOCKS)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Stats:
    def __init__(self, skipped):
        self.skipped = skipped
        self.tests = [1, 1]


class _NoEngine:
    def set_option(self, k, v):
        pass

    def close(self):
        pass


class _OracleBlockSet:
    """block b: synthetic correlation block of SIZES[b] markers + 4 traits; block 3 is 'skipped'"""
    num_blocks, num_samples, num_phen = NBLOCKS, 2000, 4

    def markers(self, i):
        return SIZES[i]

    def costs(self):
        from cigwas_amd import shard

        return [shard.predicted_cost(m, self.num_samples, self.num_phen) for m in SIZES]

    def run_block(self, eng, b):
        from cigwas_amd import shard, synth
        from oracle import oracle as O

        if b == 3:
            return None, _Stats(1)
        Cm = synth.synth_corr_block(SIZES[b], 4, N=2000, block_index=b)
        red = O.cusk_from_corr(Cm, 4, O.threshold_array(2000, 1e-3), 2, 3, 1)
        return shard.BlockResult(b, f"1_{b}_{b}", 4, red.max_level, red.new_to_old, red.G, red.C, red.S), _Stats(0)


class _OracleBatch:
    """stand-in for run_blocks.BatchResult: the results of a batch, computed by the oracle block by block"""

    def __init__(self, results):
        self._r = results
        self.block_indices = [r.block_index for r in results]

    def pack(self, with_sep=True):
        import copy

        rs = self._r
        if not with_sep:
            rs = [copy.copy(r) for r in rs]
            for r in rs:
                r.sep = None
        return np.concatenate([r.pack() for r in rs]) if rs else np.zeros(0, np.uint8)

    def write(self, outdir):
        for r in self._r:
            r.write(outdir)

    def free(self):
        pass


class _BatchStats:
    def __init__(self, blocks, skipped):
        self.blocks, self.skipped, self.tests = blocks, skipped, [1, 1]


class _OracleBatchSet(_OracleBlockSet):
    def run_batch(self, eng, blocks):
        res = [self.run_block(eng, b)[0] for b in blocks]
        return _OracleBatch([r for r in res if r is not None]), _BatchStats(len(blocks), sum(r is None for r in res))


def _batch_worker(rank, world, port, outdir, mode, q):
    import sys

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cigwas_amd import run_blocks as rb

    schedule, writer, bv = mode.split("+")
    tm = {}
    blockfile = os.path.join(outdir, "..", "b.blocks")
    done, stats, owned = rb.run_job(_OracleBatchSet(), outdir, device=0, schedule=schedule, engine_factory=_NoEngine,
                                    store_key=f"nextb_{mode}", writer=writer, batch_vars=int(bv), timings=tm, blockfile=blockfile)
    if writer == "rank0":
        assert (done is None) == (rank != 0) and "gather_s" in tm
        if rank == 0:
            assert done == [b for b in range(NBLOCKS) if b != 3]
    q.put((rank, sum(s_.blocks for s_ in stats), len(stats)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["lpt+rank0+300", "dynamic+rank0+200", "lpt+local+100000", "lpt+merge+250"])
def test_two_rank_batched_job_gathers_packed_results(tmp_path, oracle, synth, mode):
    """the batched job path (run_blocks.run_job with batch_vars > 0) over gloo: batches cut from the rank's queue, ONE gather
    of the packed results to rank 0, files written by the library's packed writer (host code of libcusk_hip.so)"""
    import torch.multiprocessing as mp

    out = tmp_path / "o"
    out.mkdir()
    with open(tmp_path / "b.blocks", "w") as f:  # the stand-in blocks are called 1_<b>_<b>
        f.write("".join(f"1\t{b}\t{b}\n" for b in range(NBLOCKS)))
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_batch_worker, args=(2, _free_port(), str(out), mode, q), nprocs=2, join=True)
    got = [q.get() for _ in range(2)]
    assert sum(g[1] for g in got) == NBLOCKS  # every block in exactly one batch of one rank
    if mode.endswith("+100000"):
        assert all(g[2] <= 1 for g in got)  # one batch per rank
    else:
        assert sum(g[2] for g in got) >= 3
    if mode.split("+")[1] == "merge":
        # the merged skeleton rank 0 wrote from the gathered results = the merge of the per-block files (oracle restatement,
        # pinned by the files the reference wrote)
        from oracle import merge_oracle as MO

        MO.write_mm(MO.merge(str(tmp_path / "b.blocks"), str(out) + "/"), str(tmp_path / "exp"))
        for sfx in ("_sam.mtx", "_scm.mtx", ".mdim", ".ixs"):
            assert open(str(out / "merged_blocks") + sfx, "rb").read() == open(str(tmp_path / "exp") + sfx, "rb").read(), sfx
            os.remove(str(out / "merged_blocks") + sfx)
    stems = sorted({f.rsplit(".", 1)[0] for f in os.listdir(out)})
    assert stems == [f"1_{b}_{b}" for b in range(NBLOCKS) if b != 3]
    for b in (1, 5):
        Cm = synth.synth_corr_block(SIZES[b], 4, N=2000, block_index=b)
        red = oracle.cusk_from_corr(Cm, 4, oracle.threshold_array(2000, 1e-3), 2, 3, 1)
        oracle.write_reduced(red, str(tmp_path / f"ref_{b}"), with_sep=True)
        for ext in (".mdim", ".ixs", ".adj", ".corr", ".sep"):
            assert open(str(out / f"1_{b}_{b}") + ext, "rb").read() == open(str(tmp_path / f"ref_{b}") + ext, "rb").read()


def _worker(rank, world, port, outdir, schedule, q):
    import sys

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cigwas_amd import run_blocks as rb

    writer = "local" if schedule.endswith("+local") else "rank0"
    schedule = schedule.split("+")[0]
    allr, stats, owned = rb.run_job(_OracleBlockSet(), outdir, device=0, inflight=2, schedule=schedule,
                                    engine_factory=_NoEngine, store_key=f"next_{schedule}_{writer}", writer=writer)
    if writer == "rank0":
        assert (allr is None) == (rank != 0)
    else:  # every rank wrote the files of its own blocks and returns them
        assert sorted(r.block_index for r in allr) == [b for b in sorted(stats) if b != 3]
    q.put((rank, sorted(stats)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("schedule", ["lpt", "dynamic", "lpt+local"])
def test_two_rank_block_driver_partitions_and_gathers(tmp_path, oracle, synth, schedule):
    import torch.multiprocessing as mp

    from cigwas_amd import shard

    out = tmp_path / schedule
    out.mkdir()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_worker, args=(2, _free_port(), str(out), schedule, q), nprocs=2, join=True)
    ran = dict(q.get() for _ in range(2))
    assert sorted(ran[0] + ran[1]) == list(range(NBLOCKS))  # a partition: every block exactly once
    if schedule.startswith("lpt"):
        owned = shard.assign_blocks(_OracleBlockSet().costs(), 2)
        assert ran[0] == owned[0] and ran[1] == owned[1]
    stems = sorted({f.rsplit(".", 1)[0] for f in os.listdir(out)})
    assert stems == [f"1_{b}_{b}" for b in range(NBLOCKS) if b != 3]
    for b in (0, 6):
        Cm = synth.synth_corr_block(SIZES[b], 4, N=2000, block_index=b)
        red = oracle.cusk_from_corr(Cm, 4, oracle.threshold_array(2000, 1e-3), 2, 3, 1)
        oracle.write_reduced(red, str(tmp_path / f"ref_{b}"), with_sep=True)
        for ext in (".mdim", ".ixs", ".adj", ".corr", ".sep"):
            assert open(str(out / f"1_{b}_{b}") + ext, "rb").read() == open(str(tmp_path / f"ref_{b}") + ext, "rb").read()
