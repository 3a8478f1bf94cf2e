"""CPU, world_size 2 over gloo: the multi-GPU path (block assignment + the one gather) without GPUs.
Block results are produced by the oracle here (this is a test of the sharding logic, not of the kernels)."""
import os
import socket

import numpy as np
import pytest


def test_assignment_is_a_partition_and_balanced():
    from cigwas_amd import shard

    rng = np.random.default_rng(0)
    sizes = rng.integers(200, 11000, 200)
    costs = [shard.predicted_cost(int(m), 16384, 20) for m in sizes]
    for world in (1, 2, 4, 8):
        owned = shard.assign_blocks(costs, world)
        flat = sorted(b for o in owned for b in o)
        assert flat == list(range(200))
        loads = [sum(costs[b] for b in o) for o in owned]
        assert max(loads) <= 1.1 * (sum(costs) / world) + max(costs)
        assert owned == shard.assign_blocks(costs, world)  # deterministic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, nblocks):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from cigwas_amd import shard, synth
    from oracle import oracle as O

    sizes = [60 + 17 * b for b in range(nblocks)]
    owned = shard.assign_blocks([shard.predicted_cost(m, 2000, 4) for m in sizes], world)[rank]
    results = []
    for b in owned:
        Cm = synth.synth_corr_block(sizes[b], 4, N=2000, block_index=b)
        red = O.cusk_from_corr(Cm, 4, O.threshold_array(2000, 1e-3), 2, 3, 1)
        results.append(shard.BlockResult(b, f"1_{b}_{b}", 4, red.max_level, red.new_to_old, red.G, red.C, red.S))
    allr = shard.gather_results(results)
    if rank == 0:
        assert [r.block_index for r in allr] == list(range(nblocks))
        for r in allr:
            r.write(outdir)
    else:
        assert allr is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_writes_the_same_files_as_one_process(tmp_path, oracle, synth):
    import torch.multiprocessing as mp

    from cigwas_amd import shard

    nblocks = 5
    out2 = tmp_path / "w2"
    out2.mkdir()
    mp.spawn(_worker, args=(2, _free_port(), str(out2), nblocks), nprocs=2, join=True)
    # single-process reference run of the same blocks
    for b in range(nblocks):
        m = 60 + 17 * b
        Cm = synth.synth_corr_block(m, 4, N=2000, block_index=b)
        red = oracle.cusk_from_corr(Cm, 4, oracle.threshold_array(2000, 1e-3), 2, 3, 1)
        oracle.write_reduced(red, str(tmp_path / f"ref_{b}"), with_sep=True)
        for ext in (".mdim", ".ixs", ".adj", ".corr", ".sep"):
            assert open(str(out2 / f"1_{b}_{b}") + ext, "rb").read() == open(str(tmp_path / f"ref_{b}") + ext, "rb").read()
    # pack/unpack round trip incl. the sepset-free (cuskss) form
    br = shard.BlockResult(3, "x_1_2", 2, 14, np.arange(4, dtype=np.int32), np.eye(4, dtype=np.int32),
                           np.eye(4, dtype=np.float32))
    back, pos = shard.BlockResult.unpack(br.pack(), 0)
    assert pos == br.pack().size and back.sep is None and np.array_equal(back.adj, br.adj) and back.stem == "x_1_2"


def _min_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cigwas_amd import shard

    ok = True
    for dtype, none, width in ((np.int64, -1, 62), (np.int32, -1, 30)):
        rng = np.random.default_rng(5)  # same stream on both ranks: rank r keeps column r
        vals = rng.integers(0, 2 ** width, size=(1000, world)).astype(dtype)
        vals[rng.random((1000, world)) < 0.4] = none  # "no separating set": all ones
        mine = vals[:, rank].copy()
        shard.unsigned_min_allreduce_(torch.from_numpy(mine))
        un = vals.astype(np.uint64) if dtype == np.int64 else vals.astype(np.uint32)
        ok = ok and np.array_equal(mine.view(un.dtype), un.min(axis=1))
    # the exchange function on a host-staged buffer, as the engine hands it over
    buf = np.array([7, -1, 3, -1, 5] if rank == 0 else [9, -1, 2, 4, -1], np.int64)
    ex = shard.make_min_exchange()
    ok = ok and ex(2, buf.ctypes.data, 5, 8, False, None) == 0 and buf.tolist() == [7, -1, 2, 4, 5]
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_unsigned_min_exchange_of_the_row_sharded_sweep():
    """the join of a row-sharded level is an element-wise UNSIGNED minimum with all ones = none"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_min_worker, args=(2, _free_port(), q), nprocs=2, join=True)
    assert q.get() and q.get()
