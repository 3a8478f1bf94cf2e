#!/usr/bin/env python3
"""Multi-GPU block driver: every LD block of a `.blocks` file through the `cusk` pipeline, one process per GPU.

The reference runs one LD block per `mps cusk` invocation and leaves the loop over blocks to the cluster scheduler
(/root/reference/README.md:62, cusk/src/cli.cpp:507-512).  Here one job does the whole file:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        ci-gwas_amd/run_blocks.py <.phen> <bfiles> <.blocks> <alpha> <max-level> <max-level-two> <depth> <outdir>

(the positional arguments of `mps cusk` without the block index).  Every rank opens the block set once
(`cusk_blockset_open`: .phen, .dim, .bim, .means, .stds, thresholds; the .bed is memory-mapped), takes its share of the
blocks -- static longest-processing-time assignment on predicted costs (`shard.assign_blocks`), or a shared counter
in the job's c10d store for `--schedule dynamic` -- and runs each one through `cusk_blockset_run_block`, which is the
very code `mps cusk` runs for a block (csrc/host/block_pipeline.h): correlation build -> skeleton stage one -> prune
-> stage two -> reduction, everything on the rank's GPU.  There is no collective in the data path.  The one exchange
is the gather of the per-block reduced results (a few hundred variables each) to rank 0 (`shard.gather_results`: RCCL
when the backend is nccl, gloo for CPU rehearsals), which writes `<outdir>/<chr>_<first>_<last>.{mdim,ixs,adj,corr,sep}`
-- byte-identical to per-block `mps cusk` runs, so `merge-block-outputs` consumes them unchanged
(cusk_postprocessing/merge_blocks.py:361-395; skipped blocks simply have no files, :371-386).

Small blocks do not fill an MI355X (a 500-SNP block is a chain of launch-latency-bound kernels), so a rank keeps
`--inflight` blocks going at once: that many engines (own streams, own scratch) on the same GPU, one host thread each.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

if __package__ in (None, ""):  # run as a script (torchrun): make `cigwas_amd` importable
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import cigwas_amd  # noqa: F401
    from cigwas_amd import shard
    from cigwas_amd._lib import CuskBatchStats, CuskBlockStats, lib
    from cigwas_amd.skeleton import Engine
else:
    from . import shard
    from ._lib import CuskBatchStats, CuskBlockStats, lib
    from .skeleton import Engine


class BlockSet:
    """cusk_blockset_* of include/cusk_hip.h: the inputs of `mps cusk`, opened once for all blocks."""

    def __init__(self, phen: str, bfiles: str, blocks: str, alpha: float, max_level: int, max_level_two: int, depth: int):
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = lib().cusk_blockset_open(C.byref(h), phen.encode(), bfiles.encode(), blocks.encode(), float(np.float32(alpha)),
                                      int(max_level), int(max_level_two), int(depth), err, len(err))
        if rc != 0:
            raise RuntimeError(f"cusk_blockset_open: {err.value.decode()}")
        self.h = h
        self._engines = {}  # device -> engines of this block set (run_rank creates them on first use, close() ends them)
        self.num_blocks = lib().cusk_blockset_num_blocks(h)
        self.num_samples = int(lib().cusk_blockset_num_samples(h))
        self.num_phen = lib().cusk_blockset_num_phen(h)

    def close(self):
        for engines in getattr(self, "_engines", {}).values():
            for e in engines:
                if self.h and e.h:
                    lib().cusk_blockset_release_engine(self.h, e.h)
                e.close()
        self._engines = {}
        if self.h:
            lib().cusk_blockset_close(self.h)
            self.h = None

    def markers(self, i: int) -> int:
        return int(lib().cusk_blockset_block_markers(self.h, i))

    def stem(self, i: int) -> str:
        buf = C.create_string_buffer(256)
        if lib().cusk_blockset_block_stem(self.h, i, buf, len(buf)) != 0:
            raise IndexError(i)
        return buf.value.decode()

    def stage(self, eng: Engine) -> bool:
        """cusk_blockset_stage: the whole .bed, the phenotypes, means and stds to the engine's GPU, once"""
        return lib().cusk_blockset_stage(self.h, eng.h) == 0

    def costs(self) -> list[float]:
        return [shard.predicted_cost(self.markers(i), self.num_samples, self.num_phen) for i in range(self.num_blocks)]

    def run_batch(self, eng: Engine, blocks):
        """cusk_blockset_run_batch: every block of `blocks` in one set of device runs -> (BatchResult, CuskBatchStats)"""
        ix = np.ascontiguousarray(blocks, np.int32)
        res = C.c_void_p()
        st = CuskBatchStats()
        rc = lib().cusk_blockset_run_batch(self.h, eng.h, ix.ctypes.data_as(C.c_void_p), len(ix), C.byref(res), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"batch {list(blocks)[:4]}...: {lib().cusk_blockset_last_error().decode()}")
        return BatchResult(res), st

    def run_block(self, eng: Engine, i: int, next_block: int = -1):
        """-> (shard.BlockResult | None if the block is skipped, CuskBlockStats).  next_block: the block this engine
        runs next (its correlation matrix is then built beside this block's sweeps), -1 = unknown / none"""
        res = C.c_void_p()
        st = CuskBlockStats()
        rc = lib().cusk_blockset_run_block_next(self.h, eng.h, int(i), int(next_block), C.byref(res), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"block {i}: {lib().cusk_blockset_last_error().decode()}")
        if not res:
            return None, st
        try:
            nv, nph, ml = C.c_longlong(), C.c_longlong(), C.c_longlong()
            lib().cusk_block_result_dims(res, C.byref(nv), C.byref(nph), C.byref(ml))
            k, ml_ = int(nv.value), int(ml.value)

            def arr(ptr, ctype, count, dtype):
                if count == 0:
                    return np.zeros(0, dtype)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).astype(dtype, copy=True)

            br = shard.BlockResult(
                int(i), lib().cusk_block_result_stem(res).decode(), int(nph.value), ml_,
                arr(lib().cusk_block_result_ixs(res), C.c_int32, k, np.int32),
                arr(lib().cusk_block_result_adj(res), C.c_int32, k * k, np.int32).reshape(k, k),
                arr(lib().cusk_block_result_corr(res), C.c_float, k * k, np.float32).reshape(k, k),
                arr(lib().cusk_block_result_sep(res), C.c_int32, k * k * ml_, np.int32).reshape(k, k, ml_))
        finally:
            lib().cusk_block_result_free(res)
        return br, st


    def run_block_to_files(self, eng: Engine, i: int, outdir: str, next_block: int = -1):
        """One block the way `mps cusk` runs it (cli.cpp:521-677): pipeline, then the five result files written by the library
        straight from its result (no copy into numpy arrays).  -> (written: bool, CuskBlockStats, seconds spent writing)"""
        import time

        res = C.c_void_p()
        st = CuskBlockStats()
        rc = lib().cusk_blockset_run_block_next(self.h, eng.h, int(i), int(next_block), C.byref(res), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"block {i}: {lib().cusk_blockset_last_error().decode()}")
        if not res:
            return False, st, 0.0
        try:
            t0 = time.perf_counter()
            if lib().cusk_block_result_write(res, outdir.encode()) != 0:
                raise RuntimeError(f"block {i}: {lib().cusk_blockset_last_error().decode()}")
            dt = time.perf_counter() - t0
        finally:
            lib().cusk_block_result_free(res)
        return True, st, dt


class BatchResult:
    """cusk_batch_result of include/cusk_hip.h: the reduced results of the blocks of one batch, owned by the library"""

    def __init__(self, handle):
        self.h = handle
        self.count = lib().cusk_batch_result_count(handle)
        self.block_indices = [lib().cusk_batch_result_block_index(handle, i) for i in range(self.count)]

    def write(self, outdir: str) -> None:
        if lib().cusk_batch_result_write(self.h, outdir.encode()) != 0:
            raise RuntimeError(f"writing batch results: {lib().cusk_blockset_last_error().decode()}")

    def pack(self, with_sep=True) -> np.ndarray:
        """the results as one byte string (shard.BlockResult.pack layout, block after block); with_sep = False leaves the
        .sep arrays out (nine tenths of the bytes; the merge does not read them)"""
        form = 2 if with_sep == 2 else (1 if with_sep else 0)  # 2: the separating sets as a list (C readers only)
        nbytes = int(lib().cusk_batch_result_packed_bytes_ex(self.h, form))
        buf = np.zeros(nbytes, np.uint8)
        if nbytes and lib().cusk_batch_result_pack_ex(self.h, buf.ctypes.data_as(C.c_void_p), nbytes, form) != 0:
            raise RuntimeError("cusk_batch_result_pack failed")
        return buf

    def results(self) -> list:
        buf, pos, out = self.pack(), 0, []
        while pos < buf.size:
            r, pos = shard.BlockResult.unpack(buf, pos)
            out.append(r)
        return out

    def free(self):
        if self.h:
            lib().cusk_batch_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def write_packed(buf: np.ndarray, outdir: str) -> int:
    """cusk_packed_results_write: the files of every block in a packed byte string (rank 0 after the gather)"""
    n = C.c_int(0)
    buf = np.ascontiguousarray(buf, np.uint8)
    if lib().cusk_packed_results_write(buf.ctypes.data_as(C.c_void_p), buf.size, outdir.encode(), C.byref(n)) != 0:
        raise RuntimeError(f"writing gathered results: {lib().cusk_blockset_last_error().decode()}")
    return int(n.value)


def merge_packed(blockfile: str, buf: np.ndarray, basepath: str) -> None:
    """cusk_merge_packed: `merge-block-outputs` on the gathered results in memory -> <basepath>_sam.mtx, _scm.mtx, .mdim, .ixs"""
    buf = np.ascontiguousarray(buf, np.uint8)
    if lib().cusk_merge_packed(blockfile.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, basepath.encode()) != 0:
        raise RuntimeError(f"merging the gathered results: {lib().cusk_blockset_last_error().decode()}")


def make_batches(blocks: list[int], sizes: dict, num_phen: int, batch_vars: int) -> list[list[int]]:
    """Consecutive blocks of `blocks` whose padded variable counts -- (markers + traits) rounded up to 64 -- sum to at most
    batch_vars (a single block larger than that is a batch of its own).  The batch's correlation matrix takes
    4 x (that sum)^2 bytes of HBM; one level loop serves the whole batch."""
    out, cur, tot = [], [], 0
    for b in blocks:
        v = (sizes[b] + num_phen + 63) // 64 * 64
        if cur and tot + v > batch_vars:
            out.append(cur)
            cur, tot = [], 0
        cur.append(b)
        tot += v
    if cur:
        out.append(cur)
    return out


class _Queue:
    """Where a rank's workers get their next block from: the rank's own list (static assignment) or one counter
    shared by all ranks in the job's c10d store (dynamic: blocks are handed out in descending predicted cost)."""

    def __init__(self, order: list[int], store=None, key: str = "cusk_next_block"):
        self.order, self.store, self.key = order, store, key
        self.lock = threading.Lock()
        self.pos = 0

    def next(self):
        if self.store is not None:
            k = int(self.store.add(self.key, 1)) - 1
        else:
            with self.lock:
                k = self.pos
                self.pos += 1
        return self.order[k] if k < len(self.order) else None

    def peek(self):
        """the block next() would return now, or -1 when that is not known (shared counter, end of the list)"""
        if self.store is not None:
            return -1
        with self.lock:
            return self.order[self.pos] if self.pos < len(self.order) else -1


def run_rank_batched(bs, queue: "_Queue", device: int, batch_vars: int, options: dict | None = None, write_dir: str | None = None,
                     engine_factory=None, inflight: int = 1):
    """The rank's blocks in batches of at most `batch_vars` padded variables, each batch through cusk_blockset_run_batch
    (one level loop per stage for the whole batch).  -> (list of BatchResult, list of CuskBatchStats).  write_dir: the
    files of a batch are written (by the library, cusk_batch_result_write) on a thread of their own beside the next batch."""
    inflight = max(1, int(inflight))
    if engine_factory is not None:  # CPU tests of the batching / gather logic with a stand-in block set
        engines = [engine_factory() for _ in range(inflight)]
    else:
        cache = bs._engines.setdefault(device, [])
        while len(cache) < inflight:
            cache.append(Engine(device))
        engines = cache[:inflight]
    for eng in engines:
        for k, v in (options or {}).items():
            if k != "corr_ahead":
                eng.set_option(k, int(v))
    if engine_factory is None and not bs.stage(engines[0]):
        raise RuntimeError("batched runs need the block set's inputs on the device (cusk_blockset_stage failed)")
    def batches():
        """batches drawn from the queue as they are needed (with the shared counter of the dynamic schedule a rank must not
        take more than it is about to run); a block that does not fit the current batch opens the next one"""
        carry = None
        while True:
            cur, tot = [], 0
            while True:
                b = carry if carry is not None else queue.next()
                carry = None
                if b is None:
                    break
                v = (bs.markers(b) + bs.num_phen + 63) // 64 * 64
                if cur and tot + v > batch_vars:
                    carry = b
                    break
                cur.append(b)
                tot += v
            if not cur:
                return
            yield cur

    results, stats, errors = [], [], []
    import queue as _q

    wq = _q.Queue() if write_dir is not None else None

    def writer():
        while True:
            r = wq.get()
            if r is None:
                return
            try:
                r.write(write_dir)
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

    wt = threading.Thread(target=writer) if wq is not None else None
    if wt is not None:
        wt.start()
    # `inflight` batches at a time, one host thread and one engine each: the host part of a batch (tables, prefilter,
    # pruning, reduction: about half of its wall clock) runs beside the device part of another
    gen = batches()
    glock = threading.Lock()

    def worker(eng):
        try:
            while not errors:
                with glock:
                    batch = next(gen, None)
                if batch is None:
                    return
                br, st = bs.run_batch(eng, batch)
                with glock:
                    results.append(br)
                    stats.append(st)
                if wq is not None:
                    wq.put(br)
        except Exception as exc:  # noqa: BLE001 -- re-raised on the calling thread
            errors.append(exc)

    try:
        if inflight == 1:
            worker(engines[0])
        else:
            threads = [threading.Thread(target=worker, args=(e_,)) for e_ in engines]
            for t_ in threads:
                t_.start()
            for t_ in threads:
                t_.join()
    finally:
        if wt is not None:
            wq.put(None)
            wt.join()
    if errors:
        raise errors[0]
    return results, stats


def gather_packed(payload: np.ndarray, device=None, group=None):
    """every rank's packed results -> rank 0 (list of byte arrays, one per rank; None elsewhere): a size all_gather and
    one padded all_gather -- RCCL when the group's backend is nccl -- as shard.gather_results"""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else "cpu"
    size = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, size, group=group)
    sizes = [int(v) for v in sizes.cpu()]
    cap = max(1, max(sizes))
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if payload.size:
        mine[: payload.size] = torch.from_numpy(payload).to(dev)
    # a GATHER, not an all-gather: only rank 0 needs the results (over RCCL: grouped send / recv into rank 0, each byte
    # crosses one xGMI link once)
    bufs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bufs, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank != 0:
        return None
    return [bufs[r][: sizes[r]].cpu().numpy() for r in range(world)]


def run_rank(bs, queue: _Queue, device: int, inflight: int = 1, options: dict | None = None, engine_factory=None,
             stage: bool = True, write_dir: str | None = None):
    """Runs blocks from `queue` on GPU `device` with `inflight` engines (one host thread each).
    -> (results sorted by block index, {block index: CuskBlockStats}).  `engine_factory` exists for the CPU tests of
    the scheduling / gather logic (a stand-in block set that needs no device); the product always runs on Engine."""
    inflight = max(1, int(inflight))
    # engines live as long as the block set (a job that runs several passes, or a service, creates them once: creating
    # and destroying an engine costs ~10 ms -- streams, events, pinned mirrors, and the device buffers of the first blocks)
    cache = getattr(bs, "_engines", None) if engine_factory is None else None
    if cache is not None:
        engines = cache.setdefault(device, [])
        while len(engines) < inflight:
            engines.append(Engine(device))
        engines = engines[:inflight]
    else:
        engines = [engine_factory() if engine_factory else Engine(device) for _ in range(inflight)]
    for e in engines:
        for k, v in (options or {}).items():
            if k != "corr_ahead":  # (the driver's own switch: correlation build of the next block beside this block's sweeps)
                e.set_option(k, int(v))
    if stage and hasattr(bs, "stage"):
        bs.stage(engines[0])  # best effort: without it every block uploads its own slice
    results, stats, errors = [], {}, []
    lock = threading.Lock()
    # write_dir: the files of a block are written by a thread of their own as soon as the block is done, beside the
    # engine's work on the next block (the engine calls release the interpreter lock)
    import queue as _q

    wq = _q.Queue() if write_dir is not None else None

    def writer():
        while True:
            r = wq.get()
            if r is None:
                return
            try:
                r.write(write_dir)
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

    wt = threading.Thread(target=writer) if wq is not None else None
    if wt is not None:
        wt.start()
    ahead = engine_factory is None and stage and int((options or {}).get("corr_ahead", 1)) != 0

    def worker(eng):
        try:
            while True:
                b = queue.next()
                if b is None or errors:
                    return
                # one engine per rank: the block it takes next is known, and its correlations are built ahead
                nxt = queue.peek() if (inflight == 1 and ahead) else -1
                br, st = bs.run_block(eng, b, nxt) if nxt >= 0 else bs.run_block(eng, b)
                with lock:
                    stats[b] = st
                    if br is not None:
                        results.append(br)
                        if wq is not None:
                            wq.put(br)
        except Exception as exc:  # noqa: BLE001 -- re-raised on the calling thread
            errors.append(exc)

    if inflight == 1:
        worker(engines[0])
    else:
        threads = [threading.Thread(target=worker, args=(e,)) for e in engines]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if wt is not None:
        wq.put(None)
        wt.join()
    if cache is None:
        for e in engines:
            e.close()
    if errors:
        raise errors[0]
    return sorted(results, key=lambda r: r.block_index), stats


def run_job(bs, outdir: str | None, device: int, inflight: int = 1, schedule: str = "lpt", collective_device=None,
            options: dict | None = None, group=None, engine_factory=None, store_key: str = "cusk_next_block",
            stage: bool = True, writer: str = "rank0", batch_vars: int = 0, timings: dict | None = None, blockfile: str | None = None):
    """One rank's part of the job (call on every rank of an initialised process group, or without one for a
    single-process run).  Returns (all results on rank 0 / None elsewhere, this rank's stats, assignment).

    writer = "rank0": the per-block results are gathered to rank 0 (one exchange at the end of the job), which writes
    every file.  writer = "local": every rank writes the files of its own blocks into `outdir` (a directory all ranks
    of the node see) -- what the reference's one-process-per-block runs do; no exchange at all, only a barrier, and the
    writing is spread over the ranks instead of serialised on one (returns this rank's results).  writer = "merge"
    (batched execution; needs `blockfile`): both -- every rank writes its own blocks' files, and what `merge-block-outputs`
    reads of them (indices, adjacency, correlations: no separating sets, a tenth of the bytes) is gathered to rank 0,
    which writes the MERGED skeleton `<outdir>/merged_blocks{_sam.mtx,_scm.mtx,.mdim,.ixs}` straight from memory
    (cusk_merge_packed): the job ends with the input of `cuskss-merged` / `sepselect` on disk and rank 0 writes four small
    files instead of five per block."""
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    costs = bs.costs()
    if schedule == "dynamic" and distributed and world > 1:
        order = sorted(range(bs.num_blocks), key=lambda b: (-costs[b], b))
        store = dist.distributed_c10d._get_default_store()
        queue = _Queue(order, store, store_key)
        owned = None
    else:
        owned = shard.assign_blocks(costs, world)[rank]
        queue = _Queue(sorted(owned, key=lambda b: (-costs[b], b)))  # big blocks first within the rank as well
    if batch_vars > 0 and hasattr(bs, "run_batch"):
        # batched execution: the rank's blocks in batches, one level loop per stage and batch.  Returns the block indices
        # that were written (all of the job's on rank 0 with writer = rank0, the rank's own with writer = local), the
        # per-batch stats, and the assignment
        t0 = time.perf_counter()
        if writer == "merge" and blockfile is None:
            raise ValueError("writer = 'merge' needs the job's .blocks file")
        bres, bstats = run_rank_batched(bs, queue, device, batch_vars, options, write_dir=outdir if writer in ("local", "merge") else None,
                                        engine_factory=engine_factory, inflight=inflight)
        done = sorted(b for r in bres for b in r.block_indices)
        t1 = time.perf_counter()
        if writer == "local":
            if distributed:
                dist.barrier(group)
        elif writer == "merge":
            payload = np.concatenate([r.pack(with_sep=False) for r in bres]) if bres else np.zeros(0, np.uint8)
            parts = gather_packed(payload, device=collective_device, group=group) if distributed else [payload]
            t2 = time.perf_counter()
            if timings is not None:
                timings["gather_s"] = t2 - t1
            if rank == 0 and outdir is not None:
                allp = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
                if allp.size:
                    merge_packed(blockfile, allp, os.path.join(outdir, "merged_blocks"))
            if timings is not None:
                timings["write_s"] = time.perf_counter() - t2
            if distributed:
                dist.barrier(group)  # every rank's own files are on disk when the job returns
        else:
            # (with_sep = 2: the separating sets travel as a list -- a twentieth of the dense arrays; cusk_packed_results_write
            # on rank 0 streams the .sep files from it)
            payload = np.concatenate([r.pack(with_sep=2) for r in bres]) if bres else np.zeros(0, np.uint8)
            if distributed:
                parts = gather_packed(payload, device=collective_device, group=group)
            else:
                parts = [payload]
            t2 = time.perf_counter()
            if timings is not None:
                timings["gather_s"] = t2 - t1
            if rank == 0:
                done = []
                for part in parts:
                    pos = 0
                    while pos < part.size:  # block indices of what arrived (headers only)
                        head = part[pos:pos + 24].view(np.int32)
                        bi, k, _nph, ml, form, ns = (int(v) for v in head)
                        done.append(bi)
                        pos += 24 + ns + 4 * (k + 2 * k * k)
                        if form == 1:  # dense separating sets
                            pos += 4 * k * k * ml
                        elif form == 2:  # a list: int32 count + count records of 17 int32 (include/cusk_hip.h)
                            pos += 4 + 68 * int(part[pos:pos + 4].view(np.int32)[0])
                    if outdir is not None and part.size:
                        write_packed(part, outdir)
                done.sort()
            else:
                done = None
            if timings is not None:
                timings["write_s"] = time.perf_counter() - t2
        if timings is not None:
            timings["compute_s"] = t1 - t0
        for r in bres:
            r.free()
        return done, bstats, (owned if owned is not None else None)
    results, stats = run_rank(bs, queue, device, inflight, options, engine_factory, stage,
                              write_dir=outdir if writer == "local" else None)
    if writer == "local":
        if distributed:
            dist.barrier(group)
        return results, stats, (owned if owned is not None else sorted(stats))
    if distributed:
        allr = shard.gather_results(results, device=collective_device, group=group)
    else:
        allr = results
    if rank == 0 and outdir is not None:
        for r in allr:
            r.write(outdir)
    return allr, stats, (owned if owned is not None else sorted(stats))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="run_blocks", description="cusk on every LD block of a .blocks file, sharded over the GPUs of one node")
    ap.add_argument("phen")
    ap.add_argument("bfiles")
    ap.add_argument("blocks")
    ap.add_argument("alpha", type=float)
    ap.add_argument("max_level", type=int)
    ap.add_argument("max_level_two", type=int)
    ap.add_argument("depth", type=int)
    ap.add_argument("outdir")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo: CPU collectives)")
    ap.add_argument("--inflight", type=int, default=1, help="blocks in flight per GPU")
    ap.add_argument("--schedule", choices=["lpt", "dynamic"], default="lpt")
    ap.add_argument("--writer", choices=["merge", "rank0", "local"], default="rank0",
                    help="merge (batched execution): every rank writes the files of its own blocks, the merged skeleton is gathered "
                         "to rank 0 (RCCL) and written there as merged_blocks*; rank0: full results gathered to rank 0, which writes "
                         "every file; local: every rank writes its own blocks' files, no exchange")
    ap.add_argument("--batch-vars", type=int, default=16384,
                    help="blocks are run in batches of at most this many (padded) variables, one level loop per stage for the "
                         "whole batch; 0: one block per engine run")
    ap.add_argument("--no-stage", action="store_true", help="do not keep the whole .bed in HBM; every block uploads its slice")
    ap.add_argument("--device", type=int, default=None, help="GPU of this rank (default LOCAL_RANK modulo the device count)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    ndev = max(torch.cuda.device_count(), 1)
    device = args.device if args.device is not None else local_rank % ndev
    cdev = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            cdev = torch.device("cuda", device)
        else:
            dist.init_process_group(args.backend)
    if not os.path.isdir(args.outdir):
        sys.exit(f"file or directory not found: {args.outdir}")
    t0 = time.perf_counter()
    bs = BlockSet(args.phen, args.bfiles, args.blocks, args.alpha, args.max_level, args.max_level_two, args.depth)
    t_open = time.perf_counter() - t0
    # no per-level HIP events: nothing here reads the per-level kernel times, and every event costs the launch-bound
    # small blocks a few microseconds of device time
    batch_vars = 0 if args.no_stage else max(0, args.batch_vars)
    writer = args.writer if (batch_vars > 0 or args.writer != "merge") else "rank0"  # (merge is part of the batched path)
    allr, stats, owned = run_job(bs, args.outdir, device, args.inflight, args.schedule, cdev, stage=not args.no_stage,
                                 options={"timing": 0}, writer=writer, batch_vars=batch_vars, blockfile=args.blocks)
    dt = time.perf_counter() - t0
    if batch_vars > 0:
        tests = sum(int(s.tests[0]) + int(s.tests[1]) for s in stats)
        nblk, nskip = sum(int(s.blocks) for s in stats), sum(int(s.skipped) for s in stats)
    else:
        tests = sum(int(s.tests[0]) + int(s.tests[1]) for s in stats.values())
        nblk, nskip = len(stats), sum(1 for s in stats.values() if s.skipped)
    print(f"[rank {rank}/{world}] gpu {device}: {nblk} blocks ({nskip} skipped), "
          f"{tests:.3e} CI tests, open {t_open:.2f} s, total {dt:.2f} s", flush=True)
    if rank == 0 or writer in ("local", "merge"):
        print(f"[rank {rank}] wrote {len(allr)} of {bs.num_blocks} blocks to {args.outdir}"
              + (" + merged_blocks*" if (writer == "merge" and rank == 0) else ""), flush=True)
    bs.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
