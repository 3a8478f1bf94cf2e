"""LD-block sharding over the GPUs of one node.

LD blocks are independent `cusk` problems (reference README.md:62, cli.cpp:507-512: one process
invocation per block), so the multi-GPU path is: one process per GPU, a static
longest-processing-time assignment of blocks to ranks, no communication during compute, and ONE
exchange at the end -- the per-block reduced outputs (new_to_old, adjacency, correlations; a few
hundred variables per block) gathered to rank 0, which writes the reference's per-block files so
that merge-block-outputs consumes them unchanged.  The payload is small and ragged, so it is
flattened per rank and moved with a size all_gather + one padded all_gather (works on RCCL, where
backend "nccl" IS RCCL, and on gloo for the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def predicted_cost(num_markers: int, num_individuals: int, num_traits: int) -> float:
    """relative cost of one block: the O(m^2 N) contingency GEMM dominates, the sweep scales ~ m"""
    m = float(num_markers)
    return m * m * float(num_individuals) + 4.0e4 * m * float(num_traits + 32)


def assign_blocks(costs, world_size: int) -> list[list[int]]:
    """Longest-processing-time first: deterministic, identical on every rank.
    Returns for each rank the ascending list of block indices it owns."""
    order = sorted(range(len(costs)), key=lambda b: (-float(costs[b]), b))
    load = [0.0] * world_size
    owned = [[] for _ in range(world_size)]
    for b in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owned[r].append(b)
        load[r] += float(costs[b])
    return [sorted(x) for x in owned]


@dataclass
class BlockResult:
    """what `mps cusk` / `cuskss` write per block (reference parent_set.h:42-52,99-108)"""
    block_index: int
    stem: str
    num_phen: int
    max_level: int
    new_to_old: np.ndarray  # int32 [k]
    adj: np.ndarray         # int32 [k, k]
    corr: np.ndarray        # float32 [k, k]
    sep: np.ndarray | None = None  # int32 [k, k, max_level]

    def pack(self) -> np.ndarray:
        k = int(self.new_to_old.size)
        stem = np.frombuffer(self.stem.encode(), np.uint8)
        head = np.array([self.block_index, k, self.num_phen, self.max_level, 0 if self.sep is None else 1, stem.size],
                        np.int32)
        parts = [head.view(np.uint8), stem, np.ascontiguousarray(self.new_to_old, np.int32).view(np.uint8),
                 np.ascontiguousarray(self.adj, np.int32).reshape(-1).view(np.uint8),
                 np.ascontiguousarray(self.corr, np.float32).reshape(-1).view(np.uint8)]
        if self.sep is not None:
            parts.append(np.ascontiguousarray(self.sep, np.int32).reshape(-1).view(np.uint8))
        return np.concatenate(parts)

    @staticmethod
    def unpack(buf: np.ndarray, pos: int) -> tuple["BlockResult", int]:
        head = buf[pos:pos + 24].view(np.int32)
        bi, k, nph, ml, has_sep, ns = (int(v) for v in head)
        pos += 24
        stem = bytes(buf[pos:pos + ns]).decode()
        pos += ns
        n2o = buf[pos:pos + 4 * k].view(np.int32).copy()
        pos += 4 * k
        adj = buf[pos:pos + 4 * k * k].view(np.int32).reshape(k, k).copy()
        pos += 4 * k * k
        corr = buf[pos:pos + 4 * k * k].view(np.float32).reshape(k, k).copy()
        pos += 4 * k * k
        sep = None
        if has_sep:
            sep = buf[pos:pos + 4 * k * k * ml].view(np.int32).reshape(k, k, ml).copy()
            pos += 4 * k * k * ml
        return BlockResult(bi, stem, nph, ml, n2o, adj, corr, sep), pos

    def write(self, outdir: str) -> None:
        import os

        base = os.path.join(outdir, self.stem)
        with open(base + ".mdim", "w") as f:
            f.write(f"{self.new_to_old.size}\t{self.num_phen}\t{self.max_level}\n")
        self.new_to_old.astype(np.int32).tofile(base + ".ixs")
        self.adj.astype(np.int32).tofile(base + ".adj")
        self.corr.astype(np.float32).tofile(base + ".corr")
        if self.sep is not None:
            self.sep.astype(np.int32).tofile(base + ".sep")


def gather_results(results: list[BlockResult], device=None, group=None) -> list[BlockResult] | None:
    """The one exchange of the multi-GPU path: every rank's block results -> rank 0 (returns the
    full list, ordered by block index, on rank 0 and None elsewhere)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    payload = np.concatenate([r.pack() for r in results]) if results else np.zeros(0, np.uint8)
    dev = device if device is not None else "cpu"
    size = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    cap = max(1, int(max(int(s.item()) for s in sizes)))
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if payload.size:
        mine[: payload.size] = torch.from_numpy(payload.copy()).to(dev)
    bufs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        buf = bufs[r].cpu().numpy()
        n, pos = int(sizes[r].item()), 0
        while pos < n:
            br, pos = BlockResult.unpack(buf, pos)
            out.append(br)
    return sorted(out, key=lambda b: b.block_index)


# ---------------------------------------------------------------------------
# Row-sharded sweep of ONE block
# ---------------------------------------------------------------------------
def unsigned_min_allreduce_(t, group=None):
    """In-place element-wise UNSIGNED minimum of an int32 / int64 torch tensor across the group.  torch.distributed
    has no unsigned types: flipping the sign bit maps unsigned order onto signed order ("none" = all ones becomes
    the largest signed value), so a signed MIN does it."""
    import torch
    import torch.distributed as dist

    sign = -(1 << (8 * t.element_size() - 1))
    t.bitwise_xor_(sign)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    t.bitwise_xor_(sign)
    return t


def make_min_exchange(group=None, device: int | None = None):
    """The exchange function a row-sharded engine needs (Engine.set_row_shard): host-staged buffers go through
    the group's CPU backend (gloo); device buffers are copied into a torch tensor on `device` and reduced by the
    group's device backend (RCCL), then copied back."""
    import ctypes as C

    import torch

    hip = None

    def exchange(level, buf, count, elem_bytes, on_device, stream):
        nonlocal hip
        if count == 0:
            return 0
        np_dtype, t_dtype = (np.int32, torch.int32) if elem_bytes == 4 else (np.int64, torch.int64)
        if not on_device:
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int32 if elem_bytes == 4 else C.c_int64)), shape=(count,))
            unsigned_min_allreduce_(torch.from_numpy(arr), group)
            return 0
        if hip is None:
            hip = C.CDLL("libamdhip64.so")
        t = torch.empty(count, dtype=t_dtype, device=torch.device("cuda", device if device is not None else torch.cuda.current_device()))
        nbytes = count * elem_bytes
        if hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(buf), C.c_size_t(nbytes), 3) != 0:
            return 1
        unsigned_min_allreduce_(t, group)
        torch.cuda.synchronize()
        return 0 if hip.hipMemcpy(C.c_void_p(buf), C.c_void_p(t.data_ptr()), C.c_size_t(nbytes), 3) == 0 else 1

    return exchange
