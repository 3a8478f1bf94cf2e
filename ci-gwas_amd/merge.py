"""`merge-block-outputs` and the post-step of `cuskss --marker-indices`: the sparse merged-skeleton files.

Mirror of the reference's cusk_postprocessing/merge_blocks.py (`merge_block_outputs` :361-395,
`reformat_cuskss_merged_output` :398-425, `GlobalBdpcResult.write_mm` :298-325) for the two call sites of its CLI
(ci-gwas.py:452-464), so that the whole-chromosome flow -- per-block `cusk` -> merge -> `cuskss-merged` -> sepselect --
runs from this package alone.  Same files byte for byte: tests/golden/merge/ holds files the reference's own code
wrote (tests/golden/make_merge_golden.py) and tests/test_merge_golden.py compares.

The merged variable space is 1-based and puts the traits first: trait t -> t + 1, the k-th selected marker of the
chromosome -> num_phen + k + 1.  Entry order in the .mtx files is the reference's (Python dict insertion order),
which the merge rules below reproduce:
  * adjacency (`_sam`): trait-trait links survive only while every block has them -- the reference tests the
    0-based pairs (i, j), i, j < num_phen against 1-based keys, so links of the LAST trait are never intersected but
    overwritten by each block like marker links; everything else is taken from the block;
  * correlations (`_scm`): later blocks overwrite (the trait-trait entries are the same in every block);
  * `.ixs`: global (.bim row) index of every selected marker = index inside its block + markers of all blocks before.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

BASE_INDEX = 1


def block_stems(blockfile: str) -> list[str]:
    stems = []
    with open(blockfile) as f:
        for line in f:
            w = line.strip().split("\t")
            stems.append(f"{w[0]}_{w[1]}_{w[2]}")
    return stems


def _block_size(path: str) -> int:
    first, last = path.split("_")[-2:]
    return int(last) - int(first) + 1


def _dense_to_sparse_index(num_m: int, num_p: int, marker_offset: int) -> np.ndarray:
    """block-local variable index (markers first, then traits) -> merged 1-based index"""
    ix = np.arange(num_m + num_p)
    return np.where(ix < num_m, ix + marker_offset + num_p + BASE_INDEX, ix - num_m + BASE_INDEX)


def _sparse_entries(path: str, n: int, to_sparse: np.ndarray, dtype):
    """non-zero entries of a dense n x n file in row-major order: [((i, j), value), ...] in merged indices"""
    dm = np.fromfile(path, dtype=dtype).reshape(n, n)
    r, c = np.nonzero(dm)
    return [((int(i), int(j)), v) for i, j, v in zip(to_sparse[r], to_sparse[c], dm[r, c])]


@dataclass
class MergedSkeleton:
    sam: dict
    scm: dict
    gmi: dict
    num_var: int
    num_phen: int
    max_level: int

    def write_mm(self, basepath: str) -> None:
        dim = max(k[0] for k in self.sam)  # both headers carry the adjacency's largest row index
        with open(basepath + "_sam.mtx", "w") as f:
            f.write("%%MatrixMarket matrix coordinate integer general\n")
            f.write(f"{dim}\t{dim}\t{len(self.sam)}\n")
            f.write("".join(f"{i}\t{j}\t{v}\n" for (i, j), v in self.sam.items()))
        with open(basepath + "_scm.mtx", "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n")
            f.write(f"{dim}\t{dim}\t{len(self.scm)}\n")
            f.write("".join(f"{i}\t{j}\t{v}\n" for (i, j), v in self.scm.items()))
        with open(basepath + ".mdim", "w") as f:
            f.write(f"{self.num_var}\t{self.num_phen}\t{self.max_level}\n")
        np.array(sorted(self.gmi.values()), dtype=np.int32).tofile(basepath + ".ixs")


def merge_block_outputs(blockfile: str, outdir: str) -> MergedSkeleton:
    """outdir is prefixed to the stems as is (the reference's CLI appends the '/')."""
    sam: dict = {}
    scm: dict = {}
    gmi: dict = {}
    marker_offset = 0   # selected markers of all blocks so far
    global_offset = 0   # markers (selected or not) of all blocks so far
    last = None
    for index, path in enumerate(outdir + s for s in block_stems(blockfile)):
        if not os.path.exists(path + ".mdim"):
            print(f"Missing: {path}")  # a block without signal writes no files (cli.cpp:572-576)
            global_offset += _block_size(path)
            continue
        with open(path + ".mdim") as f:
            num_var, num_p, max_level = (int(v) for v in f.readline().strip().split("\t"))
        num_m = num_var - num_p
        to_sparse = _dense_to_sparse_index(num_m, num_p, marker_offset)
        adj = _sparse_entries(path + ".adj", num_var, to_sparse, np.int32)
        cor = _sparse_entries(path + ".corr", num_var, to_sparse, np.float32)
        if index == 0:
            sam.update(adj)  # only the first block listed is taken as it is (a missing first block changes the rules)
        else:
            have = {k for k, _ in adj}
            for i in range(num_p):
                for j in range(num_p):
                    if (i, j) in sam and (i, j) not in have:
                        del sam[(i, j)]
            for (i, j), v in adj:
                if i >= num_p or j >= num_p:
                    sam[(i, j)] = v
        scm.update(cor)
        rel = np.fromfile(path + ".ixs", dtype=np.int32)
        for dm_ix, sm_ix in enumerate(to_sparse):
            if sm_ix >= num_p + BASE_INDEX:
                gmi[int(sm_ix)] = rel[dm_ix] + global_offset
        marker_offset += num_m
        global_offset += _block_size(path)
        last = (num_p, max_level)
    if last is None:
        raise FileNotFoundError("no block output found under " + outdir)
    return MergedSkeleton(sam, scm, gmi, marker_offset + last[0], last[0], last[1])


def reformat_cuskss_merged_output(cusk_dir: str) -> MergedSkeleton:
    """`cuskss --marker-indices` writes dense files over the markers it retained; this turns them into the merged
    sparse form, with the retained markers' global indices looked up in merged_blocks.ixs."""
    with open(f"{cusk_dir}/cuskss_merged.mdim") as f:
        num_var, num_p, max_level = (int(v) for v in next(f).split())
    num_m = num_var - num_p
    old_global = np.fromfile(f"{cusk_dir}/merged_blocks.ixs", dtype=np.int32)
    ixs = np.fromfile(f"{cusk_dir}/cuskss_merged.ixs", dtype=np.int32)
    glob = old_global[ixs[:-num_p]]
    to_sparse = _dense_to_sparse_index(num_m, num_p, 0)
    base = f"{cusk_dir}/cuskss_merged"
    return MergedSkeleton(dict(_sparse_entries(base + ".adj", num_var, to_sparse, np.int32)),
                          dict(_sparse_entries(base + ".corr", num_var, to_sparse, np.float32)),
                          {k: v for k, v in enumerate(glob)}, num_var, num_p, max_level)
