"""Whole-chromosome run on ONE GPU (C4 per-GPU share): per-block execution against batched execution
(cusk_blockset_run_batch), phase breakdown per batch.  usage: python tools/chrom_batch.py [nblocks] [max_level] [max_level_two] [batch_vars,...]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
import cigwas_amd  # noqa
from cigwas_amd import synth, run_blocks as rb

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 25
L1 = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L2 = int(sys.argv[3]) if len(sys.argv) > 3 else 14
BV = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 4096, 8192, 16384, 32768]
INFL = int(sys.argv[5]) if len(sys.argv) > 5 else 1
N, p = 16384, 20
d = tempfile.mkdtemp(prefix="chrom_", dir="/tmp")
sizes = synth.chromosome_block_sizes(nb)
G, contrib = synth.chromosome_segment(sizes, 0, nb, N, p)
Y = synth.chromosome_traits(contrib)
means, stds = synth.bed_stats(G)
stem = os.path.join(d, "chr")
synth.write_bfiles(stem, synth.pack_bed(G), N, means, stds)
synth.write_phen_fast(os.path.join(d, "y.phen"), Y)
synth.write_blocks_file(os.path.join(d, "c.blocks"), sizes)
print(f"{nb} blocks, {sum(sizes)} markers, sizes {sizes}", flush=True)
bs = rb.BlockSet(os.path.join(d, "y.phen"), stem, os.path.join(d, "c.blocks"), 1e-4, L1, L2, 1)
k = 0
for bv in BV:
    for writer in ("local", "rank0"):
        for rep in range(3):
            out = os.path.join(d, f"out{k}"); os.makedirs(out); k += 1
            tm = {}
            t0 = time.perf_counter()
            done, stats, owned = rb.run_job(bs, out, 0, writer=writer, batch_vars=bv, options={"timing": 0}, timings=tm, inflight=INFL)
            dt = time.perf_counter() - t0
        if bv:
            ph = {q: sum(getattr(s, q) for s in stats) for q in ("ms_corr", "ms_stage1", "ms_prune", "ms_stage2", "ms_reduce")}
            canon = sum(int(s.canonical[0]) + int(s.canonical[1]) for s in stats)
            print(f"batch_vars {bv} writer {writer}: {dt * 1e3:.2f} ms, {nb / dt:.0f} blocks/s, {len(stats)} batches, vars {[int(s.vars_stage1) for s in stats]} / {[int(s.vars_stage2) for s in stats]}, "
                  f"canonical {canon / dt:.3e} tests/s, " + ", ".join(f"{q[3:]} {v:.2f}" for q, v in ph.items()) + f", {tm}", flush=True)
            if writer == "local" and bv == BV[-1]:
                for s in stats:
                    for st in (0, 1):
                        S = s.stage[st]
                        print(f"   stage {st + 1}: levels {S.levels_run} total_ms {S.total_ms:.3f} maxdeg {list(S.max_degree)[:S.levels_run]} tests {list(S.tests)[:S.levels_run]}")
        else:
            ph = {q: sum(getattr(s, q) for s in stats.values()) for q in ("ms_corr", "ms_stage1", "ms_prune", "ms_stage2", "ms_reduce")}
            print(f"per block, writer {writer}: {dt * 1e3:.2f} ms, {nb / dt:.0f} blocks/s, " + ", ".join(f"{q[3:]} {v:.2f}" for q, v in ph.items()), flush=True)
bs.close()
