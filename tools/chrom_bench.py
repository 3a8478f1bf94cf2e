"""Whole-chromosome run on ONE GPU through the block driver (C4 per-GPU share): phase breakdown and the effect of
blocks in flight.  usage: python tools/chrom_bench.py [nblocks] [max_level] [max_level_two]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
import cigwas_amd  # noqa
from cigwas_amd import synth, run_blocks as rb

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 25
L1 = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L2 = int(sys.argv[3]) if len(sys.argv) > 3 else 14
N, p = 16384, 20
d = tempfile.mkdtemp(prefix="chrom_", dir="/tmp")
t0 = time.time()
sizes = synth.chromosome_block_sizes(nb)
G, contrib = synth.chromosome_segment(sizes, 0, nb, N, p)
Y = synth.chromosome_traits(contrib)
means, stds = synth.bed_stats(G)
stem = os.path.join(d, "chr")
synth.write_bfiles(stem, synth.pack_bed(G), N, means, stds)
synth.write_phen_fast(os.path.join(d, "y.phen"), Y)
synth.write_blocks_file(os.path.join(d, "c.blocks"), sizes)
print(f"{nb} blocks, {sum(sizes)} markers, sizes {sizes}; generated + written in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
bs = rb.BlockSet(os.path.join(d, "y.phen"), stem, os.path.join(d, "c.blocks"), 1e-4, L1, L2, 1)
print(f"open {time.time() - t0:.2f} s", flush=True)
for rep, K in enumerate((1, 1, 2, 4, 8)):
    out = os.path.join(d, f"out{rep}"); os.makedirs(out)
    t0 = time.perf_counter()
    allr, stats, owned = rb.run_job(bs, out, 0, inflight=K)
    dt = time.perf_counter() - t0
    tests = sum(int(s.tests[0]) + int(s.tests[1]) for s in stats.values())
    ph = {k: sum(getattr(s, k) for s in stats.values()) for k in ("ms_inputs", "ms_corr", "ms_stage1", "ms_prune", "ms_stage2", "ms_reduce")}
    print(f"inflight {K}: {dt * 1e3:.1f} ms, {nb / dt:.1f} blocks/s, {tests / dt:.3e} tests/s, skipped {sum(s.skipped for s in stats.values())}, "
          + ", ".join(f"{k[3:]} {v:.1f}" for k, v in ph.items()), flush=True)
    if rep == 1:
        for b in owned:
            s = stats[b]
            print(f"  block {b}: m={s.markers} kept={s.retained} t1={s.tests[0]} t2={s.tests[1]} corr {s.ms_corr:.2f} s1 {s.ms_stage1:.2f} prune {s.ms_prune:.2f} "
                  f"s2 {s.ms_stage2:.2f} (levels {s.stage[1].levels_run}, maxdeg {list(s.stage[1].max_degree)[:s.stage[1].levels_run]}) red {s.ms_reduce:.2f}")
