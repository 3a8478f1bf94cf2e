import os, sys, tempfile, time, shutil
import numpy as np
sys.path.insert(0, ".")
import cigwas_amd  # noqa
from cigwas_amd import synth, run_blocks as rb
from cigwas_amd.skeleton import Engine
nb, N, p = 25, 16384, 20
d = tempfile.mkdtemp(prefix="chrom_", dir="/tmp")
sizes = synth.chromosome_block_sizes(nb)
G, contrib = synth.chromosome_segment(sizes, 0, nb, N, p)
Y = synth.chromosome_traits(contrib)
means, stds = synth.bed_stats(G)
stem = os.path.join(d, "chr")
synth.write_bfiles(stem, synth.pack_bed(G), N, means, stds)
synth.write_phen_fast(os.path.join(d, "y.phen"), Y)
synth.write_blocks_file(os.path.join(d, "c.blocks"), sizes)
bs = rb.BlockSet(os.path.join(d, "y.phen"), stem, os.path.join(d, "c.blocks"), 1e-4, 5, 14, 1)
e = Engine(0)
br, st = bs.run_batch(e, list(range(nb)))
buf = br.pack()
print("blocks", br.count, "packed MB", buf.size / 1e6, "retained", st.retained)
for base in ("/tmp", "/dev/shm", os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"):
    for rep in range(3):
        o = tempfile.mkdtemp(prefix="w_", dir=base)
        t0 = time.perf_counter(); br.write(o); t1 = time.perf_counter()
        o2 = tempfile.mkdtemp(prefix="w_", dir=base)
        t2 = time.perf_counter(); rb.write_packed(buf, o2); t3 = time.perf_counter()
        print(f"{base}: batch_result_write {1e3*(t1-t0):.2f} ms, packed write {1e3*(t3-t2):.2f} ms")
        shutil.rmtree(o); shutil.rmtree(o2)
os.system("df /tmp /dev/shm | cat; mount | grep -E ' /tmp | / ' | cat")
