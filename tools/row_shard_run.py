#!/usr/bin/env python3
"""Row-sharded sweep of ONE block over several engines: run, check, time.

Every rank builds the same synthetic correlation matrix, holds it whole in HBM and sweeps the rows
X % world == rank; after each level the ranks join their per-edge selection state with an unsigned-MIN
all-reduce (ci-gwas_amd/shard.py: make_min_exchange).  Rank 0 also runs the block on a single unsharded engine
and compares: adjacency, level counter and every separating-set record must be bit-identical.

  one GPU, W ranks sharing it, CPU collectives (what the tests use; timing is meaningless here, the ranks
  time-slice one device):
      python tools/row_shard_run.py --world 2
  one rank per GPU over RCCL (a multi-GPU node):
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
          tools/row_shard_run.py --backend nccl --markers 50000

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # see ci-gwas_amd/_lib.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist

    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1) if args.backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    import cigwas_amd as cg
    from cigwas_amd import shard, synth

    m, p, N = args.markers, args.traits, args.individuals
    Cm = synth.synth_corr_block(m, p, N=N, block_index=args.seed)
    n = Cm.shape[0]
    Th = cg.threshold_array(N, args.alpha)
    Cd = cg.DeviceArray(Cm)
    eng = cg.Engine(dev)
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    t_ex = [0.0]
    inner = shard.make_min_exchange(device=dev)

    def exchange(level, buf, count, elem_bytes, on_device, stream):
        t0 = time.perf_counter()
        rc = inner(level, buf, count, elem_bytes, on_device, stream)
        t_ex[0] += time.perf_counter() - t0
        return rc

    eng.set_row_shard(rank, world, exchange, host_staging=(args.backend != "nccl"))
    # the hetcor engine (cuskss): uniform effective sample size, or per-pair sizes on the trait rows + a time index
    th_het = cg.hetcor_threshold(args.alpha)
    ti = np.zeros(n, np.int32)
    ti[m:] = 1 + (np.arange(p) % 3)
    Nh, Nd = None, None
    if args.engine == "het":
        rng = np.random.default_rng(args.seed)
        Nh = np.full((n, n), float(N), np.float32)
        ess = (rng.uniform(0.5, 1.0, (n, p)) * N).astype(np.float32)
        Nh[:, m:] = ess
        Nh[m:, :] = ess.T
        Nh[m:, m:] = np.maximum(Nh[m:, m:], Nh[m:, m:].T)
        Nd = cg.DeviceArray(Nh)

    def run(e_):
        if args.engine == "skeleton":
            return e_.run_skeleton(Cd.ptr, n, Th, args.max_level)
        if args.engine == "het":
            return e_.run_hetcor(Cd.ptr, n, th_het, args.max_level, N_dev=Nd.ptr, time_index=ti)
        return e_.run_hetcor(Cd.ptr, n, th_het, args.max_level, ess_uniform=float(N), time_index=ti)

    run(eng)  # warm
    dist.barrier()
    t_ex[0] = 0.0
    t0 = time.perf_counter()
    st = run(eng)
    t_sharded = time.perf_counter() - t0
    G = eng.adjacency()
    rec = eng.sepsets() if args.engine == "skeleton" else None
    tests = torch.tensor([float(sum(st.tests))], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    dist.all_reduce(tests)
    out = None
    if rank == 0:
        one = cg.Engine(dev)
        for kv in args.option:
            k, v = kv.split("=")
            one.set_option(k, int(v))
        run(one)
        t0 = time.perf_counter()
        st1 = run(one)
        t_single = time.perf_counter() - t0

        def canon(r):
            x, y, lv, z, S = r
            order = np.lexsort((y, x))
            return x[order], y[order], lv[order], z[order], S[order]

        same = bool(np.array_equal(G, one.adjacency()) and st.level == st1.level)
        if args.engine == "skeleton":
            a, b = canon(rec), canon(one.sepsets())
            same = same and all(np.array_equal(u, v) for u, v in zip(a, b))
        out = {"workload": f"{args.engine}: {m} SNPs x {p} traits, N={N}, l<={args.max_level}, one block on {world} engines",
               "backend": args.backend, "world": world, "identical_to_single_engine": same, "level": st.level,
               "records": int(len(rec[0])) if rec is not None else 0, "tests_all_ranks": float(tests.item()),
               "tests_single": float(sum(st1.tests)), "sharded_s": t_sharded, "exchange_s_rank0": t_ex[0],
               "single_engine_s": t_single, "edges_level1": int(st.edges[1]), "edges_final": int(G.sum() // 2)}
        if args.oracle:  # the sharded result against the CPU oracle (not only against another engine)
            from oracle import oracle as O

            if args.engine == "skeleton":
                ref = O.skeleton(Cm, O.threshold_array(N, args.alpha), args.max_level)
                x, y, lv, z, S = canon(rec)
                rx, ry = np.nonzero(ref.sepset[:, :, 0] != -1)
                ok = (np.array_equal(G, ref.G) and st.level == ref.level and np.array_equal(rx, x) and np.array_equal(ry, y)
                      and np.array_equal(ref.sepset[rx, ry], S))
            else:
                Nref = Nh if Nh is not None else np.full((n, n), N, np.float32)
                ref = O.hetcor_skeleton(Cm, np.ones((n, n), np.int32), Nref, O.hetcor_threshold(args.alpha), args.max_level, ti)
                ok = np.array_equal(G, ref.G) and st.level == ref.level
            out["identical_to_oracle"] = bool(ok)
            same = same and bool(ok)
            out["identical_to_single_engine"] = bool(out["identical_to_single_engine"])
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if (out is None or (out["identical_to_single_engine"] and out.get("identical_to_oracle", True))) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2, help="ranks to spawn when not started by torchrun")
    ap.add_argument("--backend", choices=["gloo", "nccl"], default="gloo")
    ap.add_argument("--markers", type=int, default=1500)
    ap.add_argument("--traits", type=int, default=10)
    ap.add_argument("--individuals", type=int, default=16384)
    ap.add_argument("--max-level", type=int, default=4)
    ap.add_argument("--alpha", type=float, default=1e-4)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--engine", choices=["skeleton", "hetcor", "het"], default="skeleton",
                    help="skeleton: cusk engine; hetcor: cuskss engine, uniform sample size + time index; het: per-pair sizes")
    ap.add_argument("--oracle", action="store_true", help="also compare the sharded result with the CPU oracle")
    args = ap.parse_args()
    if "RANK" in os.environ:
        sys.exit(worker(args))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = rc or pr.wait()
    sys.exit(rc)


if __name__ == "__main__":
    main()
