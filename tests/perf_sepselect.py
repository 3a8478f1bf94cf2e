#!/usr/bin/env python3
"""Timing of the sepselect path on a synthetic merged skeleton: the device kernel and the whole
`orient_v_structures_merged` call, beside the numpy oracle's greedy loop (the reference's algorithm: one matrix
inverse per candidate per round) on a bounded sample of the same outer pairs.

Lives under tests/ because it runs the oracle (test infrastructure) as the CPU side of the comparison; pytest does
not collect it.

Usage: python tests/perf_sepselect.py [--traits 40] [--markers 4000] [--sample-pairs 300]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--traits", type=int, default=40)
    ap.add_argument("--markers", type=int, default=4000)
    ap.add_argument("--alpha", type=float, default=1e-4)
    ap.add_argument("--num-samples", type=int, default=458747)
    ap.add_argument("--sample-pairs", type=int, default=300)
    ap.add_argument("--trait-edge-prob", type=float, default=0.5)
    ap.add_argument("--noise", type=float, default=2e-3, help="estimation noise on the trait columns (no exact ties)")
    args = ap.parse_args()
    import scipy.sparse as sp
    from scipy.io import mmwrite

    from cigwas_amd import sepselect as SS
    from cigwas_amd import synth
    from oracle import sepselect_oracle as SO

    p, m = args.traits, args.markers
    adj, corr, ixs, _ = synth.merged_skeleton(2025, p, m, trait_edge_prob=args.trait_edge_prob, noise=args.noise)
    with tempfile.TemporaryDirectory() as d:
        stem = os.path.join(d, "all_merged")
        mmwrite(stem + "_sam.mtx", sp.coo_matrix(adj.astype(np.int32)))
        mmwrite(stem + "_scm.mtx", sp.coo_matrix(corr))
        open(stem + ".mdim", "w").write(f"{p + m}\t{p}\t3\n")
        ixs.tofile(stem + ".ixs")
        t0 = time.perf_counter()
        cr = SS.MergedCuskResults(stem)
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        cr.get_rfci_relevant_unshielded_triples()
        t_triples = time.perf_counter() - t0
        cr.find_maximal_and_min_pcorr_sepsets_incr(args.alpha, args.num_samples)  # warm (engine creation, first launch)
        t0 = time.perf_counter()
        cr.find_maximal_and_min_pcorr_sepsets_incr(args.alpha, args.num_samples)
        t_select = time.perf_counter() - t0
        kernel_ms = cr.kernel_ms
        t0 = time.perf_counter()
        cr.orient_v_structures(args.alpha, args.num_samples)
        cr.mark_ambiguous_triples()
        t_orient = time.perf_counter() - t0
        pairs = sorted(cr.max_sepsets)
        rng = np.random.default_rng(1)
        take = [pairs[k] for k in rng.choice(len(pairs), size=min(args.sample_pairs, len(pairs)), replace=False)]
        g = {"corr": cr.corr, "adj": cr.adj, "num_phen": cr.num_phen}
        t0 = time.perf_counter()
        grown, _ = SO.greedy_sepsets(g, take, args.alpha, args.num_samples)
        t_cpu = time.perf_counter() - t0
        same = all([int(v) for v in grown[k]] == cr.max_sepsets[k] for k in take)
        cand = np.array([len(cr.trait_neighbors(i)) for i, _ in pairs])
        print(json.dumps({
            "workload": f"merged skeleton, {p} traits + {m} markers", "outer_pairs": len(pairs),
            "candidates_mean": float(cand.mean()), "candidates_max": int(cand.max()),
            "max_sepset_len": cr.max_level_maximal_sepsets,
            "device_kernel_ms": kernel_ms, "select_call_s": t_select, "pairs_per_s_device_call": len(pairs) / t_select,
            "load_s": t_load, "triples_s": t_triples, "orient_and_ambiguous_s": t_orient,
            "cpu_oracle_pairs": len(take), "cpu_oracle_s": t_cpu, "pairs_per_s_cpu_oracle": len(take) / t_cpu,
            "cpu_oracle_extrapolated_s": t_cpu / len(take) * len(pairs), "sample_sets_equal": bool(same)}))


if __name__ == "__main__":
    main()
