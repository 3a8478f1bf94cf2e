"""sepselect_oracle.py -- CPU restatement of the reference's `sepselect` post-processing.

TEST INFRASTRUCTURE ONLY: imported by tests/ (parity tests and the timing script tests/perf_sepselect.py) and nothing
in the product path.

What it restates (reference file:line, /root/reference/cusk_postprocessing/sepselect.py):
  * loading a merged skeleton and dropping collinear markers ........ :427-480
  * unshielded triples of the merged graph ........................... :140-160
  * the sRFCI-relevant triples and their outer pairs ................. :70-92, :535-536 (merged `is_marker`)
  * greedy forward selection of separating sets (the hot loop) ....... :262-329, with the partial correlation
    of (i, j | S) read off the inverse of the correlation sub-matrix (:8-18, :162-164) and the Fisher z /
    threshold of :21-31
  * ambiguous triples ................................................ :98-110
  * v-structure orientation with an orientation prior ................ :482-508
  * the writers (.mdim, _sam/_scm/_spm.mtx, .atr, .ut, .ssm) ......... :538-568

Pinned by tests/golden/sepselect_kat.json: inputs and the files the reference itself wrote for them in this
container (tests/golden/make_sepselect_golden.py imports the reference and runs
`orient_v_structures_merged(...).to_file(...)`).

Two behaviours of the reference that this restatement keeps on purpose (the fixtures show both):
  * the list stored as the "minimal partial correlation" set of a pair is the same list object that keeps
    growing afterwards, so it ends up equal to the pair's maximal set (:292-294, :308); what survives of the
    minimum search is only WHETHER the minimum was seen;
  * the merged `is_marker` is `index > num_phen` (:535-536), so the first marker counts as a trait.
Row order of .ut / .atr is the iteration order of a CPython set of int tuples; the restatement builds the same
set with the same insertion sequence, which reproduces that order under the same interpreter.
"""
import numpy as np
from scipy.io import mmread, mmwrite
from scipy.sparse import coo_matrix
from scipy.stats import norm


def fisher_z(r):
    """:21-22"""
    return np.abs(0.5 * np.log(np.abs((1 + r) / (1 - r))))


def z_threshold(alpha, num_samples, level):
    """:25-26"""
    return norm.ppf(1 - (alpha / 2)) / np.sqrt(num_samples - level - 3)


def load_merged(stem):
    """:427-447 -- returns a dict with adj (bool), corr (f64, unit diagonal), ixs, num_var, num_phen"""
    with open(stem + ".mdim") as f:
        num_var, num_phen, _max_level = (int(t) for t in f.readline().split())
    g = {
        "num_var": num_var,
        "num_phen": num_phen,
        "ixs": np.fromfile(stem + ".ixs", dtype=np.int32),
        "adj": mmread(stem + "_sam.mtx").toarray().astype(bool),
        "corr": np.array(mmread(stem + "_scm.mtx").toarray(), dtype=np.float64),
    }
    np.fill_diagonal(g["corr"], 1.0)
    drop_collinear_markers(g)
    return g


def drop_collinear_markers(g):
    """:467-480 -- a marker whose correlation row holds more than one exact 1 goes, one at a time"""
    k = g["num_phen"]
    while k < g["num_var"]:
        if np.count_nonzero(g["corr"][k] == 1) > 1:
            for name in ("corr", "adj"):
                g[name] = np.delete(np.delete(g[name], k, 0), k, 1)
            g["ixs"] = np.delete(g["ixs"], k - g["num_phen"])
            g["num_var"] -= 1
        else:
            k += 1


def unshielded_triples(adj):
    """:140-154 -- set of (outer, middle, outer); same insertion sequence as the reference"""
    nb = [np.flatnonzero(adj[v]) for v in range(adj.shape[0])]
    linked = adj | adj.T
    out = set()
    for mid in range(adj.shape[0]):
        for b in nb[mid]:
            for c in nb[mid]:
                if b != c and not linked[b, c]:
                    out.add((b, mid, c))
            for c in nb[b]:
                if c != mid and not linked[mid, c]:
                    out.add((mid, b, c))
    return out


def relevant_triples(triples, num_phen):
    """:70-85 with the merged is_marker (:535-536)"""
    rows = [[x, y, z] for (x, y, z) in triples
            if not y > num_phen and x < z and sum(int(v > num_phen) for v in (x, y, z)) < 2]
    return np.array(rows, dtype=np.int32)


def outer_pairs(rel):
    """:87-92"""
    s = set()
    for x, _y, z in rel:
        s.add((x, z))
        s.add((z, x))
    return s


def partial_z(corr, variables):
    """:8-18, :162-164 -- |Fisher z| of the partial correlation of the first two variables given the rest"""
    prec = np.linalg.inv(corr[np.ix_(variables, variables)])
    return fisher_z(-(prec[0, 1] / np.sqrt(np.abs(prec[0, 0] * prec[1, 1]))))


def greedy_pair(corr, i, j, pool, alpha, num_samples):
    """:267-311 for one outer pair; `pool` is consumed.  Returns (chosen list, minimum seen?)"""
    chosen = []
    separated = partial_z(corr, [i, j]) < z_threshold(alpha, num_samples, 0)
    seen_minimum = False
    previous = np.inf
    for size in range(1, len(pool) + 1):
        best, pick = np.inf, None
        for t in pool:
            z = partial_z(corr, [i, j] + chosen + [t])
            if z <= best:
                best, pick = z, t
        if best > previous and separated and not seen_minimum:
            seen_minimum = True  # from here on the reference's "minimum" entry is this same, still growing, list
        indep = best < z_threshold(alpha, num_samples, size)
        if separated and not indep:
            break
        separated = separated or indep
        previous = best
        chosen.append(pick)
        pool.remove(pick)
    return chosen, seen_minimum


def greedy_sepsets(g, pairs, alpha, num_samples):
    """:262-329 -- returns {pair: list} of maximal sets and {pair: list} of the pairs that saw a minimum"""
    corr, adj, num_phen = g["corr"], g["adj"], g["num_phen"]
    grown, with_minimum = {}, {}
    for (i, j) in pairs:
        row = np.flatnonzero(adj[i])
        chosen, seen = greedy_pair(corr, i, j, set(row[row < num_phen]), alpha, num_samples)
        grown[(i, j)] = chosen
        if seen:
            with_minimum[(i, j)] = chosen
    return grown, with_minimum


def ambiguous_triples(triples, grown, with_minimum):
    """:98-110 -- b sits in the maximal set of (a, c) and not in its minimum set (absent = all -1)"""
    rows = []
    for a, b, c in triples:
        mx = grown.get((a, c), [])
        mn = with_minimum.get((a, c), [])
        if b in mx and b not in mn:
            rows.append([a, b, c])
    return np.array(rows, dtype=np.int32)


def orient(g, rel, grown, prior):
    """:482-508 -- PAG marks: 1 adjacent, 2 arrowhead at the column variable, 3 tail"""
    pag = np.zeros(g["adj"].shape, dtype=np.int32)
    pag[g["adj"]] = 1
    for x, y, z in rel:
        collider = y not in grown[(x, z)] and y not in grown[(z, x)]
        for u in (x, z):
            if prior[u, y] == 1:
                pag[u, y], pag[y, u] = 2, 3
            elif prior[y, u] == 1:
                pag[y, u], pag[u, y] = 2, 3
            elif collider:
                pag[u, y] = 2
    return pag


def orientation_prior(g, prior_file=None):
    """:450-459"""
    p = g["num_phen"]
    prior = np.zeros(g["adj"].shape, dtype=np.int32)
    prior[p:, :p] = g["adj"][p:, :p]
    if prior_file is not None:
        given = np.fromfile(prior_file, dtype=np.int32)
        assert given.shape[0] == p * p
        prior[:p, :p] = given.reshape(p, p)
    return prior


def run(stem, alpha, num_samples, prior_file=None):
    """orient_v_structures_merged (:571-578): everything the writers need"""
    g = load_merged(stem)
    triples = unshielded_triples(g["adj"])
    rel = relevant_triples(triples, g["num_phen"])
    grown, with_minimum = greedy_sepsets(g, outer_pairs(rel), alpha, num_samples)
    if not grown or not with_minimum:
        raise ValueError("max() arg is an empty sequence")  # :316, :328 on an empty dict
    pag = orient(g, rel, grown, orientation_prior(g, prior_file))
    amb = ambiguous_triples(triples, grown, with_minimum)
    return {"g": g, "triples": triples, "rel": rel, "max_sepsets": grown, "min_sepsets": with_minimum, "pag": pag,
            "ambiguous": amb}


def write(res, stem):
    """:538-568"""
    g = res["g"]
    longest = max(len(v) for v in res["max_sepsets"].values())
    with open(stem + ".mdim", "w") as f:
        f.write(f"{g['num_var']}\t{g['num_phen']}\t{longest}\t{res['ambiguous'].shape[0]}\t{res['rel'].shape[0]}\n")
    mmwrite(stem + "_sam.mtx", coo_matrix(g["adj"].astype(np.int32)))
    mmwrite(stem + "_scm.mtx", coo_matrix(g["corr"]))
    mmwrite(stem + "_spm.mtx", coo_matrix(res["pag"]))
    res["ambiguous"].tofile(stem + ".atr")
    res["rel"].tofile(stem + ".ut")
    with open(stem + ".ssm", "w") as f:
        for (i, j) in sorted(res["max_sepsets"]):
            s = res["max_sepsets"][(i, j)]
            if s:
                f.write(" ".join(str(int(v) + 1) for v in [i, j] + list(s)) + "\n")
