"""Host side of `ci-gwas sepselect` / `orient-v-structs` on the MI355X.

Mirrors the reference's interface for this path -- cusk_postprocessing/sepselect.py: class
`MergedCuskResults` (:427-568) and the entry points `sepselect_merged` / `orient_v_structures_merged`
(:571-590) -- with the same attribute names, argument meaning, output files and error behaviour.  The hot
loop of the reference, `find_maximal_and_min_pcorr_sepsets_incr` (:262-329: one np.linalg.inv per candidate per
round per outer pair), runs as ONE batched launch of the HIP kernel behind `cusk_sepselect_greedy`
(csrc/sepselect.hip); there is no CPU fallback: without the HIP library this module raises.

Kept on purpose, because the output files depend on them (the golden files of tests/golden/sepselect_kat.json
were written by the reference itself):
  * variables are laid out traits first (0 .. num_phen-1), then markers; `is_marker(v)` is `v > num_phen` as at
    :535-536, so the first marker counts as a trait when triples are filtered;
  * rows of `.ut` / `.atr` come out in the iteration order of a Python set of (x, y, z) tuples filled in the
    reference's insertion sequence (:140-154);
  * the "minimal partial correlation" set recorded for a pair is the list that keeps growing afterwards
    (:292-294, :308), i.e. it equals the pair's maximal set; only WHETHER a minimum was passed matters for
    `mark_ambiguous_triples` (:98-110);
  * a candidate ties another one at the minimum -> the later one in set-iteration order wins (`<=`, :286).
Deliberate deviations: dense num_var x num_var x max_level sepset arrays are not materialised (dicts are kept,
`maximal_sepset_arr` builds the dense form on request), and `to_file` after plain `sepselect_merged` (no PAG)
writes every other file and skips `_spm.mtx`, where the reference's writer stops with an exception (:550).
"""
from __future__ import annotations

import sys

import numpy as np

from .skeleton import Engine


def alpha_thr(alpha: float, n: int, l):
    """sepselect.py:25-26"""
    from scipy.stats import norm

    return norm.ppf(1 - (alpha / 2)) / np.sqrt(n - l - 3)


class MergedCuskResults:
    """A merged cusk skeleton (`<stem>.mdim`, `.ixs`, `_sam.mtx`, `_scm.mtx`) and what sepselect derives from it."""

    def __init__(self, stem: str, orientation_prior_file=None, device: int = 0, engine: Engine | None = None):
        from scipy.io import mmread

        with open(f"{stem}.mdim") as fin:
            self.num_var, self.num_phen, self.max_level = (int(e) for e in next(fin).split())
        self.num_m = self.num_var - self.num_phen
        self.ixs = np.fromfile(f"{stem}.ixs", dtype=np.int32)
        self.adj = mmread(f"{stem}_sam.mtx").toarray().astype(bool)
        self.corr = np.asarray(mmread(f"{stem}_scm.mtx").toarray(), dtype=np.float64)
        np.fill_diagonal(self.corr, 1.0)
        self.max_sepsets = None
        self.min_sepsets = None
        self.max_level_maximal_sepsets = None
        self.max_level_minimal_pcorr_sepsets = None
        self.unshielded_triples = None
        self.ambiguous_triples = None
        self.rfci_relevant_unshielded_triples = None
        self.pag = None
        self.kernel_ms = 0.0
        self._engine = engine
        self._device = device
        self.rm_collinear_markers()
        p = self.num_phen
        self.orientation_prior = np.zeros(self.adj.shape, dtype=np.int32)
        self.orientation_prior[p:, :p] = self.adj[p:, :p]  # marker -> trait
        if orientation_prior_file is not None:
            given = np.fromfile(orientation_prior_file, dtype=np.int32)
            assert given.shape[0] == p * p, "orientation prior has to have n_trait * n_trait entries"
            self.orientation_prior[:p, :p] = given.reshape(p, p)

    # ---- graph helpers (same names as the reference) ----
    def rm_collinear_markers(self):
        """:467-480 -- drop, one at a time, every marker whose correlation row holds a second exact 1"""
        p = self.num_phen
        keep = np.ones(self.num_var, dtype=bool)
        ones = self.corr == 1
        count = ones.sum(axis=1)
        # removing variable k lowers the count of every row that had a 1 in column k; process in index order
        for k in range(p, self.num_var):
            if count[k] > 1:
                keep[k] = False
                count -= ones[:, k]
        n_rm = int((~keep).sum())
        if n_rm:
            self.corr = np.ascontiguousarray(self.corr[np.ix_(keep, keep)])
            self.adj = np.ascontiguousarray(self.adj[np.ix_(keep, keep)])
            self.ixs = self.ixs[keep[p:]]
            self.num_var -= n_rm
        print(f"Removed {n_rm} collinear markers")

    def neighbors(self, v: int):
        return np.flatnonzero(self.adj[v])

    def trait_neighbors(self, v: int):
        nb = self.neighbors(v)
        return nb[nb < self.num_phen]

    def non_neighbors(self, v: int):
        return np.flatnonzero(~self.adj[v])

    def adjacent(self, a: int, b: int) -> bool:
        return bool(self.adj[a, b] or self.adj[b, a])

    def is_marker(self, v: int) -> bool:
        return v > self.num_phen

    def get_unshielded_triples(self):
        """:140-154 -- the set of (outer, middle, outer), filled in the reference's insertion sequence"""
        if self.unshielded_triples is None:
            linked = self.adj | self.adj.T
            nb = [np.flatnonzero(row) for row in self.adj]
            found = set()
            for a in range(self.num_var):
                na = nb[a]
                for b in na.tolist():
                    cs = na[~linked[b, na]]
                    found.update((b, a, c) for c in cs.tolist() if c != b)
                    cs = nb[b][~linked[a, nb[b]]]
                    found.update((a, b, c) for c in cs.tolist() if c != a)
            self.unshielded_triples = found
        return self.unshielded_triples

    def get_unshielded_triples_outer_pairs(self):
        return {(t[0], t[2]) for t in self.get_unshielded_triples()}

    def get_rfci_relevant_unshielded_triples(self):
        """:70-85 -- middle is a trait, at most one marker, x < z; rows in set-iteration order"""
        if self.rfci_relevant_unshielded_triples is None:
            p = self.num_phen
            rows = [t for t in self.get_unshielded_triples()
                    if t[1] <= p and t[0] < t[2] and (t[0] > p) + (t[1] > p) + (t[2] > p) < 2]
            self.rfci_relevant_unshielded_triples = np.array(rows, dtype=np.int32).reshape(len(rows), 3)
        return self.rfci_relevant_unshielded_triples

    def get_rfci_relevant_unshielded_triples_outer_pairs(self):
        pairs = set()
        for x, _y, z in self.get_rfci_relevant_unshielded_triples().tolist():
            pairs.add((x, z))
            pairs.add((z, x))
        return pairs

    # ---- the hot loop: one batched device launch ----
    def find_maximal_and_min_pcorr_sepsets_incr(self, alpha: float, num_samples: int):
        """:262-329 for every sRFCI-relevant outer pair, on the device"""
        p = self.num_phen
        if not np.array_equal(self.corr[:, :p], self.corr[:p, :].T):
            raise ValueError("sepselect: the correlation matrix is not symmetric on its trait rows / columns")
        pairs = sorted(self.get_rfci_relevant_unshielded_triples_outer_pairs())
        if not pairs:
            raise ValueError("max() arg is an empty sequence")  # what :316 raises on an empty dict
        pair_i = np.array([a for a, _ in pairs], dtype=np.int32)
        pair_j = np.array([b for _, b in pairs], dtype=np.int32)
        order = {}  # candidates of i in the order the reference's `for neighbor in remaining_neighbors` sees them
        for i in np.unique(pair_i).tolist():
            order[i] = np.array(list(set(self.trait_neighbors(i))), dtype=np.int32).reshape(-1)
        sizes = np.array([order[i].shape[0] for i in pair_i.tolist()], dtype=np.int64)
        cand_off = np.zeros(len(pairs) + 1, dtype=np.int64)
        np.cumsum(sizes, out=cand_off[1:])
        cand = np.concatenate([order[i] for i in pair_i.tolist()]) if len(pairs) else np.zeros(0, np.int32)
        thr = alpha_thr(alpha, num_samples, np.arange(int(sizes.max()) + 1 if len(sizes) else 1, dtype=np.float64))
        if self._engine is None:
            self._engine = Engine(self._device)
        sel, sel_len, flags, self.kernel_ms = self._engine.sepselect_greedy(
            np.ascontiguousarray(self.corr[:, :p]), pair_i, pair_j, self.corr[pair_i, pair_j], cand_off, cand, thr)
        status = flags >> 8
        if np.any(status == 2):
            k = int(np.flatnonzero(status == 2)[0])
            print(f"vars: {[int(pair_i[k]), int(pair_j[k])]} + subset of {order[int(pair_i[k])].tolist()}")
            print("Singular matrix")
            sys.exit()  # the reference's pcorr() exits on LinAlgError (:12-18)
        if np.any(status == 1):
            raise KeyError(None)  # remaining_neighbors.remove(None), :309
        self.max_sepsets, self.min_sepsets = {}, {}
        for k, key in enumerate(pairs):
            chosen = sel[cand_off[k]:cand_off[k] + sel_len[k]].tolist()
            self.max_sepsets[key] = chosen
            if flags[k] & 1:
                self.min_sepsets[key] = chosen  # the same list, as in the reference
        self.max_level_maximal_sepsets = max(len(v) for v in self.max_sepsets.values())
        self.max_level_minimal_pcorr_sepsets = max(len(v) for v in self.min_sepsets.values())  # ValueError if none

    @staticmethod
    def _dense(sets, n, depth):
        arr = np.full((n, n, depth), -1, dtype=np.int32)
        for (i, j), v in sets.items():
            arr[i, j, :len(v)] = v
        return arr

    @property
    def maximal_sepset_arr(self):
        if self.max_sepsets is None:
            return None
        return self._dense(self.max_sepsets, self.num_var, self.max_level_maximal_sepsets)

    @property
    def minimal_pcorr_sepset_arr(self):
        if self.min_sepsets is None:
            return None
        return self._dense(self.min_sepsets, self.num_var, self.max_level_minimal_pcorr_sepsets)

    def mark_ambiguous_triples(self):
        """:98-110 -- b in the maximal set of (a, c) and not in its minimal-pcorr set (no entry = all -1)"""
        if self.max_sepsets is None or self.min_sepsets is None:
            raise RuntimeError("Cannot mark ambiguous triples without minimal pcorr and maximal sepsets")
        rows = [t for t in self.get_unshielded_triples()
                if t[1] in self.max_sepsets.get((t[0], t[2]), ()) and t[1] not in self.min_sepsets.get((t[0], t[2]), ())]
        self.ambiguous_triples = np.array(rows, dtype=np.int32)

    def orient_v_structures(self, alpha: float, num_samples: int):
        """:482-508 -- PAG marks 1 (adjacent), 2 (arrowhead), 3 (tail)"""
        self.pag = np.zeros(self.adj.shape, dtype=np.int32)
        self.pag[self.adj] = 1
        if self.max_sepsets is None:
            self.find_maximal_and_min_pcorr_sepsets_incr(alpha, num_samples)
        prior = self.orientation_prior
        for x, y, z in self.get_rfci_relevant_unshielded_triples().tolist():
            collider = y not in self.max_sepsets[(x, z)] and y not in self.max_sepsets[(z, x)]
            for u in (x, z):
                if prior[u, y] == 1:
                    self.pag[u, y], self.pag[y, u] = 2, 3
                elif prior[y, u] == 1:
                    self.pag[y, u], self.pag[u, y] = 2, 3
                elif collider:
                    self.pag[u, y] = 2

    # ---- writers ----
    def to_file(self, stem: str):
        """:538-556"""
        from scipy.io import mmwrite
        from scipy.sparse import coo_matrix

        triples = self.get_rfci_relevant_unshielded_triples()
        with open(stem + ".mdim", "w") as fout:
            fout.write(f"{self.num_var}\t{self.num_phen}\t{self.max_level_maximal_sepsets}\t"
                       f"{self.ambiguous_triples.shape[0]}\t{triples.shape[0]}\n")
        mmwrite(f"{stem}_sam.mtx", coo_matrix(self.adj.astype(np.int32)))
        mmwrite(f"{stem}_scm.mtx", coo_matrix(self.corr))
        if self.pag is not None:
            mmwrite(f"{stem}_spm.mtx", coo_matrix(self.pag))
        else:
            print("sepselect: no PAG was oriented, _spm.mtx not written", file=sys.stderr)
        self.ambiguous_triples.tofile(f"{stem}.atr")
        triples.tofile(f"{stem}.ut")
        self.max_sepset_to_file(stem)

    def max_sepset_to_file(self, stem: str):
        """:558-568 -- one line per ordered pair with a non-empty maximal set, 1-based, row-major pair order"""
        with open(f"{stem}.ssm", "w") as fout:
            for (i, j) in sorted(self.max_sepsets):
                s = self.max_sepsets[(i, j)]
                if s:
                    fout.write(" ".join(str(v + 1) for v in [i, j] + s) + "\n")


def orient_v_structures_merged(cusk1_result_stem: str, alpha: float, num_samples: int, orientation_prior_file=None,
                               device: int = 0) -> MergedCuskResults:
    """:571-578"""
    cr = MergedCuskResults(cusk1_result_stem, orientation_prior_file=orientation_prior_file, device=device)
    print("Orienting v-structures")
    cr.orient_v_structures(alpha=alpha, num_samples=num_samples)
    cr.mark_ambiguous_triples()
    return cr


def sepselect_merged(cusk1_result_stem: str, alpha: float, num_samples: int, device: int = 0) -> MergedCuskResults:
    """:581-588"""
    cr = MergedCuskResults(cusk1_result_stem, device=device)
    print("Starting sepselect")
    cr.find_maximal_and_min_pcorr_sepsets_incr(alpha, num_samples)
    cr.mark_ambiguous_triples()
    return cr
