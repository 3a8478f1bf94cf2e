"""Deterministic synthetic inputs for tests and bench.py.

Mirrors the parameters of the reference's simulator
(/root/reference/simulation/simulate_dag.R:106-115: SNP -> trait effects,
trait -> trait DAG, unit noise) on genotypes with realistic block LD so that
level-0 degrees look like an LD-pruned block: windows of `window` SNPs, latent
AR(1) Gaussian (rho) inside a window, two latent copies thresholded at the
MAF and summed to {0,1,2}.  Seeds: numpy Generator(PCG64(20251003 + block_index)).
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20251003


def rng_for(block_index: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(BASE_SEED + int(block_index)))


def make_genotypes(m: int, N: int, rng, window: int = 100, rho: float = 0.85, miss: float = 0.001) -> np.ndarray:
    """m x N int8 dosages in {0,1,2}, -1 = missing."""
    from scipy.stats import norm

    G = np.empty((m, N), np.int8)
    maf = rng.uniform(0.05, 0.5, size=m)
    thr = norm.ppf(1.0 - maf).astype(np.float32)
    s = np.float32(np.sqrt(1.0 - rho * rho))
    for w0 in range(0, m, window):
        w1 = min(m, w0 + window)
        acc = np.zeros((w1 - w0, N), np.int8)
        for _copy in range(2):
            z = rng.standard_normal((w1 - w0, N), dtype=np.float32)
            for j in range(1, w1 - w0):
                z[j] = np.float32(rho) * z[j - 1] + s * z[j]
            acc += (z > thr[w0:w1, None]).astype(np.int8)
        G[w0:w1] = acc
    if miss > 0:
        mask = rng.random((m, N), dtype=np.float32) < miss
        G[mask] = -1
    return G


def make_traits(G: np.ndarray, p: int, rng, causal_per_1000: float = 5.0) -> np.ndarray:
    """p x N float32 standardised traits: sparse SNP effects + lower-triangular trait DAG."""
    m, N = G.shape
    Y = np.zeros((p, N), np.float64)
    ncausal = max(1, int(round(causal_per_1000 * m / 1000.0)))
    for k in range(p):
        idx = rng.choice(m, size=min(ncausal, m), replace=False)
        beta = rng.uniform(0.02, 0.08, size=idx.size) * rng.choice([-1.0, 1.0], size=idx.size)
        g = G[idx].astype(np.float64)
        g[g < 0] = np.nan
        mu = np.nanmean(g, axis=1, keepdims=True)
        sd = np.nanstd(g, axis=1, keepdims=True)
        sd[sd == 0] = 1.0
        gs = np.nan_to_num((g - mu) / sd)
        y = beta @ gs
        for k2 in range(k):
            if rng.random() < 3.0 / p:
                y = y + rng.uniform(0.05, 0.2) * rng.choice([-1.0, 1.0]) * Y[k2]
        y = y + rng.standard_normal(N)
        Y[k] = (y - y.mean()) / y.std()
    return Y.astype(np.float32)


_CODE = np.array([3, 2, 0, 1], np.uint8)  # dosage 0,1,2,missing(-1) -> 2-bit code 11,10,00,01


def pack_bed(G: np.ndarray) -> np.ndarray:
    """m x ceil(N/4) uint8, PLINK SNP-major 2-bit codes, low bits first (no magic bytes).
    Padding samples get code 0 like PLINK's writers."""
    m, N = G.shape
    codes = _CODE[G.astype(np.int64)]  # -1 indexes the last entry (missing)
    pad = (-N) % 4
    if pad:
        codes = np.concatenate([codes, np.zeros((m, pad), np.uint8)], axis=1)
    c = codes.reshape(m, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def bed_stats(G: np.ndarray):
    """per-marker mean and population std over non-missing samples (reference prep.cpp:56-73)."""
    m = G.shape[0]
    mean = np.empty(m, np.float32)
    std = np.empty(m, np.float32)
    for r0 in range(0, m, 512):
        g = G[r0:r0 + 512].astype(np.float64)
        g[g < 0] = np.nan
        mean[r0:r0 + 512] = np.nanmean(g, axis=1)
        std[r0:r0 + 512] = np.nanstd(g, axis=1)
    return mean, std


def synth_bed_block(m: int, N: int, p: int, block_index: int = 0, **kw):
    """(bed bytes m x ceil(N/4), phen column-major p*N float32, means, stds, G)."""
    rng = rng_for(block_index)
    G = make_genotypes(m, N, rng, **kw)
    Y = make_traits(G, p, rng)
    means, stds = bed_stats(G)
    return pack_bed(G), np.ascontiguousarray(Y).reshape(-1), means, stds, G


def corr_from_samples(G: np.ndarray, Y: np.ndarray, device: str | None = None) -> np.ndarray:
    """(m+p) x (m+p) float32 Pearson correlation of mean-imputed dosages and traits."""
    m, N = G.shape
    X = np.concatenate([G.astype(np.float32), Y.astype(np.float32)], axis=0)
    X[:m][G < 0] = np.nan
    mu = np.nanmean(X, axis=1, keepdims=True)
    X = np.nan_to_num(X - mu)
    X /= np.maximum(np.sqrt((X * X).sum(axis=1, keepdims=True)), 1e-30)
    if device is not None:
        import torch

        t = torch.from_numpy(X).to(device)
        Cm = (t @ t.T).float().cpu().numpy()
    else:
        Cm = X @ X.T
    Cm = np.clip(Cm, -1.0, 1.0).astype(np.float32)
    Cm = np.maximum(Cm, Cm.T) * (np.abs(Cm) >= np.abs(Cm.T)) + np.minimum(Cm, Cm.T) * (np.abs(Cm) < np.abs(Cm.T))
    np.fill_diagonal(Cm, 1.0)
    return np.ascontiguousarray(Cm, np.float32)


def synth_corr_block(m: int, p: int, N: int = 16384, block_index: int = 0, device: str | None = None, **kw):
    """Summary-statistics style input for cuskss: symmetric n x n fp32 correlation matrix."""
    rng = rng_for(block_index)
    G = make_genotypes(m, N, rng, **kw)
    Y = make_traits(G, p, rng)
    return corr_from_samples(G, Y, device)


def random_corr(n: int, seed: int, k: int | None = None, strength: float = 1.0) -> np.ndarray:
    """Small dense test matrix: sample correlation of a random sparse linear SEM
    (recipe of /root/reference/cusk/scripts/skeleton_gen.py:19-90, own code)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k = k or 4 * n
    B = np.tril(rng.random((n, n)) < min(1.0, 2.5 / n), -1) * rng.uniform(0.3, 0.9, (n, n)) * strength
    B *= rng.choice([-1.0, 1.0], size=(n, n))
    X = np.zeros((n, k))
    E = rng.standard_normal((n, k))
    for i in range(n):
        X[i] = B[i] @ X + E[i]
    Cm = np.corrcoef(X).astype(np.float32)
    Cm = np.triu(Cm, 1)
    Cm = Cm + Cm.T
    np.fill_diagonal(Cm, 1.0)
    return np.ascontiguousarray(Cm, np.float32)


def write_bfiles(stem: str, bed: np.ndarray, N: int, means, stds, chr_ids=None) -> None:
    """<stem>.bed/.bim/.fam/.dim/.means/.stds as the reference's `mps prep` leaves them
    (reference io.h:30-64, prep.cpp:157-201, bfiles_base.h:8-9)."""
    m = bed.shape[0]
    chr_ids = chr_ids if chr_ids is not None else ["1"] * m
    with open(stem + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(np.ascontiguousarray(bed, np.uint8).tobytes())
    with open(stem + ".bim", "w") as f:
        for i in range(m):
            f.write(f"{chr_ids[i]}\trs{i}\t0\t{1000 + i}\tA\tG\n")
    with open(stem + ".fam", "w") as f:
        for i in range(N):
            f.write(f"f{i} i{i} 0 0 0 -9\n")
    with open(stem + ".dim", "w") as f:
        f.write(f"{N}\t{m}\n")
    for sfx, arr in ((".means", means), (".stds", stds)):
        with open(stem + sfx, "w") as f:
            for v in np.asarray(arr, np.float32):
                f.write(repr(float(v)) + "\n")


def write_phen(path: str, phen_colmajor: np.ndarray, N: int, p: int) -> None:
    """header + `FID IID v1..vp` rows, NaN written as NA (reference phen.cpp:9-74)."""
    Y = np.asarray(phen_colmajor, np.float32).reshape(p, N)
    with open(path, "w") as f:
        f.write("FID IID " + " ".join(f"T{k}" for k in range(p)) + "\n")
        for i in range(N):
            f.write(f"f{i} i{i} " + " ".join("NA" if np.isnan(Y[k, i]) else repr(float(Y[k, i])) for k in range(p)) + "\n")


def synth_corr_block_torch(m: int, p: int, N: int = 16384, block_index: int = 0, device: str = "cuda",
                           window: int = 100, rho: float = 0.85):
    """Full-size variant of synth_corr_block generated on the GPU with torch (plumbing only): same
    generative model (LD windows with latent AR(1), MAF ~ U(0.05, 0.5), sparse SNP effects + trait DAG),
    torch's generator instead of numpy's.  Returns a CUDA float32 tensor (m+p) x (m+p), symmetric, unit diagonal."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(BASE_SEED + int(block_index))
    n = m + p
    X = torch.empty((n, N), dtype=torch.float32, device=device)
    s = (1.0 - rho * rho) ** 0.5
    nrm = torch.distributions.Normal(0.0, 1.0)
    for w0 in range(0, m, window):
        w1 = min(m, w0 + window)
        maf = torch.rand(w1 - w0, generator=g, device=device) * 0.45 + 0.05
        thr = nrm.icdf(1.0 - maf).unsqueeze(1)
        acc = torch.zeros((w1 - w0, N), dtype=torch.float32, device=device)
        for _copy in range(2):
            z = torch.randn((w1 - w0, N), generator=g, device=device)
            for j in range(1, w1 - w0):
                z[j] = rho * z[j - 1] + s * z[j]
            acc += (z > thr).float()
        X[w0:w1] = acc
    Xs = X[:m] - X[:m].mean(1, keepdim=True)
    Xs = Xs / Xs.std(1, keepdim=True).clamp_min(1e-6)
    ncausal = max(1, int(round(5.0 * m / 1000.0)))
    for k in range(p):
        idx = torch.randint(0, m, (ncausal,), generator=g, device=device)
        beta = (torch.rand(ncausal, generator=g, device=device) * 0.06 + 0.02) * (
            torch.randint(0, 2, (ncausal,), generator=g, device=device).float() * 2 - 1)
        y = beta @ Xs[idx]
        for k2 in range(k):
            if float(torch.rand(1, generator=g, device=device)) < 3.0 / p:
                y = y + (0.05 + 0.15 * float(torch.rand(1, generator=g, device=device))) * X[m + k2]
        y = y + torch.randn(N, generator=g, device=device)
        X[m + k] = (y - y.mean()) / y.std()
    del Xs
    X -= X.mean(1, keepdim=True)
    X /= X.norm(dim=1, keepdim=True).clamp_min(1e-30)
    Cm = X @ X.T
    del X
    Cm.clamp_(-1.0, 1.0)
    Cm = torch.triu(Cm, 1)
    Cm = Cm + Cm.T
    Cm.fill_diagonal_(1.0)
    return Cm


def hub_corr(nleaf: int, nhub: int, seed: int, noise: float = 1.0) -> np.ndarray:
    """Population correlation matrix of `nleaf` nearly independent variables and `nhub` hubs, each a random signed
    combination of ALL leaves plus noise.  A hub stays adjacent to every leaf whatever is conditioned on (the edges are
    real), so the skeleton keeps a degree of nleaf + nhub - 1 to the deepest level: the matrix that drives the sweep
    through levels 9..14 in the tests (level l of a hub row enumerates C(nleaf + nhub - 1, l) conditioning sets)."""
    rng = np.random.default_rng(seed)
    n = nleaf + nhub
    A = np.eye(nleaf) + 0.02 * rng.normal(size=(nleaf, nleaf))
    SL = A @ A.T
    W = rng.uniform(0.5, 1.0, size=(nhub, nleaf)) * rng.choice([-1.0, 1.0], size=(nhub, nleaf))
    S = np.zeros((n, n))
    S[:nleaf, :nleaf] = SL
    S[nleaf:, :nleaf] = W @ SL
    S[:nleaf, nleaf:] = S[nleaf:, :nleaf].T
    S[nleaf:, nleaf:] = W @ SL @ W.T + noise * np.eye(nhub)
    d = np.sqrt(np.diag(S))
    C = (S / d[:, None] / d[None, :]).astype(np.float32)
    C = np.minimum(C, C.T)
    np.fill_diagonal(C, 1.0)
    return np.ascontiguousarray(C)


def merged_skeleton(seed: int, p: int, m: int, duplicates: int = 0, with_prior: bool = False, block: int = 5,
                    trait_edge_prob: float = 0.35, noise: float = 0.0):
    """A merged cusk skeleton in the layout `merge-block-outputs` writes (traits 0..p-1, then m selected markers):
    the population correlation matrix of a linear model -- markers in LD blocks of `block`, each trait driven by
    a few markers, a random lower-triangular trait DAG -- and an adjacency read off it.  Marker-marker entries
    outside a block are dropped (zero), as in a merged `_scm.mtx`; the trait rows and columns are complete, so
    every sub-matrix the separation-set search inverts is positive definite.  `duplicates` markers are exact
    copies of their predecessor (what `rm_collinear_markers` is there for).  `noise` > 0 adds symmetric Gaussian
    estimation noise of that scale to the stored off-diagonal entries (population values carry structural exact
    zeros and with them exact ties between candidates, which sample correlations do not have).
    Returns (adj bool n x n, corr f64 n x n with unit diagonal, ixs i32 m, prior i32 p x p or None)."""
    rng = np.random.default_rng(seed)
    n = p + m
    blk = np.arange(m) // block
    Sm = np.where(blk[:, None] == blk[None, :], 0.6 ** np.abs(np.arange(m)[:, None] - np.arange(m)[None, :]), 0.0)
    B = np.zeros((p, m))
    for t in range(p):
        k = rng.integers(2, 5)
        cols = rng.choice(m, size=min(k, m), replace=False)
        B[t, cols] = rng.uniform(0.12, 0.35, size=cols.size) * rng.choice([-1.0, 1.0], size=cols.size)
    A = np.tril(rng.uniform(0.15, 0.45, size=(p, p)) * rng.choice([-1.0, 1.0], size=(p, p)), -1)
    A *= np.tril(rng.random((p, p)) < trait_edge_prob, -1)
    T = np.linalg.inv(np.eye(p) - A)
    # variables [traits, markers]:  y = T (B x + e),  x ~ (0, Sm),  e ~ (0, I)
    cov = np.zeros((n, n))
    cov[p:, p:] = Sm
    cov[:p, p:] = T @ B @ Sm
    cov[p:, :p] = cov[:p, p:].T
    cov[:p, :p] = T @ (B @ Sm @ B.T + np.eye(p)) @ T.T
    sd = np.sqrt(np.diag(cov))
    corr = cov / np.outer(sd, sd)
    corr = 0.5 * (corr + corr.T)
    dup = rng.choice(np.arange(1, m), size=duplicates, replace=False) if duplicates else []
    for b in dup:  # marker b becomes an exact copy of marker b - 1
        a = p + b - 1
        b = p + b
        corr[b, :] = corr[a, :]
        corr[:, b] = corr[:, a]
        corr[a, b] = corr[b, a] = 1.0
    np.fill_diagonal(corr, 1.0)
    same_block = np.zeros((n, n), dtype=bool)
    same_block[p:, p:] = blk[:, None] == blk[None, :]
    adj = np.abs(corr) > 0.08
    adj[p:, p:] &= same_block[p:, p:] & (np.abs(corr[p:, p:]) > 0.2)
    np.fill_diagonal(adj, False)
    keep = np.ones((n, n), dtype=bool)
    keep[p:, p:] = same_block[p:, p:]
    corr = np.where(keep, corr, 0.0)
    if noise > 0.0:
        g = np.random.default_rng(seed + 7919).normal(scale=noise, size=(n, p))
        e = np.zeros((n, n))
        e[:, :p] = g
        e[:p, :] = g.T
        e[:p, :p] = 0.5 * (g[:p] + g[:p].T)
        np.fill_diagonal(e, 0.0)
        corr = corr + e
    ixs = np.sort(rng.choice(50 * m, size=m, replace=False)).astype(np.int32)
    prior = None
    if with_prior:
        prior = np.zeros((p, p), dtype=np.int32)
        ii, jj = np.nonzero(np.triu(adj[:p, :p], 1))
        for k in rng.choice(ii.size, size=min(3, ii.size), replace=False) if ii.size else []:
            prior[jj[k], ii[k]] = 1
    return adj, corr, ixs, prior


# ---------------------------------------------------------------------------
# whole-chromosome file sets (BASELINE.json config 4: ~200 LD blocks x ~500 SNPs x 20 traits)
# ---------------------------------------------------------------------------
def chromosome_block_sizes(nblocks: int, seed: int = 0, mean: int = 500, lo: int = 100, hi: int = 2000, quantum: int = 100):
    """Unequal LD-block sizes (multiples of the LD window): log-normal around `mean`, clipped to [lo, hi].  Block b's
    size depends on (seed, b) only, so a job with more ranks extends the same chromosome."""
    out = []
    for b in range(nblocks):
        r = np.random.Generator(np.random.PCG64(BASE_SEED + 7919 * (seed + 1) + b))
        s = mean * float(np.exp(r.normal(-0.1, 0.6)))
        out.append(int(min(hi, max(lo, quantum * round(s / quantum)))))
    return out


def chromosome_segment(sizes, seg_first: int, seg_last: int, N: int, p: int, seed: int = 0, **kw):
    """Genotypes of blocks seg_first..seg_last-1 of the chromosome `sizes` (own RNG stream per block) and their
    genetic contribution to the p traits: (G int8 [m_seg, N], contrib float64 [p, N]).  Each trait gets
    5 causal SNPs per 1,000 with |beta| ~ U(0.02, 0.08) as in make_traits."""
    Gs, contrib = [], np.zeros((p, N), np.float64)
    for b in range(seg_first, seg_last):
        rng = np.random.Generator(np.random.PCG64(BASE_SEED + 104729 * (seed + 1) + b))
        G = make_genotypes(sizes[b], N, rng, **kw)
        g = G.astype(np.float32)
        g[G < 0] = np.nan
        mu = np.nanmean(g, axis=1, keepdims=True)
        sd = np.nanstd(g, axis=1, keepdims=True)
        sd[sd == 0] = 1.0
        ncausal_f = 5.0 * sizes[b] / 1000.0
        for k in range(p):
            nc = int(ncausal_f) + (1 if rng.random() < ncausal_f - int(ncausal_f) else 0)
            if nc == 0:
                continue
            idx = rng.choice(sizes[b], size=nc, replace=False)
            beta = rng.uniform(0.02, 0.08, size=nc) * rng.choice([-1.0, 1.0], size=nc)
            contrib[k] += beta @ np.nan_to_num((g[idx] - mu[idx]) / sd[idx]).astype(np.float64)
        Gs.append(G)
    return np.concatenate(Gs, axis=0), contrib


def chromosome_traits(contrib: np.ndarray, seed: int = 0) -> np.ndarray:
    """p x N float32 standardised traits from the summed genetic contributions: lower-triangular trait DAG
    (edge probability 3/p, |b| ~ U(0.05, 0.2)) + unit noise, as make_traits."""
    p, N = contrib.shape
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + 15485863 * (seed + 1)))
    Y = np.zeros((p, N), np.float64)
    for k in range(p):
        y = contrib[k].copy()
        for k2 in range(k):
            if rng.random() < 3.0 / p:
                y = y + rng.uniform(0.05, 0.2) * rng.choice([-1.0, 1.0]) * Y[k2]
        y = y + rng.standard_normal(N)
        Y[k] = (y - y.mean()) / y.std()
    return Y.astype(np.float32)


def write_blocks_file(path: str, sizes, chr_id: str = "1") -> None:
    first = 0
    with open(path, "w") as f:
        for s in sizes:
            f.write(f"{chr_id}\t{first}\t{first + s - 1}\n")
            first += s


def write_phen_fast(path: str, Y: np.ndarray) -> None:
    """same format as write_phen (header + `FID IID v1..vp` rows), %.9g values, written in one go"""
    p, N = Y.shape
    cols = Y.T.astype(np.float64)
    with open(path, "w") as f:
        f.write("FID IID " + " ".join(f"T{k}" for k in range(p)) + "\n")
        f.write("".join(f"f{i} i{i} " + " ".join("NA" if np.isnan(v) else "%.9g" % v for v in cols[i]) + "\n" for i in range(N)))


def rand_dag_corr(snp: int = 500, tr: int = 5, nl: int = 2, n: int = 16000, deg: float = 3.0, prob_pleio: float = 0.2,
                  lo_mp: float = 0.001, hi_mp: float = 0.05, lo_pp: float = 0.001, hi_pp: float = 0.2, seed: int = 1,
                  return_dag: bool = False):
    """BASELINE config 1: the simulated correlation matrix of the reference's random-DAG simulator, restated in NumPy
    (/root/reference/simulation/simulate_dag.R:3-98 `gen_rand_dag`, parameters :106-115 with SNP = 500, Tr = 5; own RNG
    stream -- PCG64(BASE_SEED + seed) -- so not draw-for-draw identical to R's).

    Variables in generation order: `snp` markers, `nl` latent confounders, `tr` traits (pq = snp + nl + tr).  Edges
    only point forward in that order (:22-28 marker -> any later variable with probability deg / snp; :42-48 latent or
    trait -> any later variable with probability min(deg / tr, 1)); a marker with exactly one trait child gets
    pleiotropic edges to the other traits with probability prob_pleio each (:30-40: the whole draw replaces the other
    entries, and only when it is not all zero).  Effects (:54-78): marker -> marker and latent / trait -> anything
    U(lo_pp, hi_pp), marker -> latent / trait U(lo_mp, hi_mp), random sign.  Data (:80-91): a root is N(0, 1), a child
    is its parents' linear combination g plus N(0, 1 - var(g)) noise (sample variance, n - 1 denominator as R's var).
    Returns the (snp + tr) x (snp + tr) fp32 sample correlation of markers and traits (latents dropped, :120-127:
    `cor(dag_data_mat)`), markers first -- the variable order of every n x n object on the path; with return_dag also
    the 0/1 DAG and the effect matrix over all pq variables."""
    rng = np.random.Generator(np.random.PCG64(BASE_SEED + 7919 * int(seed)))
    pq = snp + nl + tr
    t0 = snp + nl  # first trait
    prob1 = deg / snp
    prob2 = min(deg / tr, 1.0)
    G = np.zeros((pq, pq), np.int8)
    for i in range(snp):
        G[i, i + 1:] = rng.random(pq - i - 1) < prob1
    for i in range(snp):
        iv = np.flatnonzero(G[i, t0:] == 1)
        if iv.size == 1 and tr > 1:
            draw = (rng.random(tr - 1) < prob_pleio).astype(np.int8)
            if draw.any():
                others = np.delete(np.arange(tr), iv[0])
                G[i, t0 + others] = draw
    for j in range(snp, pq):
        G[j, j + 1:] = rng.random(pq - j - 1) < prob2

    def effects(k, lo, hi):
        return rng.uniform(lo, hi, k) * np.where(rng.standard_normal(k) < 0, -1.0, 1.0)

    A = np.zeros((pq, pq), np.float64)
    for i in range(snp):
        d = np.flatnonzero(G[i, :snp])
        A[i, d] = effects(d.size, lo_pp, hi_pp)
        d = snp + np.flatnonzero(G[i, snp:])
        A[i, d] = effects(d.size, lo_mp, hi_mp)
    for i in range(snp, pq):
        d = np.flatnonzero(G[i])
        A[i, d] = effects(d.size, lo_pp, hi_pp)
    X = np.empty((pq, n), np.float64)
    for i in range(pq):
        anc = np.flatnonzero(G[:, i])
        if anc.size == 0:
            X[i] = rng.standard_normal(n)
        else:
            g = A[anc, i] @ X[anc]
            X[i] = g + rng.standard_normal(n) * np.sqrt(max(1.0 - g.var(ddof=1), 0.0))
    keep = np.concatenate([np.arange(snp), np.arange(t0, pq)])
    Cm = np.corrcoef(X[keep]).astype(np.float32)
    Cm = np.triu(Cm, 1)
    Cm = Cm + Cm.T
    np.fill_diagonal(Cm, 1.0)
    Cm = np.ascontiguousarray(Cm, np.float32)
    if return_dag:
        return Cm, G, A
    return Cm
