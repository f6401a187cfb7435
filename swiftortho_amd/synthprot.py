"""Deterministic synthetic proteomes for the find_hit benchmark configs.

``synthprot(N, L=300, seed=0x5EED0001)`` -> FASTA bytes of exactly N proteins:

* taxa  T = max(2, N // 2000), ids ``>t%04d|p%07d`` (taxon | global protein ordinal)
* families F = N / (0.8*T); the ancestor of a family is L residues drawn iid from
  the Robinson-Robinson background frequencies (20 standard letters, upper case)
* every taxon receives a copy of every family with p = 0.8; 5 % of copies get an
  in-paralog duplicate in the same taxon
* each copy: substitutions at rate d ~ U(0.05, 0.6) (per member), deletions started
  at 1 %/site and insertions at 1 %/site, lengths geometric with mean 2
* sequences are written taxon-major, one line per sequence

``synthprot(N, lengths="lognormal")`` is the LENGTH-HETEROGENEOUS variant (round 4): the same taxa / family / divergence
model, but every family has its own ancestor length -- log-normal with median 270 (sigma 0.6, clipped to 60 ... 5000), one
family in two hundred drawn from U(1500, 5000) instead (the multi-domain tail of real proteomes), and one singleton protein
of 30 000 residues per 50 000 proteins -- so that a 50 k-sequence reference chunk always holds sequences far above every
fixed-width assumption (packed 16-bit alignment cells, 32-bit hit words).  The uniform variant's random stream is untouched.

``uniform_proteins(N, L, seed)`` is the adversarial companion set: iid residues, no
homologs (exercises the high-frequency seed cap and the phase-2 early stop).

The "seed 111111" in BASELINE.json's configs is the *spaced-seed pattern*, not an
RNG seed; the RNG seed here is fixed so every box generates identical input.
"""
import numpy as np

AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
# Robinson & Robinson (1991) background frequencies, order = AA above
RR_FREQ = np.array([0.07805, 0.05129, 0.04487, 0.05364, 0.01925, 0.04264, 0.06295, 0.07377,
                    0.02199, 0.05142, 0.09019, 0.05744, 0.02243, 0.03856, 0.05203, 0.07120,
                    0.05841, 0.01330, 0.03216, 0.06441], dtype=np.float64)
RR_FREQ = RR_FREQ / RR_FREQ.sum()


def _draw(rng, n):
    return rng.choice(20, size=n, p=RR_FREQ).astype(np.uint8)


def _evolve(rng, anc, d, valid=None):
    """anc: (M, L) uint8 residue indices, d: (M,) substitution rates; valid: optional (M, L) mask of the
    positions that belong to the (shorter) ancestors of a padded batch.
    Returns (flat residues, lengths) after substitution + indels."""
    M, L = anc.shape
    seq = anc.copy()
    sub = rng.random((M, L)) < d[:, None]
    seq[sub] = _draw(rng, int(sub.sum()))
    # deletions: a start site deletes g >= 1 sites, g ~ Geom(0.5)
    dstart = rng.random((M, L)) < 0.01
    dlen = np.where(dstart, rng.geometric(0.5, size=(M, L)), 0)
    deleted = np.zeros((M, L), dtype=bool)
    for k in range(8):
        sh = np.zeros((M, L), dtype=bool)
        if k == 0:
            sh = dlen > 0
        else:
            sh[:, k:] = dlen[:, :-k] > k
        deleted |= sh
    ins = np.where(rng.random((M, L)) < 0.01, rng.geometric(0.5, size=(M, L)), 0)
    keep = (~deleted).astype(np.int64)
    if valid is not None:
        keep = keep * valid
        ins = ins * valid
    rep = keep + ins
    # never let a sequence vanish
    rep[:, 0] = np.maximum(rep[:, 0], 1)
    keep[:, 0] = np.maximum(keep[:, 0], 1)
    flat = np.repeat(seq.ravel(), rep.ravel())
    starts = np.cumsum(rep.ravel()) - rep.ravel()
    pos_in = np.arange(flat.size, dtype=np.int64) - np.repeat(starts, rep.ravel())
    is_ins = pos_in >= np.repeat(keep.ravel(), rep.ravel())
    flat[is_ins] = _draw(rng, int(is_ins.sum()))
    return flat, rep.sum(axis=1)


def _family_lengths(rng, F):
    ln = np.exp(rng.normal(np.log(270.0), 0.6, size=F))
    tail = rng.random(F) < 0.005
    ln[tail] = rng.uniform(1500.0, 5000.0, size=int(tail.sum()))
    return np.clip(np.rint(ln), 60, 5000).astype(np.int64)


def _synthprot_het(N, seed):
    """Length-heterogeneous proteome (see the module text): (residues, lengths, taxa)."""
    rng = np.random.default_rng(seed ^ 0x4E7)
    T = max(2, N // 2000)
    G = max(1, N // 50000)           # 30 000-residue singletons
    Nf = N - G
    F = max(1, int(round(Nf / (0.8 * T))))
    present = rng.random((F, T)) < 0.8
    fam, tax = np.nonzero(present)
    dup = rng.random(fam.size) < 0.05
    fam = np.concatenate([fam, fam[dup]])
    tax = np.concatenate([tax, tax[dup]])
    n = fam.size
    if n > Nf:
        sel = np.sort(rng.choice(n, size=Nf, replace=False))
        fam, tax = fam[sel], tax[sel]
    elif n < Nf:
        extra = Nf - n
        fam = np.concatenate([fam, F + np.arange(extra)])
        tax = np.concatenate([tax, rng.integers(0, T, size=extra)])
        F += extra
    flen = _family_lengths(rng, F)
    # the giants: families of their own with one member each
    fam = np.concatenate([fam, F + np.arange(G)])
    tax = np.concatenate([tax, rng.integers(0, T, size=G)])
    flen = np.concatenate([flen, np.full(G, 30000, dtype=np.int64)])
    F += G
    order = np.lexsort((rng.random(N), tax))
    fam, tax = fam[order], tax[order]
    aoff = np.concatenate([[0], np.cumsum(flen)])
    anc = _draw(rng, int(aoff[-1]))
    d = rng.uniform(0.05, 0.6, size=N)
    # members are evolved in batches of similar ancestor length (padded to the batch's longest), then put back in file order
    by_len = np.argsort(flen[fam], kind="stable")
    out_res = [None] * N
    lens = np.zeros(N, dtype=np.int64)
    s = 0
    while s < N:
        Lmax = int(flen[fam[by_len[s]]])
        e = s + 1
        while e < N and e - s < 20000:
            Ln = int(flen[fam[by_len[e]]])
            if (e - s + 1) * Ln > 6000000:
                break
            Lmax = Ln
            e += 1
        idx = by_len[s:e]
        M = idx.size
        cols = np.arange(Lmax)
        fl = flen[fam[idx]]
        valid = (cols[None, :] < fl[:, None])
        src = aoff[fam[idx]][:, None] + np.minimum(cols[None, :], fl[:, None] - 1)
        flat, ln = _evolve(rng, anc[src], d[idx], valid.astype(np.int64))
        o = np.concatenate([[0], np.cumsum(ln)])
        for k in range(M):
            out_res[idx[k]] = AA[flat[o[k]:o[k + 1]]]
            lens[idx[k]] = ln[k]
        s = e
    return np.concatenate(out_res), lens, tax.astype(np.int32)


def synthprot_arrays(N, L=300, seed=0x5EED0001, lengths="uniform"):
    """Return (residues uint8 ASCII flat array, lengths int64[N], taxon int32[N])."""
    if lengths == "lognormal":
        return _synthprot_het(N, seed)
    if lengths != "uniform":
        raise ValueError("lengths must be 'uniform' or 'lognormal'")
    rng = np.random.default_rng(seed)
    T = max(2, N // 2000)
    F = max(1, int(round(N / (0.8 * T))))
    present = rng.random((F, T)) < 0.8
    fam, tax = np.nonzero(present)
    dup = rng.random(fam.size) < 0.05
    fam = np.concatenate([fam, fam[dup]])
    tax = np.concatenate([tax, tax[dup]])
    n = fam.size
    if n > N:
        sel = np.sort(rng.choice(n, size=N, replace=False))
        fam, tax = fam[sel], tax[sel]
    elif n < N:
        extra = N - n
        fam = np.concatenate([fam, F + np.arange(extra)])
        tax = np.concatenate([tax, rng.integers(0, T, size=extra)])
        F += extra
    # taxon-major order, random family order inside a taxon
    order = np.lexsort((rng.random(N), tax))
    fam, tax = fam[order], tax[order]
    ancestors = _draw(rng, F * L).reshape(F, L)
    d = rng.uniform(0.05, 0.6, size=N)
    chunks, lens = [], []
    B = 20000
    for s in range(0, N, B):
        e = min(N, s + B)
        flat, ln = _evolve(rng, ancestors[fam[s:e]], d[s:e])
        chunks.append(AA[flat])
        lens.append(ln)
    return np.concatenate(chunks), np.concatenate(lens).astype(np.int64), tax.astype(np.int32)


def _to_fasta(res, lens, tax):
    out = []
    off = 0
    mv = res.tobytes()
    for i in range(lens.size):
        ln = int(lens[i])
        out.append(b">t%04d|p%07d\n" % (int(tax[i]), i))
        out.append(mv[off:off + ln])
        out.append(b"\n")
        off += ln
    return b"".join(out)


def synthprot(N, L=300, seed=0x5EED0001, lengths="uniform"):
    return _to_fasta(*synthprot_arrays(N, L, seed, lengths))


def uniform_proteins(N, L=300, seed=0x5EED0002):
    rng = np.random.default_rng(seed)
    res = AA[rng.integers(0, 20, size=N * L).astype(np.uint8)]
    lens = np.full(N, L, dtype=np.int64)
    tax = (np.arange(N) * max(2, N // 2000) // max(N, 1)).astype(np.int32)
    return _to_fasta(res, lens, tax)


def write(path, data):
    with open(path, "wb") as f:
        f.write(data)


if __name__ == "__main__":
    import sys
    n = int(sys.argv[1])
    write(sys.argv[2], synthprot(n) if len(sys.argv) < 4 else uniform_proteins(n))
