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


def _evolve(rng, anc, d):
    """anc: (M, L) uint8 residue indices, d: (M,) substitution rates.
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


def synthprot_arrays(N, L=300, seed=0x5EED0001):
    """Return (residues uint8 ASCII flat array, lengths int64[N], taxon int32[N])."""
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


def synthprot(N, L=300, seed=0x5EED0001):
    return _to_fasta(*synthprot_arrays(N, L, seed))


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
