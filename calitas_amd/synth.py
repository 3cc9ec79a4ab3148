"""Seeded synthetic references for tests and bench.py: i.i.d. ACGT with a GC fraction, soft-masked
(lower-case) runs, upper-case N runs, tandem repeats and planted near-matches of the guides on both strands.
Pure numpy; no reference data involved."""
import numpy as np

HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
                135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
                46709983, 50818468, 156040895, 57227415, 16569]
HG38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "chrM"]
ECOLI_LENGTH = 4641652

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
_IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "T", "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
          "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


def random_bases(rng, n, gc=0.41):
    """n ASCII bases as a uint8 array."""
    at = (1.0 - gc) / 2
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    # inverse-CDF on uniform bytes keeps this a single pass over a uint8 array
    u = rng.integers(0, 256, size=n, dtype=np.uint8)
    t = np.array([at, at + gc / 2, at + gc], dtype=np.float64) * 256
    idx = (u >= t[0]).astype(np.uint8)
    idx += u >= t[1]
    idx += u >= t[2]
    return lut[idx]


def realise(rng, pattern):
    """Concrete ACGT string compatible with an IUPAC pattern (case ignored)."""
    return "".join(rng.choice(list(_IUPAC[c])) for c in pattern.upper())


def mutate(rng, proto, n_edits, allow_gaps=True):
    """Applies n_edits random edits (mismatch / 1-base insertion / 1-base deletion) to an ACGT string."""
    s = list(proto)
    for _ in range(n_edits):
        kind = rng.integers(0, 3 if allow_gaps else 1)
        pos = int(rng.integers(0, len(s)))
        if kind == 0:
            s[pos] = rng.choice([b for b in "ACGT" if b != s[pos]])
        elif kind == 1:
            s.insert(pos, rng.choice(list("ACGT")))
        elif len(s) > 1:
            del s[pos]
    return "".join(s)


def plant_site(rng, seq, pos, protospacer, pam, pam5, n_edits, minus, gap_to_pam=0):
    """Writes a near-match of the guide into seq (uint8 array) at pos; returns the number of bases written."""
    site = mutate(rng, realise(rng, protospacer), n_edits)
    p = realise(rng, pam) if pam else ""
    filler = "".join(rng.choice(list("ACGT")) for _ in range(gap_to_pam))
    full = (p + filler + site) if pam5 else (site + filler + p)
    if minus:
        full = revcomp(full)
    b = np.frombuffer(full.encode(), dtype=np.uint8)
    end = min(len(seq), pos + len(b))
    if pos < 0 or end <= pos:
        return 0
    seq[pos:end] = b[:end - pos]
    return end - pos


def make_contig(rng, length, gc=0.41, softmask=0.5, n_run_ends=0, n_block=0, tandem_frac=0.01):
    seq = random_bases(rng, length, gc)
    # tandem repeats: units of 2-7 bp, tracts of 30-300 bp
    if tandem_frac > 0 and length > 1000:
        n_tr = max(1, int(length * tandem_frac / 150))
        starts = rng.integers(0, max(1, length - 400), size=n_tr)
        for s in starts:
            unit = int(rng.integers(2, 8))
            tract = int(rng.integers(30, 300))
            u = seq[s:s + unit].copy()
            reps = np.tile(u, tract // unit + 1)[:tract]
            e = min(length, s + tract)
            seq[s:e] = reps[:e - s]
    # soft masking: lower-case runs of 0.3-5 kb covering ~softmask of the contig
    if softmask > 0 and length > 2000:
        mean_run = 2650
        n_runs = int(length * softmask / mean_run)
        starts = rng.integers(0, length, size=n_runs)
        lens = rng.integers(300, 5000, size=n_runs)
        delta = np.zeros(length + 1, dtype=np.int32)
        np.add.at(delta, starts, 1)
        np.add.at(delta, np.minimum(starts + lens, length), -1)
        low = np.cumsum(delta[:-1]) > 0
        seq[low] |= 0x20
    if n_run_ends > 0:
        seq[:min(n_run_ends, length)] = ord("N")
        seq[max(0, length - n_run_ends):] = ord("N")
    if n_block > 0 and 2 * length // 3 - n_block > length // 3:
        s = int(rng.integers(length // 3, 2 * length // 3 - n_block))
        seq[s:s + n_block] = ord("N")
    return seq


def make_genome(spec, seed, guides=(), sites_per_guide=40, step_hint=971, **kw):
    """spec: list of (name, length).  guides: list of (protospacer, pam, pam_is_5prime).
    Returns (names, [uint8 arrays]).  Each contig uses its own seed so ranks can build shards independently."""
    names, seqs = [], []
    for ci, (name, length) in enumerate(spec):
        rng = np.random.default_rng([seed, ci])
        seq = make_contig(rng, length, **kw)
        if length > 5000:
            for gi, (proto, pam, pam5) in enumerate(guides):
                # share of this contig in the planted sites, at least one
                n_sites = max(1, int(round(sites_per_guide * length / max(1, sum(l for _, l in spec)))))
                for k in range(n_sites):
                    n_edits = int(rng.integers(0, 7))
                    minus = bool(rng.integers(0, 2))
                    mode = k % 4
                    if mode == 0 and step_hint > 0:   # straddle a window start
                        w = int(rng.integers(1, max(2, length // step_hint)))
                        pos = w * step_hint - int(rng.integers(0, 40))
                    elif mode == 1:                   # near a contig end
                        pos = int(rng.integers(0, 60)) if rng.integers(0, 2) else length - int(rng.integers(20, 80))
                    else:
                        pos = int(rng.integers(0, length - 64))
                    plant_site(rng, seq, pos, proto, pam, pam5, n_edits, minus, gap_to_pam=int(rng.integers(0, 3)) if k % 3 == 0 else 0)
        names.append(name)
        seqs.append(seq)
    return names, seqs


def hg38_like_spec(scale=1.0):
    return [(n, max(1000, int(l * scale))) for n, l in zip(HG38_NAMES, HG38_LENGTHS)]


def random_guides(seed, n, length=20, pam="nrg"):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        proto = "".join(rng.choice(list("ACGT"), size=length))
        out.append(proto + pam)
    return out
