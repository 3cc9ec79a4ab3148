"""GPU: the candidate filter (scan_rows.hip) on its own, through calitas_scan_candidates.

The filter decides which end columns the aligner kernel looks at: every position and strand whose seamless glocal bottom-row
score reaches minGuideScore, which for the reference's linear costs is "semi-global edit distance of the protospacer <= E"
(SearchReference.scala:432-441; enumeration rule of fgbio's Aligner.align(query, target, minScore) as called at
SequentialGuideAligner.scala:261-299).  Checked here against (a) a plain numpy dynamic programme of that definition and
(b) the first-generation column-wise kernel, which must emit the same records bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SETS = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3, "B": 14, "D": 13, "H": 11, "V": 7, "N": 15}
_COMP4 = [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]   # IUPAC set of the complementary bases


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


def target_sets(seq):
    """Per base of an ASCII contig: (set of ACGT it can stand for as a 4-bit mask, wildcard flag).  The filter's rule for the
    target: ACGT/U match their own letter, N / unknown bytes match nothing, any other IUPAC code matches every row."""
    s = np.frombuffer(seq.upper().encode(), dtype=np.uint8)
    sets = np.zeros(len(s), dtype=np.uint8)
    wild = np.zeros(len(s), dtype=bool)
    for ch, m in _SETS.items():
        sel = s == ord(ch)
        if ch in "ACGTU":
            sets[sel] = m
        elif ch != "N":
            wild[sel] = True
    return sets, wild


def last_row(query_sets, sets, wild):
    """Bottom row of the semi-global edit-distance matrix (free start in the target), one value per target position."""
    n = len(sets)
    prev = np.zeros(n + 1, dtype=np.int32)
    idx = np.arange(n + 1, dtype=np.int32)
    for i, q in enumerate(query_sets, start=1):
        match = wild | ((sets & q) != 0)
        sub = prev[:-1] + (~match).astype(np.int32)
        up = prev[1:] + 1
        m = np.minimum(sub, up)
        v = np.concatenate(([i], m)).astype(np.int32) - idx     # cur[j] = min over j' <= j of (v[j'] + j - j')
        prev = np.minimum.accumulate(v) + idx
    return prev[1:]


def dp_candidates(contigs, guides, E):
    """[(contig, offset, pass, guide)] by definition.  Pass 1 = the guide against the reverse complement; its end column is
    reported at the contig offset of the alignment's first base in forward coordinates."""
    out = []
    for ci, (_, seq) in enumerate(contigs):
        if not seq:
            continue
        sets, wild = target_sets(seq)
        rsets = np.array([_COMP4[x] for x in sets[::-1]], dtype=np.uint8)
        rwild = wild[::-1]
        for gi, proto in enumerate(guides):
            q = [_SETS[c] for c in proto.upper()]
            fw = last_row(q, sets, wild)
            out += [(ci, int(j), 0, gi) for j in np.nonzero(fw <= E)[0]]
            rv = last_row(q, rsets, rwild)
            out += [(ci, len(seq) - 1 - int(j), 1, gi) for j in np.nonzero(rv <= E)[0]]
    out.sort()
    return out


def genome(seed, guides, lengths=(70000, 30011, 2000, 95, 31, 12)):
    from calitas_amd import synth
    rng = np.random.default_rng(seed)
    contigs = []
    for ci, n in enumerate(lengths):
        seq = synth.make_contig(rng, n, softmask=0.3, n_run_ends=40 if n > 1000 else 0, n_block=700 if n > 20000 else 0, tandem_frac=0.02)
        if n > 1000:
            for proto in guides:
                for k in range(40):
                    pos = int(rng.integers(0, n - 40))
                    if k % 5 == 0:
                        pos = (pos // 512) * 512 - int(rng.integers(0, 24))       # straddling scan-lane and tile boundaries
                    if k % 7 == 0:
                        pos = int(rng.integers(0, 30)) if k % 2 else n - int(rng.integers(20, 60))
                    synth.plant_site(rng, seq, max(0, pos), proto, "", False, int(rng.integers(0, 8)), bool(rng.integers(0, 2)))
            for pos in rng.integers(0, n, size=n // 400):                             # IUPAC codes and stray N in the target
                seq[pos] = ord(rng.choice(list("RYKMSWBDHVNn")))
        contigs.append(("ctg%d" % ci, seq.tobytes().decode()))
    return contigs


CASES = [
    # id, guides (same length), d, chunk
    ("d5-chunk512", ["CTTGCCCCACAGGGCAGTAA"], 5, "512"),
    ("d5-chunk64", ["CTTGCCCCACAGGGCAGTAA"], 5, "64"),
    ("d3-chunk128-two-guides", ["CTTGCCCCACAGGGCAGTAA", "GACCTTGAAGTCTCAGTATA"], 3, "128"),
    ("d8-chunk256", ["CTTGCCCCACAGGGCAGTAA"], 8, "256"),
    ("iupac-guide-d4", ["GAGAATTGNTTGAACCCRGG"], 4, "512"),
    ("L32-d6-two-warm-up-words", ["CTTGCCCCACAGGGCAGTAACGGTTCAATGCA"], 6, "512"),
    ("L32-d6-chunk64", ["CTTGCCCCACAGGGCAGTAACGGTTCAATGCA"], 6, "64"),
    ("L12-d2", ["GCAGTAACCTGA"], 2, "256"),
    ("d0", ["CTTGCCCCACAGGGCAGTAA"], 0, "512"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_scan_candidates_equal_plain_dp_and_columnwise_kernel(C, case, monkeypatch):
    cid, guides, d, chunk = case
    contigs = genome(len(cid) + d, guides)
    monkeypatch.setenv("CALITAS_CHUNK", chunk)        # bases per scan lane (read when the reference is packed)
    ctx = C.Context(0)
    try:
        ctx.set_reference([n for n, _ in contigs], [s.encode() for _, s in contigs])
        G = [C.Guide(g) for g in guides]              # PAM-less: the filter only sees the protospacer
        params = C.make_params(max_guide_diffs=d, max_pam_mismatches=0, max_gaps_between_guide_and_pam=0)
        rows_kernel = ctx.scan_candidates(G, params)
        cols_kernel = ctx.scan_candidates(G, params, columnwise=True)   # round 1's kernel: an independent second implementation
    finally:
        ctx.close()
    want = dp_candidates(contigs, guides, d)          # default costs: a bottom-row score >= minGuideScore <=> <= d edits
    assert len(want) > 10
    assert rows_kernel == want, (cid, len(rows_kernel), len(want), sorted(set(rows_kernel) ^ set(want))[:5])
    assert cols_kernel == want, (cid, len(cols_kernel), len(want), sorted(set(cols_kernel) ^ set(want))[:5])


def test_scan_queue_overflow_on_dense_repeats(C, monkeypatch):
    """A homopolymer matched by a homopolymer guide flags every word of every lane: the tile's suspect queue and record stage
    overflow, and the kernel has to resolve / append in place without losing a column."""
    contigs = [("polyA", "A" * 40000 + "C" * 3000 + "A" * 9000), ("mixed", "ACGT" * 2000 + "A" * 5000)]
    monkeypatch.setenv("CALITAS_CHUNK", "512")
    ctx = C.Context(0)
    try:
        ctx.set_reference([n for n, _ in contigs], [s.encode() for _, s in contigs])
        got = ctx.scan_candidates([C.Guide("A" * 20)], C.make_params(max_guide_diffs=2, max_pam_mismatches=0, max_gaps_between_guide_and_pam=0))
    finally:
        ctx.close()
    want = dp_candidates(contigs, ["A" * 20], 2)
    assert len(want) > 50000
    assert got == want
