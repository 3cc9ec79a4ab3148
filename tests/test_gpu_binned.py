"""GPU: the per-bin tail (binned.hip) -- per-window filter, coordinates, ReferenceHit.sort, removeOverlaps and rows for the hits whose
coordinate_start lies in an 8-kb bin, from the bin and the edges of its neighbours -- against the oracle and against the general
device kernels it stands in front of, with the cases that stress what is new: hits and overlap clusters on bin boundaries, windows that
straddle them, repeats long enough for a bin to decline (the call must then finish on the general kernels with the same text), the
text buffer's regrow path."""
import os

import numpy as np
import pytest

import oracle_lib as O
from fasta_util import write_fasta

pytestmark = pytest.mark.gpu

SKIP_COLS = {"aligner_version", "time_stamp"}
BIN = 8192
GUIDE = "CTTGCCCCACAGGGCAGTAAnrg"


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


def strip(rows):
    return [{k: v for k, v in r.items() if k not in SKIP_COLS} for r in rows]


def both_paths(C, fa, guide, monkeypatch, expect_binned=True, **pk):
    """(rows, binned_lanes): the default call and the same call on the general kernels return the same text."""
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        monkeypatch.delenv("CALITAS_BINNED", raising=False)
        text, n = C.SearchReference(guide=guide, guide_id="a", context=ctx, **pk).run("v0", "stamp")
        lanes = ctx.timing()["binned_lanes"]
        monkeypatch.setenv("CALITAS_BINNED", "0")
        text0, n0 = C.SearchReference(guide=guide, guide_id="a", context=ctx, **pk).run("v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 0
        monkeypatch.delenv("CALITAS_BINNED")
        # every bin through the wave-per-bin kernel (by default only the bins with more than four alignments in their context get there)
        monkeypatch.setenv("CALITAS_BINNED_COMPLEX", "1")
        text1, n1 = C.SearchReference(guide=guide, guide_id="a", context=ctx, **pk).run("v0", "stamp")
        lanes1 = ctx.timing()["binned_lanes"]
        monkeypatch.delenv("CALITAS_BINNED_COMPLEX")
        # a text of up to 128 KB is written into page-locked host memory by the rows kernel; this one takes the device buffer and the copy
        had = os.environ.get("CALITAS_BINNED_HOST_TEXT")
        monkeypatch.setenv("CALITAS_BINNED_HOST_TEXT", "0")
        text2, n2 = C.SearchReference(guide=guide, guide_id="a", context=ctx, **pk).run("v0", "stamp")
        lanes2 = ctx.timing()["binned_lanes"]
        if had is None:
            monkeypatch.delenv("CALITAS_BINNED_HOST_TEXT")
        else:
            monkeypatch.setenv("CALITAS_BINNED_HOST_TEXT", had)
    finally:
        ctx.close()
    assert text == text0 and n == n0
    assert text1 == text0 and lanes1 == lanes
    assert text2 == text0 and lanes2 == lanes
    if expect_binned is not None:
        assert (lanes > 0) == expect_binned, lanes
    return C.read_hits(text), lanes


def planted(rng, length, sites, site="CTTGCCCCACAGGGCAGTAATGG"):
    """Random contig with (position, edits, reverse) sites planted: the site itself, mutated at `edits` protospacer positions."""
    seq = rng.choice(list(b"ACGT"), size=length).astype(np.uint8)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    for pos, edits, rev in sites:
        s = bytearray(site.encode())
        for k in rng.choice(20, size=edits, replace=False):
            s[k] = ord("ACGT"[("ACGT".index(chr(s[k])) + 1) % 4])
        if rev:
            s = bytearray(comp[c] for c in reversed(s))
        seq[pos:pos + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
    return seq.tobytes().decode()


def test_hits_and_clusters_on_bin_boundaries(C, tmp_path, monkeypatch):
    """Sites planted around every bin boundary of two contigs: starting just left of it, just right of it, straddling it, on both
    strands, in overlapping pairs (one of each pair is removed by removeOverlaps, whichever bin its partner falls into), and around the
    window starts nearest the boundary."""
    rng = np.random.default_rng(1234)
    sites = []
    for b in (BIN, 2 * BIN, 3 * BIN):
        for d in (-40, -23, -22, -12, -1, 0, 1, 7, 30):
            sites.append((b + d, int(rng.integers(0, 4)), bool(rng.integers(0, 2))))
    c1 = planted(rng, 3 * BIN + 9000, sites[:18] + [(3 * BIN + 5, 1, False)])
    # overlapping pairs: the same site twice, 6 / 12 bases apart, across and next to a boundary (a fresh contig: no other sites around)
    pairs = []
    for b, off in ((BIN, -30), (BIN, -9), (BIN, 3), (2 * BIN, -15), (2 * BIN, 0)):
        pairs += [(b + off, 0, False), (b + off + 29, 1, False)]            # overlap of the full alignments: 23 + 23 - 29 < 10: both stay
    c2 = planted(rng, 2 * BIN + 5000, pairs + [(2 * BIN - 971 * 3 + k, 2, True) for k in (-20, 5)])
    # tandem copies 8 bases apart (they overlap by more than maxOverlap): chains that cross the boundary
    chain = [(BIN - 60 + 8 * k, 0, False) for k in range(14)]
    c3 = planted(rng, BIN + 4000, [], site="A")                              # placeholder, filled below
    seq3 = bytearray(c3.encode())
    unit = b"CTTGCCCCACAGGGCAGTAATGG"
    for pos, _, _ in chain:
        seq3[pos:pos + len(unit)] = unit
    fa = write_fasta(str(tmp_path / "edges.fa"), [("c1", c1), ("c2", c2), ("c3", seq3.decode()), ("short", "ACGT" * 30)])
    for pk in (dict(max_gaps_between_guide_and_pam=2), dict(max_guide_diffs=3, max_overlap=3), dict(max_overlap=40)):
        # (with -O 40 the chain of tandem copies keeps more alignments than a bin's wave holds: either tail may finish that call)
        rows, lanes = both_paths(C, fa, GUIDE, monkeypatch, expect_binned=None if pk.get("max_overlap", 10) > 10 else True, **pk)
        _, want, _ = O.search_reference(fa, GUIDE, "a", d=pk.get("max_guide_diffs", 5), g=pk.get("max_gaps_between_guide_and_pam", 3),
                                        O=pk.get("max_overlap", 10), threads=4)
        assert len(want) > 10
        assert strip(rows) == strip(want), pk
        assert {r["chromosome"] for r in rows} >= {"c1", "c2", "c3"}


def test_long_repeat_declines_and_the_general_kernels_finish(C, tmp_path, monkeypatch):
    """A 3-kb tandem array of the site itself: hundreds of alignments in one bin and a chain of overlapping hits longer than the
    halo.  The bins decline (crowded / halo), the general kernels finish the call with the oracle's rows; the context remembers it
    (the next, equally permissive search does not try the bins again) and a stricter search takes the bins again."""
    rng = np.random.default_rng(77)
    base = rng.choice(list(b"ACGT"), size=2 * BIN + 3000).astype(np.uint8)
    unit = np.frombuffer(b"CTTGCCCCACAGGGCAGTAATGGAC", dtype=np.uint8)
    rep = np.tile(unit, 130)                                             # 3250 bases
    base[BIN - 1700:BIN - 1700 + len(rep)] = rep
    fa = write_fasta(str(tmp_path / "repeat.fa"), [("r1", base.tobytes().decode()), ("r2", planted(rng, 40000, [(1000, 1, False), (33000, 2, True)]))])
    rows, lanes = both_paths(C, fa, GUIDE, monkeypatch, expect_binned=False, max_gaps_between_guide_and_pam=2)
    _, want, _ = O.search_reference(fa, GUIDE, "a", g=2, threads=4)
    assert len(want) > 100 and strip(rows) == strip(want)
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        sr = C.SearchReference(guide=GUIDE, guide_id="a", context=ctx, max_gaps_between_guide_and_pam=2)
        t1, _ = sr.run("v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 0
        t2, _ = sr.run("v0", "stamp")                                    # remembered: straight to the general kernels
        assert ctx.timing()["binned_lanes"] == 0 and t2 == t1
        strict = C.SearchReference(guide="GTGACTTGAAGTCTCAGTATAnrg", guide_id="b", context=ctx, max_guide_diffs=2)
        strict.run("v0", "stamp")                                        # another guide length: not covered by the memory
        assert ctx.timing()["binned_lanes"] == 1
    finally:
        ctx.close()


def test_text_buffer_regrow(C, tmp_path, monkeypatch):
    """The rows kernel writes into a buffer sized by a guess; a text that does not fit makes it return at once (BIN_FLAG_TEXT), the
    host grows the buffer and launches it again.  Forced with a 1-KB first guess."""
    rng = np.random.default_rng(5)
    sites = [(int(p), int(rng.integers(0, 5)), bool(rng.integers(0, 2))) for p in rng.integers(200, 95000, size=80)]
    fa = write_fasta(str(tmp_path / "regrow.fa"), [("c", planted(rng, 100000, sites))])
    monkeypatch.setenv("CALITAS_BINNED_TEXT_KB", "1")
    rows, lanes = both_paths(C, fa, GUIDE, monkeypatch, max_gaps_between_guide_and_pam=2)
    _, want, _ = O.search_reference(fa, GUIDE, "a", g=2, threads=4)
    assert len(want) > 40 and strip(rows) == strip(want)


def test_lanes_and_batch_take_the_binned_tail(C, tmp_path, monkeypatch):
    """Contig ranges (lanes) and a guide batch: every lane / guide reports the binned tail, same bytes as one pass."""
    rng = np.random.default_rng(9)
    contigs = []
    for ci, length in enumerate((70000, 40000, 90000, 33000)):
        sites = [(int(p), int(rng.integers(0, 5)), bool(rng.integers(0, 2))) for p in rng.integers(100, length - 100, size=25)]
        contigs.append(("k%d" % ci, planted(rng, length, sites)))
    fa = write_fasta(str(tmp_path / "lanes.fa"), contigs)
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        sr = C.SearchReference(guide=GUIDE, guide_id="a", context=ctx, max_gaps_between_guide_and_pam=2)
        one, _ = sr.run("v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 1
        monkeypatch.setenv("CALITAS_CHUNKS", "2")
        two, _ = sr.run("v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 2 and two == one
        # where the ranges' scan inputs are queued (search.cpp: each before its own scan / all ahead of the first / on the ranges' own
        # streams, the default) and whether they are one setup launch or separate commands
        for mode, setup in (("0", "1"), ("1", "1"), ("2", "0"), ("1", "0")):
            monkeypatch.setenv("CALITAS_INPUTS_FIRST", mode)
            monkeypatch.setenv("CALITAS_LANE_SETUP", setup)
            again, _ = sr.run("v0", "stamp")
            assert again == one, (mode, setup)
        monkeypatch.delenv("CALITAS_INPUTS_FIRST")
        monkeypatch.delenv("CALITAS_LANE_SETUP")
        # the last range's text written to its final place by the rows kernel (default) against the copy, into the library's buffer
        # and into a page-locked buffer of the caller
        monkeypatch.setenv("CALITAS_TEXT_IN_PLACE_OFF", "1")
        copied, _ = sr.run("v0", "stamp")
        monkeypatch.delenv("CALITAS_TEXT_IN_PLACE_OFF")
        assert copied == one
        buf = np.zeros(1 << 22, dtype=np.uint8)
        ctx.pin_host(buf.ctypes.data, buf.nbytes)
        try:
            params = C.make_params(max_gaps_between_guide_and_pam=2)
            for _ in range(3):
                buf[:] = 0
                nb, nr = ctx.search_hits_into(C.Guide(GUIDE), "a", params, buf.ctypes.data, buf.nbytes, "v0", "stamp")
                assert bytes(buf[:nb]).decode() == one
        finally:
            ctx.unpin_host(buf.ctypes.data)
        monkeypatch.setenv("CALITAS_CHUNKS", "3")                 # three ranges: the general kernels beside the scans, the per-bin ones
        three, _ = sr.run("v0", "stamp")                           # for the last range ...
        assert ctx.timing()["binned_lanes"] == 1 and three == one
        monkeypatch.setenv("CALITAS_BINNED", "0")                 # ... none on request
        three, _ = sr.run("v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 0 and three == one
        monkeypatch.setenv("CALITAS_BINNED", "1")                 # ... the per-bin ones on request
        three, _ = sr.run("v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 3 and three == one
        monkeypatch.delenv("CALITAS_BINNED")
        monkeypatch.delenv("CALITAS_CHUNKS")
        G = [C.Guide(GUIDE), C.Guide("GTGACTTGAAGTCTCAGTATnrg"), C.Guide(GUIDE)]
        res = ctx.search_hits_batch(G, ["a", "b", "a"], C.make_params(max_gaps_between_guide_and_pam=2), "v0", "stamp")
        assert ctx.timing()["binned_lanes"] == 3
        assert res[0][0] == one and res[2][0] == one
    finally:
        ctx.close()
    _, want, _ = O.search_reference(fa, GUIDE, "a", g=2, threads=4)
    assert strip(C.read_hits(one)) == strip(want)


def test_compact_rows_arrive_in_pieces(C, tmp_path, monkeypatch):
    """The ranges of a chunked call move compact rows over PCIe in pieces and the worker pool expands what has landed (search.cpp
    compact_rows_to_host, post.cpp RowExpansion).  On a dense genome with 64 KB pieces and the job's hand-overs forced onto several
    workers: the same text as full rows over the bus, as one copy per range, and as the call in one pass."""
    rng = np.random.default_rng(77)
    contigs = []
    for ci, length in enumerate((400000, 300000, 350000, 250000)):
        pos = np.sort(rng.choice(np.arange(100, length - 100, 60), size=length // 90, replace=False))
        sites = [(int(p), int(rng.integers(0, 4)), bool(rng.integers(0, 2))) for p in pos]
        contigs.append(("d%d" % ci, planted(rng, length, sites)))
    fa = write_fasta(str(tmp_path / "dense.fa"), contigs)
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        sr = C.SearchReference(guide=GUIDE, guide_id="a", context=ctx, max_gaps_between_guide_and_pam=2)
        one, n_one = sr.run("v0", "stamp")
        assert n_one > 10000 and len(one) > 3 * (1 << 20)     # (every range's compact text: several pieces)
        monkeypatch.setenv("CALITAS_CHUNKS", "3")
        monkeypatch.setenv("CALITAS_COMPACT_ROWS", "0")
        full, _ = sr.run("v0", "stamp")
        assert full == one
        monkeypatch.delenv("CALITAS_COMPACT_ROWS")
        for piece, threads in (("64", "4"), ("64", "1"), ("128", "16"), ("65536", "3")):
            monkeypatch.setenv("CALITAS_COMPACT_PIECE_KB", piece)
            monkeypatch.setenv("CALITAS_EXPAND_THREADS", threads)
            for _ in range(3):
                again, n_again = sr.run("v0", "stamp")
                assert n_again == n_one and again == one, (piece, threads)
        # the same through the batch entry point (every guide's text is compact there)
        res = ctx.search_hits_batch([C.Guide(GUIDE)] * 3, ["a", "a", "a"], C.make_params(max_gaps_between_guide_and_pam=2), "v0", "stamp")
        assert all(t == one for t, _ in res)
    finally:
        ctx.close()


@pytest.mark.parametrize("cuts", [2, 3, 8])
def test_window_ranges_of_search_hits_concatenate(C, tmp_path, monkeypatch, cuts):
    """calitas_search_hits on a window range returns the rows whose coordinate_start lies in the range's stretch of the genome: the
    texts of consecutive ranges (minus their header lines) are the text of the whole call, wherever the cuts fall -- inside a contig,
    between two overlapping hits, inside a chain of tandem copies (there the bins decline and the stretch's contigs are searched whole
    and filtered: same rows).  Cut points are chosen to fall on such places."""
    from calitas_amd import shard
    rng = np.random.default_rng(100 + cuts)
    step = 1000 - (len(GUIDE) + 5 + 2 - 1)
    # contig a: sites around the window starts the cuts will fall on; contig b: a chain of tandem copies over a cut; c: plain; d: tiny
    la, lb, lc = 61000, 45000, 30000
    sites_a = [(int(p), int(rng.integers(0, 4)), bool(rng.integers(0, 2))) for p in rng.integers(200, la - 200, size=40)]
    ca = planted(rng, la, sites_a)
    cb = bytearray(planted(rng, lb, [(int(p), int(rng.integers(0, 4)), bool(rng.integers(0, 2))) for p in rng.integers(200, lb - 200, size=20)]).encode())
    unit = b"CTTGCCCCACAGGGCAGTAATGG"
    for k in range(12):
        cb[22 * step - 70 + 9 * k: 22 * step - 70 + 9 * k + len(unit)] = unit
    cc = bytearray(planted(rng, lc, [(int(p), 1, False) for p in rng.integers(200, lc - 200, size=10)]).encode())
    for k in range(110):                                       # a crowded bin without a chain: 110 separate copies of the site, 30 bases apart
        cc[12000 + 30 * k: 12000 + 30 * k + len(unit)] = unit
    cc = cc.decode()
    fa = write_fasta(str(tmp_path / "ranges.fa"), [("a", ca), ("b", cb.decode()), ("c", cc), ("d", "ACGT" * 20)])
    lengths = [la, lb, lc, 80]
    n_win = sum(shard.window_counts(lengths, step))
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        G = C.Guide(GUIDE)
        pk = dict(max_gaps_between_guide_and_pam=2)
        whole, n_whole = ctx.search_hits(G, "a", C.make_params(**pk), "v0", "stamp")
        # cut points: equal parts, one moved onto the chain of contig b, one onto a contig boundary
        wa = shard.window_counts(lengths, step)[0]
        bounds = sorted(set([0, n_win] + [n_win * i // cuts for i in range(1, cuts)] + ([wa + 22] if cuts > 2 else []) + ([wa] if cuts > 3 else [])))
        seen_own_general = 0
        for mode in ("default", "general", "two lanes", "three lanes", "whole contigs"):
            if mode == "whole contigs":                          # round 3's answer to a crowded bin of a stretch: the touched contigs searched whole
                monkeypatch.setenv("CALITAS_OWN_GENERAL_OFF", "1")
            if mode == "general":
                monkeypatch.setenv("CALITAS_BINNED", "0")        # the fallback: whole contigs on the general kernels, rows filtered by position
            if mode.endswith("lanes"):                           # the range cut once more into pipelined pieces, as a rank's share of a large genome is
                monkeypatch.setenv("CALITAS_CHUNKS", "2" if mode.startswith("two") else "5:3:2")
            pieces, rows = [], 0
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                text, n = ctx.search_hits(G, "a", C.make_params(first_window=lo, n_windows=hi - lo, **pk), "v0", "stamp")
                head, _, body = text.partition("\n")
                assert head + "\n" == whole[:len(head) + 1]
                pieces.append(body); rows += n
                tm = ctx.timing()
                if mode == "default":
                    seen_own_general += tm["owned_general_lanes"]
                if mode == "whole contigs":
                    assert tm["owned_general_lanes"] == 0
            monkeypatch.delenv("CALITAS_OWN_GENERAL_OFF", raising=False)
            monkeypatch.delenv("CALITAS_BINNED", raising=False)
            monkeypatch.delenv("CALITAS_CHUNKS", raising=False)
            assert rows == n_whole and whole == whole[:whole.index("\n") + 1] + "".join(pieces), (mode, bounds)
        # the stretch that holds the crowded bin of contig c was finished by the general kernels from the bins' alignments, owned rows only
        assert seen_own_general >= 1
        # calitas_search_hits_batch on a window range (BASELINE config 4 on several GPUs: every process runs all guides on its stretch):
        # per guide the text of calitas_search_hits on the same range -- on the per-bin kernels and, where the bins decline a stretch (the
        # chain of tandem copies of contig b lies on a cut), through the per-guide fallback
        G3 = [G, C.Guide("GTGACTTGAAGTCTCAGTATnrg"), G]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            pr = C.make_params(first_window=lo, n_windows=hi - lo, **pk)
            got = ctx.search_hits_batch(G3, ["a", "b", "a"], pr, "v0", "stamp")
            for g, gid, (text, n) in zip(G3, ["a", "b", "a"], got):
                assert (text, n) == ctx.search_hits(g, gid, pr, "v0", "stamp"), (lo, hi, gid)
    finally:
        ctx.close()
    _, want, _ = O.search_reference(fa, GUIDE, "a", g=2, threads=4)
    assert n_whole > 60 and strip(C.read_hits(whole)) == strip(want)


def test_a_rank_that_holds_only_the_contigs_of_its_window_range(C, tmp_path):
    """bench.py --shard windows, ranks > 0: calitas_set_reference with the contigs the rank's window range does not touch given without
    bases (names and lengths of the whole dictionary: windowIterator's sequence, coordinates and order stay the job's).  The rows a
    range owns are the rows it owns on a context that holds everything; a search that would need an absent contig is refused."""
    from calitas_amd import shard
    rng = np.random.default_rng(77)
    lens = [52000, 47000, 61000, 30000]
    seqs = [planted(rng, n, [(int(p), int(rng.integers(0, 4)), bool(rng.integers(0, 2))) for p in rng.integers(200, n - 200, size=25)]).encode() for n in lens]
    names = ["a", "b", "c", "d"]
    step = 1000 - (len(GUIDE) + 5 + 2 - 1)
    G = C.Guide(GUIDE)
    pk = dict(max_gaps_between_guide_and_pam=2)
    full = C.Context(0)
    full.set_reference(names, seqs)
    try:
        whole, n_whole = full.search_hits(G, "a", C.make_params(**pk), "v0", "stamp")
        pieces, rows = [], 0
        for first, n in shard.window_partition(lens, 3, step):
            held = shard.resident_contigs(lens, step, first, n)
            assert 0 < len(held) < len(lens)
            part = C.Context(0)
            part.set_reference(names, [s if i in held else None for i, s in enumerate(seqs)], lengths=lens)
            try:
                assert part.reference_info()["packed_bytes"] == (sum(lens[i] for i in held) + 3) // 4
                pr = C.make_params(first_window=first, n_windows=n, **pk)
                text, k = part.search_hits(G, "a", pr, "v0", "stamp")
                assert (text, k) == full.search_hits(G, "a", pr, "v0", "stamp")
                pieces.append(text.partition("\n")[2]); rows += k
                with pytest.raises(C.CalitasError, match="not resident"):
                    part.search_hits(G, "a", C.make_params(**pk), "v0", "stamp")
                gone = [i for i in range(len(lens)) if i not in held][0]
                w0 = sum(shard.window_counts(lens, step)[:gone])
                with pytest.raises(C.CalitasError, match="not resident"):
                    part.search_hits(G, "a", C.make_params(first_window=w0, n_windows=2, **pk), "v0", "stamp")
            finally:
                part.close()
        assert rows == n_whole and n_whole > 40 and whole == whole[:whole.index("\n") + 1] + "".join(pieces)
    finally:
        full.close()
