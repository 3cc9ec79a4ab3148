"""BASELINE config 3 at its full size (synthetic hg38-sized genome, 25 contigs, 3.09 Gb, seed 0xC3 -- the bench workload):
what cannot be compared row by row against the CPU oracle in seconds is checked through properties, and four whole
chromosomes are compared row by row."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP_COLS = {"aligner_version", "time_stamp"}


@pytest.fixture(scope="module")
def world():
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import calitas_amd as C
    names, seqs = bench.build_genome(1.0, torch.device("cuda", 0), contig_indices=None, guides=[bench.GUIDE0], log=None)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs, genome_build="synthetic-hg38-sized")
    yield C, ctx, names, seqs, bench.GUIDE0
    ctx.close()


def test_hg38_size_paths_agree_and_rows_are_ordered(world, monkeypatch):
    C, ctx, names, seqs, guide = world
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(guide)
    monkeypatch.delenv("CALITAS_CHUNKS", raising=False)
    lanes, n = ctx.search_hits(G, "a", params, "v0", "stamp")
    assert ctx.timing()["lanes"] == 3                              # the default cut of a reference this large
    monkeypatch.setenv("CALITAS_CHUNKS", "1")
    one, n1 = ctx.search_hits(G, "a", params, "v0", "stamp")
    monkeypatch.setenv("CALITAS_CHUNKS", "3:2:2:1")
    four, n4 = ctx.search_hits(G, "a", params, "v0", "stamp")
    monkeypatch.delenv("CALITAS_CHUNKS")
    out, k = ctx.search_raw([G], params)
    try:
        two_stage, n2 = ctx.hits_tsv_raw(G, "a", params, out, k, "v0", "stamp")
    finally:
        C._lib.lib.calitas_free(out)
    assert n > 50000 and (n, lanes) == (n1, one) == (n4, four) == (n2, two_stage)
    batch = ctx.search_hits_batch([G, G], ["a", "a"], params, "v0", "stamp")
    assert batch == [(lanes, n), (lanes, n)]
    # ReferenceHit.sort (RH:284) and removeOverlaps (SR:661-671) as properties of the output
    order = {nm: i for i, nm in enumerate(names)}
    rows = C.read_hits(lanes)
    keys = [(order[r["chromosome"]], int(r["coordinate_start"]), r["strand"], -int(r["score"])) for r in rows]
    assert keys == sorted(keys)
    last = {}
    for r in rows:   # consecutive kept hits of one (chromosome, strand) group overlap by less than maxOverlap
        g = (r["chromosome"], r["strand"])
        s = int(r["coordinate_start"])
        e = s + sum(1 for c in r["padded_target"] if c != "-") - 1
        if g in last:
            assert min(e, last[g][1]) - max(s, last[g][0]) < 10, r
        last[g] = (s, e)
    # planted perfect sites of the bench genome are all reported (exact matches survive every filter)
    perfect = sum(1 for r in rows if r["total_mm_plus_gaps"] == "0")
    assert perfect >= 3


def test_hg38_size_window_partition_concatenates_to_the_whole_text(world, monkeypatch):
    """The multi-GPU partition at BASELINE's size on one card: windowIterator's 3.18 M windows in 8 equal consecutive ranges
    (shard.window_partition: the cuts fall inside chromosomes), every range through calitas_search_hits on its window range (the rows whose
    coordinate_start lies in its stretch); the eight texts, concatenated, are the single-call text byte for byte -- with the per-bin
    kernels deciding each stretch from its own bins plus a halo, and with the whole-contig fallback."""
    from calitas_amd import shard
    C, ctx, names, seqs, guide = world
    kw = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = C.Guide(guide)
    whole, n = ctx.search_hits(G, "a", C.make_params(**kw), "v0", "stamp", decode="bytes")
    lengths = [len(s) for s in seqs]
    parts = shard.window_partition(lengths, 8, 971)
    assert sum(k for _, k in parts) == sum(shard.window_counts(lengths, 971)) > 3_000_000
    head = whole[:whole.index(b"\n") + 1]
    pieces, rows = [], 0
    for first, count in parts:
        text, k = ctx.search_hits(G, "a", C.make_params(first_window=first, n_windows=count, **kw), "v0", "stamp", decode="bytes")
        assert text.startswith(head)
        pieces.append(text[len(head):]); rows += k
        tm = ctx.timing()
        assert tm["lanes"] == 2 and tm["binned_lanes"] == 2          # a 386-Mb stretch runs as two pipelined pieces, both decided by the per-bin kernels
    assert rows == n and head + b"".join(pieces) == whole
    # the two halves a rank of two searches (1.5 Gb each: two pieces as well)
    halves = shard.window_partition(lengths, 2, 971)
    both = []
    for first, count in halves:
        text, k = ctx.search_hits(G, "a", C.make_params(first_window=first, n_windows=count, **kw), "v0", "stamp", decode="bytes")
        both.append(text[len(head):])
        assert ctx.timing()["binned_lanes"] == ctx.timing()["lanes"] == 2
    assert head + b"".join(both) == whole
    # two of the stretches again on the fallback (the touched contigs searched whole on the general kernels, rows filtered by position)
    monkeypatch.setenv("CALITAS_BINNED", "0")
    for i in (2, 7):
        first, count = parts[i]
        text, k = ctx.search_hits(G, "a", C.make_params(first_window=first, n_windows=count, **kw), "v0", "stamp", decode="bytes")
        assert text[len(head):] == pieces[i]


def test_hg38_size_whole_chromosomes_against_the_oracle(world):
    """chr19-chr22 and chrM of the full-size run (about 220 Mb), every column, against the oracle run on those contigs alone:
    windows, removeOverlaps groups and the sort never cross a contig, so the rows must be identical."""
    C, ctx, names, seqs, guide = world
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    text, _ = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
    rows = C.read_hits(text)
    pick = [i for i, nm in enumerate(names) if nm in ("chr19", "chr20", "chr21", "chr22", "chrM")]
    _, orows, _ = O.search_memory([names[i] for i in pick], [seqs[i].tobytes() for i in pick], guide, "a", d=5, p=1, g=2, threads=16)
    want = [{k: v for k, v in r.items() if k not in SKIP_COLS} for r in orows]
    chosen = {names[i] for i in pick}
    got = [{k: v for k, v in r.items() if k not in SKIP_COLS} for r in rows if r["chromosome"] in chosen]
    for r in want:
        r["genome_build"] = "synthetic-hg38-sized"                 # the oracle has no .dict for in-memory contigs
    assert len(want) > 3000 and got == want


def test_hg38_size_96_guide_batch_equals_single_calls(world):
    """BASELINE config 4 at its stated size: guide #0 + 95 random 20-mers (seed 0xC4) against the 3.09 Gb genome in one
    calitas_search_hits_batch call.  Every guide's text is held by size, row count and CRC; a sample of guides is compared byte for byte
    with its own calitas_search_hits call, and guide #0's rows are the ones the other full-size tests hold against the oracle."""
    import zlib
    from calitas_amd import synth
    C, ctx, names, seqs, guide = world
    guides = [guide] + synth.random_guides(0xC4, 95)
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    G = [C.Guide(g) for g in guides]
    ids = ["g%02d" % i for i in range(96)]
    res = ctx.search_hits_batch(G, ids, params, "v0", "stamp", decode="digest")
    assert len(res) == 96 and all(rows > 10000 for _, rows in res)
    for i in (0, 1, 37, 95):
        text, rows = ctx.search_hits(G[i], ids[i], params, "v0", "stamp", decode="bytes")
        assert (zlib.crc32(text), len(text)) == res[i][0] and rows == res[i][1], i
