"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs -- identical hits.txt rows
(every column except the run-dependent aligner_version / time_stamp), bit-exact.

Run on the GPU box with:  python -m pytest tests -m gpu
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from fasta_util import expand, write_fasta

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SKIP_COLS = {"aligner_version", "time_stamp"}


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


PATHS = []   # binned_lanes of every fused call of product_rows (which tail ran): see test_most_searches_take_the_binned_tail


def product_rows(C, fasta, guide, aux=(), chrom=None, **kw):
    pk = dict(window_size=kw.get("window_size", 1000), max_guide_diffs=kw.get("d", 5), max_pam_mismatches=kw.get("p", 1),
              max_gaps_between_guide_and_pam=kw.get("g", 3), max_total_diffs=kw.get("D"), max_overlap=kw.get("O", 10),
              eqx_by_score=(1 if kw.get("switches", 0) & 2 else 0) | (2 if kw.get("switches", 0) & 1 else 0))   # oracle bits -> ABI bits
    for k in ("guide_mismatch_net_cost", "pam_mismatch_net_cost", "genome_gap_net_cost", "guide_gap_net_cost"):
        if k in kw:
            pk[k] = kw[k]
    # the fused call (calitas_search_hits: filter, removeOverlaps, sorts and rows on the device) and the two-stage call
    # (calitas_search + calitas_hits_tsv: the same stages on the host) must agree byte for byte
    ctx = C.Context(0)
    ctx.set_reference_fasta(fasta)
    try:
        text, n = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, **pk).run("v0", "stamp")
        PATHS.append(ctx.timing()["binned_lanes"])
        text2, n2 = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, two_stage=True, **pk).run("v0", "stamp")
        # ... and so must the general device kernels (select.hip / hits.hip), which the per-bin kernels (binned.hip) stand in front of
        os.environ["CALITAS_BINNED"] = "0"
        try:
            text3, n3 = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, **pk).run("v0", "stamp")
            assert ctx.timing()["binned_lanes"] == 0
        finally:
            del os.environ["CALITAS_BINNED"]
        os.environ["CALITAS_BINNED_COMPLEX"] = "1"     # binned.hip's wave-per-bin kernel for every bin (by default: the crowded ones)
        try:
            text4, _ = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, **pk).run("v0", "stamp")
        finally:
            del os.environ["CALITAS_BINNED_COMPLEX"]
        assert text4 == text
        # align_kernel packs three jobs of 21 lanes into a wave for guides of up to 20 rows; two jobs of 32 lanes must give the same bytes
        os.environ["CALITAS_ALIGN_LPJ"] = "32"
        try:
            text5, _ = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, **pk).run("v0", "stamp")
        finally:
            del os.environ["CALITAS_ALIGN_LPJ"]
        assert text5 == text, "align_kernel with two and with three jobs per wave differ"
        # one job per lane group (align_kernel) against two in sixteen-bit halves (align_pk_kernel, the default where the cells fit)
        os.environ["CALITAS_ALIGN_PACK"] = "0"
        try:
            text7, _ = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, **pk).run("v0", "stamp")
        finally:
            del os.environ["CALITAS_ALIGN_PACK"]
        assert text7 == text, "align_kernel and align_pk_kernel differ"
        # the lane's small inputs as separate stream commands instead of the one setup launch (kernels.hpp, LaneSetupArgs)
        os.environ["CALITAS_LANE_SETUP"] = "0"
        try:
            text6, _ = C.SearchReference(guide=guide, guide_id="a", context=ctx, auxiliary_pams=aux, chrom=chrom, **pk).run("v0", "stamp")
        finally:
            del os.environ["CALITAS_LANE_SETUP"]
        assert text6 == text, "setup launch and separate input commands differ"
        tm = ctx.timing()
        assert tm["scan_kernel_ms"] > 0 and tm["align_kernel_ms"] > 0 and tm["gpu_total_ms"] >= tm["align_kernel_ms"]   # (stamps or events)
    finally:
        ctx.close()
    assert n == n2 == n3
    assert text3 == text, "binned and general device kernels differ"
    if text != text2:
        a, b = text.splitlines(), text2.splitlines()
        diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y][:2]
        raise AssertionError("fused and two-stage hits.txt differ: %d vs %d lines, first differences: %s" % (len(a), len(b), diff))
    rows = C.read_hits(text)
    assert len(rows) == n
    return rows


def oracle_rows(fasta, guide, aux=(), chrom=None, **kw):
    ok = dict(window_size=kw.get("window_size", 1000), d=kw.get("d", 5), p=kw.get("p", 1), g=kw.get("g", 3),
              D=-1 if kw.get("D") is None else kw["D"], O=kw.get("O", 10), switches=kw.get("switches", 0), threads=4)
    if "guide_mismatch_net_cost" in kw: ok["m"] = kw["guide_mismatch_net_cost"]
    if "pam_mismatch_net_cost" in kw: ok["M"] = kw["pam_mismatch_net_cost"]
    if "genome_gap_net_cost" in kw: ok["b"] = kw["genome_gap_net_cost"]
    if "guide_gap_net_cost" in kw: ok["B"] = kw["guide_gap_net_cost"]
    _, rows, _ = O.search_reference(fasta, guide, "a", aux=aux, chrom=chrom or "", **ok)
    return rows


def assert_same(prod, orac, tag=""):
    def strip(rows):
        return [{k: v for k, v in r.items() if k not in SKIP_COLS} for r in rows]
    p, o = strip(prod), strip(orac)
    if p != o:
        ps = {json.dumps(r, sort_keys=True) for r in p}
        os_ = {json.dumps(r, sort_keys=True) for r in o}
        only_p = [json.loads(x) for x in sorted(ps - os_)][:3]
        only_o = [json.loads(x) for x in sorted(os_ - ps)][:3]
        raise AssertionError("%s: product %d rows, oracle %d rows\nonly product: %s\nonly oracle: %s" % (tag, len(p), len(o), only_p, only_o))


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own end-to-end vectors (SearchReferenceTest.scala:51-92), now through the GPU path
# ---------------------------------------------------------------------------------------------------------------------
SR = json.load(open(os.path.join(GOLD, "kat_sr.json")))


@pytest.mark.parametrize("case", SR["cases"], ids=lambda c: c["id"])
def test_reference_end_to_end_vectors(C, case, tmp_path):
    contigs = [(name, expand(spec)) for name, spec in SR[case["fasta"]]["contigs"]]
    fa = write_fasta(str(tmp_path / "kat.fa"), contigs)
    out = str(tmp_path / "hits.txt")
    C.SearchReference(guide=case["guide"], guide_id="a", ref=fa, output=out, threads=1).execute()
    hits = C.read_hits(out)
    e = case["expect"]
    assert len(hits) == e["n"]
    for k in ("chromosome", "padded_alignment"):
        if k in e:
            assert [h[k] for h in hits] == e[k]
    for k in ("coordinate_start", "total_mm_plus_gaps"):
        if k in e:
            assert [int(h[k]) for h in hits] == e[k]
    assert_same(hits, oracle_rows(fa, case["guide"]), case["id"])


# ---------------------------------------------------------------------------------------------------------------------
# seeded synthetic genomes: planted sites, N runs, soft-masking, tandem repeats, IUPAC codes, short contigs
# ---------------------------------------------------------------------------------------------------------------------
def synth_fasta(tmp_path, seed, guides, lengths=(60000, 35000, 1500, 700, 26, 12), extra=None, **kw):
    from calitas_amd import synth
    spec = [("ctg%d" % i, l) for i, l in enumerate(lengths)]
    glist = []
    for g in guides:
        G = __import__("calitas_amd").Guide(g)
        pam = G.pams[0] if G.pams else ""
        glist.append((G.guide, pam, G.pam_is_five_prime))
    names, seqs = synth.make_genome(spec, seed, guides=glist, sites_per_guide=60, softmask=0.4, tandem_frac=0.03,
                                    n_run_ends=kw.get("n_run_ends", 300), n_block=kw.get("n_block", 2500), step_hint=kw.get("step_hint", 971))
    contigs = [(n, s.tobytes().decode()) for n, s in zip(names, seqs)]
    if extra:
        contigs += extra
    return write_fasta(str(tmp_path / ("synth%d.fa" % seed)), contigs)


CONFIGS = [
    # (id, guide, aux, params)
    ("nrg-d5-g2", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=5, p=1, g=2)),                     # BASELINE config 3 limits
    ("nrg-d3", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=3, p=1, g=3)),                        # BASELINE config 2 limits
    ("pamless-d5", "GTGACTTGAAGTCTCAGTATA", (), dict(d=5)),
    ("5prime-tttv", "tttvAACCAACCAACCGGTTACGT", (), dict(d=4, p=1, g=2)),
    ("aux-pams", "ACGTACATGCTCGATACGACGnngrrn", ("nngrrt", "nnagaaw"), dict(d=4, p=1, g=3)),
    ("iupac-protospacer", "GAGAATTGNTTGAACCCRGG", (), dict(d=3)),
    ("d0", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=0, p=0, g=0)),
    ("wide-overlap", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=5, p=1, g=3, O=100)),
    ("zero-overlap", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=4, p=1, g=1, O=0)),
    ("tight-D", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=5, p=1, g=3, D=4)),
    ("small-window", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=4, p=1, g=2, window_size=120)),
    ("eqx-by-score", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=5, p=1, g=2, switches=2)),
    ("costs", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=4, p=1, g=2, guide_mismatch_net_cost=-100, pam_mismatch_net_cost=-200,
                                                  genome_gap_net_cost=-104, guide_gap_net_cost=-102)),
    # costs at the edge of what align_pk_kernel's sixteen-bit cells hold (4 x max|cost| x (L + 2) < 30 000: 330 -> 29 040) and beyond it
    # (400 -> 35 200: the search takes align_kernel): both against the oracle
    ("costs-16bit-edge", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=4, p=1, g=2, guide_mismatch_net_cost=-660, pam_mismatch_net_cost=-700,
                                                             genome_gap_net_cost=-650, guide_gap_net_cost=-640)),
    ("costs-beyond-16bit", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=4, p=1, g=2, guide_mismatch_net_cost=-800, pam_mismatch_net_cost=-900,
                                                               genome_gap_net_cost=-820, guide_gap_net_cost=-810)),
    ("per-matrix", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=5, p=1, g=2, switches=1)),
    ("per-matrix-wide-overlap", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=5, p=1, g=3, O=100, switches=1)),
    ("per-matrix-eqx-pamless", "GTGACTTGAAGTCTCAGTATA", (), dict(d=5, O=40, switches=3)),
    ("per-matrix-5prime", "tttvAACCAACCAACCGGTTACGT", (), dict(d=4, p=1, g=2, O=60, switches=1)),
    ("short-guide-12", "GCAGTAACCTGAnrg", (), dict(d=2, p=1, g=1)),
    ("long-guide-32", "CTTGCCCCACAGGGCAGTAACGGTTCAATGCA", (), dict(d=6)),               # the scan's 32 rows, PAM-less
    ("long-guide-31-pam", "CTTGCCCCACAGGGCAGTAACGGTTCAATGCngg", (), dict(d=5, p=1, g=2)),   # 31 rows: lane 31 of a half wave is no row (align_kernel)
    ("long-pam-8", "CTTGCCCCACAGGGCAGTAAnnagaawn", (), dict(d=4, p=2, g=2)),
    ("three-strands-of-limits", "CTTGCCCCACAGGGCAGTAAnrg", (), dict(d=6, p=3, g=4, D=7, O=1)),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: c[0])
def test_synthetic_genome_parity(C, cfg, tmp_path):
    cid, guide, aux, params = cfg
    W = params.get("window_size", 1000)
    step = W - (len(guide) + params.get("d", 5) + params.get("g", 3) - 1)
    extra = [("iupac", "ACGTRYKMSWBDHVN" * 40 + "CTTGCCCCACAGGGCAGTAATGG" + "acgtn" * 30 + "CTTGCCCCACNGGGCAGTAAAGG" + "TTRACGGT" * 20),
             ("rna", "ACGUACGUUUGGCAUCG" * 30 + "CUUGCCCCACAGGGCAGUAAUGG" + "ACGU" * 20)]
    fa = synth_fasta(tmp_path, 7 + len(cid), [guide], extra=extra, step_hint=step)
    prod = product_rows(C, fa, guide, aux, **params)
    orac = oracle_rows(fa, guide, aux, **params)
    assert len(orac) > 0
    assert_same(prod, orac, cid)


def test_chrom_filter_and_reuse(C, tmp_path):
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    fa = synth_fasta(tmp_path, 99, [guide])
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    for chrom in ("ctg1", "ctg0", None):
        sr = C.SearchReference(guide=guide, guide_id="a", context=ctx, chrom=chrom, max_gaps_between_guide_and_pam=2)
        text, _ = sr.run()
        assert_same(C.read_hits(text), oracle_rows(fa, guide, chrom=chrom, g=2), "chrom=%s" % chrom)
    ctx.close()


def test_guide_batch_matches_single_guide_runs(C, tmp_path):
    """A multi-guide pass (BASELINE config 4 shape) returns, guide by guide, what single-guide passes return."""
    from calitas_amd import synth
    guides = ["CTTGCCCCACAGGGCAGTAAnrg"] + synth.random_guides(0xC4, 5)
    fa = synth_fasta(tmp_path, 5, guides, lengths=(50000, 20000))
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    params = C.make_params(max_gaps_between_guide_and_pam=2)
    G = [C.Guide(g) for g in guides]
    alns = ctx.search(G, params)
    for gi, g in enumerate(G):
        mine = [a for a in alns if a.guide_index == gi]
        text, _ = ctx.hits_tsv(g, "a", params, mine)
        assert_same(C.read_hits(text), oracle_rows(fa, guides[gi], g=2), "guide %d" % gi)
    ctx.close()


def test_no_silent_truncation_on_dense_output(C, tmp_path):
    """PAM-less d=8 on a short tandem-repeat contig floods the candidate buffers; the retry path must deliver everything."""
    unit = "ACGTTGCA"
    seq = (unit * 4000) + "GTGACTTGAAGTCTCAGTATA" + (unit[::-1] * 3000)
    fa = write_fasta(str(tmp_path / "dense.fa"), [("rep", seq)])
    guide = "ACGTTGCAACGTTGCAACGT"
    prod = product_rows(C, fa, guide, d=8, O=100)
    orac = oracle_rows(fa, guide, d=8, O=100)
    assert_same(prod, orac, "dense")


def test_small_input_with_a_crowded_window(C, tmp_path):
    """Fewer than 1024 raw alignments take the one-workgroup versions of the filter and hits stages (select_small_kernel,
    hits_small_kernel); a window with more than 32 of them makes the filter hand the call back to the general kernels."""
    unit = "ACGTTGCA"
    seq = "T" * 300 + unit * 14 + "GTGACTTGAAGTCTCAGTATA" + "C" * 400 + "CTTGCCCCACAGGGCAGTAATGG" + "A" * 200
    fa = write_fasta(str(tmp_path / "crowded_small.fa"), [("c1", seq), ("c2", "G" * 150 + "CTTGCCCCACAGGGCAGTTATGG" + "T" * 90)])
    for guide, kw in (("ACGTTGCAACGTTGCAACGT", dict(d=6, O=100)), ("CTTGCCCCACAGGGCAGTAAnrg", dict(d=5, p=1, g=2))):
        prod = product_rows(C, fa, guide, **kw)
        orac = oracle_rows(fa, guide, **kw)
        assert_same(prod, orac, guide)


def test_device_filter_equals_host_filter(C, tmp_path, monkeypatch):
    """The per-window overlap filter (SGA:316-331) runs on the device by default and on the host when a window holds more
    alignments than one lane handles, or on request; both must return the same records in the same order."""
    guides = ["CTTGCCCCACAGGGCAGTAAnrg", "tttvAACCAACCAACCGGTTACGT"]
    fa = synth_fasta(tmp_path, 21, guides, lengths=(80000, 30000))
    unit = "ACGTTGCA"
    dense = write_fasta(str(tmp_path / "dense2.fa"), [("rep", unit * 3000 + "CTTGCCCCACAGGGCAGTAATGG" + unit * 500)])

    def run(path, guide, **kw):
        ctx = C.Context(0)
        ctx.set_reference_fasta(path)
        al = ctx.search([C.Guide(guide)], C.make_params(**kw))
        ctx.close()
        return [(a.contig_index, a.window_start, a.strand, a.start_offset, a.end_offset, a.score, a.ops, a.pam_index, a.guide_start_offset, a.guide_end_offset) for a in al]

    cases = [(fa, guides[0], dict(max_gaps_between_guide_and_pam=2)), (fa, guides[1], dict(max_guide_diffs=4)),
             (dense, "ACGTTGCAACGTTGCAACGT", dict(max_guide_diffs=6, max_overlap=3))]   # > 256 alignments per window: host fallback
    for path, guide, kw in cases:
        monkeypatch.delenv("CALITAS_HOST_FILTER", raising=False)
        dev = run(path, guide, **kw)
        monkeypatch.setenv("CALITAS_HOST_FILTER", "1")
        host = run(path, guide, **kw)
        assert len(dev) > 0 and dev == host, guide


@pytest.mark.parametrize("chunks", ["2", "3", "4", "5:1", "1:1:6"])
def test_chunked_search_hits_equals_one_pass(C, tmp_path, monkeypatch, chunks):
    """calitas_search_hits cuts a large reference into contig ranges that flow through the stages as a pipeline (one lane
    each); the text must not depend on how the reference is cut.  CALITAS_CHUNKS forces the cut on a small reference."""
    guides = ["CTTGCCCCACAGGGCAGTAAnrg", "tttvAACCAACCAACCGGTTACGT"]
    extra = [("tiny", "ACGTTGCA" * 6 + "CTTGCCCCACAGGGCAGTAATGG" + "TTGACA" * 5), ("empty-ish", "N" * 300),
             ("rna", "ACGUACGUUUGGCAUCG" * 30 + "CUUGCCCCACAGGGCAGUAAUGG" + "ACGU" * 20)]
    fa = synth_fasta(tmp_path, 31, guides, lengths=(60000, 25000, 90000, 12000), extra=extra)
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        for guide, kw in ((guides[0], dict(max_gaps_between_guide_and_pam=2)), (guides[1], dict(max_guide_diffs=4))):
            params = C.make_params(**kw)
            monkeypatch.delenv("CALITAS_CHUNKS", raising=False)
            one, n1 = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            assert ctx.timing()["lanes"] == 1
            monkeypatch.setenv("CALITAS_CHUNKS", chunks)
            cut, n2 = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            assert ctx.timing()["lanes"] > 1                      # number of lanes the call used
            again, n3 = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            assert n1 > 20 and (n1, one) == (n2, cut) == (n3, again)
    finally:
        ctx.close()


def test_first_call_of_fresh_lanes(C, tmp_path, monkeypatch):
    """The first chunked call of a fresh context is the one that creates and clears every lane's scratch.  That clearing has to
    be ordered on the lane's own (non-blocking) stream: a null-stream hipMemset once let count_kernel run ahead of it, which
    regrouped records at random and could index out of bounds.  The guard is deterministic now: select.hip fills a new counter
    buffer with 0xFF on the using stream before it clears it, and scatter_kernel reports any slot outside [0, n) -- a clear that
    is missing or mis-ordered fails the very first call with CALITAS_EHIP instead of depending on what the allocator handed out
    (so two fresh contexts are enough here).  PAM-less, d = 6 on a 14-mer: every window is crowded."""
    guide = "ACATTCGTCAGTCG"
    fa = synth_fasta(tmp_path, 71, [guide + "nrg"], lengths=(24000, 20000))
    params = C.make_params(window_size=1000, max_guide_diffs=6, max_pam_mismatches=2, max_gaps_between_guide_and_pam=4, max_overlap=29)
    want = None
    for _ in range(2):
        ctx = C.Context(0)
        ctx.set_reference_fasta(fa)
        try:
            monkeypatch.setenv("CALITAS_CHUNKS", "2")
            got = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            if want is None:
                monkeypatch.setenv("CALITAS_CHUNKS", "1")
                want = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
                assert want[1] > 1000
            assert got == want
        finally:
            ctx.close()


def test_host_tail_of_search_hits(C, tmp_path, monkeypatch):
    """When a device stage of calitas_search_hits declines (score range, cluster length, -O 0) the host implementation of that
    stage finishes the call -- in one pass or lane by lane.  CALITAS_HOST_HITS forces that path; -O 0 takes it by itself."""
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    fa = synth_fasta(tmp_path, 43, [guide], lengths=(50000, 30000, 8000))
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        params = C.make_params(max_gaps_between_guide_and_pam=2)
        want, n = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
        for chunks in ("1", "3"):
            monkeypatch.setenv("CALITAS_CHUNKS", chunks)
            monkeypatch.setenv("CALITAS_HOST_HITS", "1")
            got, n2 = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            monkeypatch.delenv("CALITAS_HOST_HITS")
            assert n > 20 and (n, want) == (n2, got), chunks
            monkeypatch.setenv("CALITAS_HOST_FILTER", "1")        # the per-window filter on the host as well, in every lane
            got, n2 = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            monkeypatch.delenv("CALITAS_HOST_FILTER")
            assert (n, want) == (n2, got), chunks
            p0 = C.make_params(max_gaps_between_guide_and_pam=2, max_overlap=0)        # removeOverlaps with -O 0: host stage
            a, na = ctx.search_hits(C.Guide(guide), "a", p0, "v0", "stamp")
            out, k = ctx.search_raw([C.Guide(guide)], p0)
            try:
                b, nb = ctx.hits_tsv_raw(C.Guide(guide), "a", p0, out, k, "v0", "stamp")
            finally:
                C._lib.lib.calitas_free(out)
            assert (na, a) == (nb, b)
    finally:
        ctx.close()


def test_search_without_hits_and_degenerate_references(C, tmp_path, monkeypatch):
    """No candidate anywhere, contigs shorter than the guide, all-N contigs: the header alone comes back, on every path."""
    cases = [("nohit", [("c0", "ACGT" * 500), ("c1", "TTTTGGGGCCCCAAAA" * 40)], "GGATCCGAATTCAAGCTTCCnrg", dict(max_guide_diffs=0, max_pam_mismatches=0)),
             ("short", [("s0", "ACGTACGTAC"), ("s1", "G" * 22)], "CTTGCCCCACAGGGCAGTAAnrg", dict()),
             ("alln", [("n0", "N" * 5000), ("n1", "n" * 300)], "CTTGCCCCACAGGGCAGTAAnrg", dict())]
    for tag, contigs, guide, kw in cases:
        fa = write_fasta(str(tmp_path / (tag + ".fa")), contigs)
        _, orows, _ = O.search_reference(fa, guide, "a", d=kw.get("max_guide_diffs", 5), p=kw.get("max_pam_mismatches", 1))
        ctx = C.Context(0)
        ctx.set_reference_fasta(fa)
        try:
            params = C.make_params(**kw)
            for chunks in (None, "2"):
                if chunks:
                    monkeypatch.setenv("CALITAS_CHUNKS", chunks)
                else:
                    monkeypatch.delenv("CALITAS_CHUNKS", raising=False)
                text, n = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
                rows = C.read_hits(text)
                assert n == len(rows) == len(orows), (tag, chunks)
                assert text.splitlines()[0].split("\t")[0] == "guide_id" and len(text.splitlines()[0].split("\t")) == 34
                if orows:
                    assert_same(rows, orows, tag)
            assert ctx.search([C.Guide(guide)], params) == [] or orows
        finally:
            ctx.close()


def test_search_hits_batch_equals_single_calls(C, tmp_path, monkeypatch):
    """calitas_search_hits_batch pipelines guides through lanes; every guide's text must be what calitas_search_hits returns."""
    from calitas_amd import synth
    guides = ["CTTGCCCCACAGGGCAGTAAnrg"] + synth.random_guides(0xC4, 6)
    fa = synth_fasta(tmp_path, 47, guides, lengths=(50000, 20000, 9000))
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        params = C.make_params(max_gaps_between_guide_and_pam=2)
        G = [C.Guide(g) for g in guides]
        ids = ["g%d" % i for i in range(len(G))]
        single = [ctx.search_hits(g, i, params, "v0", "stamp") for g, i in zip(G, ids)]
        assert sum(n for _, n in single) > 100
        for lanes in ("3", "2", "1", "8"):
            monkeypatch.setenv("CALITAS_BATCH_LANES", lanes)
            assert ctx.search_hits_batch(G, ids, params, "v0", "stamp") == single, lanes
        with pytest.raises(Exception):
            ctx.search_hits_batch(G + [C.Guide("ACGTACGTACGTACGTACGTAAAAnrg")], ids + ["x"], params, "v0", "stamp")   # another length
    finally:
        ctx.close()


def test_text_copy_paths_agree(C, tmp_path, monkeypatch):
    """The finished text comes back over an SDMA engine (HSA runtime, dma.cpp) by default and through hipMemcpyAsync with
    CALITAS_SDMA=0; same bytes either way, in one pass and in lanes."""
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    fa = synth_fasta(tmp_path, 53, [guide], lengths=(70000, 20000))
    params = C.make_params(max_gaps_between_guide_and_pam=2)
    got = {}
    for sdma in ("1", "0"):
        monkeypatch.setenv("CALITAS_SDMA", sdma)
        ctx = C.Context(0)                       # the choice is made once per context
        ctx.set_reference_fasta(fa)
        try:
            for chunks in ("1", "2"):
                monkeypatch.setenv("CALITAS_CHUNKS", chunks)
                got[(sdma, chunks)] = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            got[(sdma, "batch")] = ctx.search_hits_batch([C.Guide(guide)] * 3, ["a"] * 3, params, "v0", "stamp")[2]
        finally:
            ctx.close()
    ref = got[("1", "1")]
    assert ref[1] > 20 and all(v == ref for v in got.values())


def test_cpp_cli_search_reference(C, tmp_path):
    """The `calitas SearchReference` binary with the reference's flags (SearchReference.scala:452-470), FASTA in, hits.txt out."""
    import subprocess
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    fa = synth_fasta(tmp_path, 41, [guide], lengths=(40000, 9000))
    exe = os.path.join(os.path.dirname(os.path.abspath(C.__file__)), "calitas")
    out = tmp_path / "cli_hits.txt"
    r = subprocess.run([exe, "SearchReference", "-i", guide, "-I", "a", "-r", fa, "-o", str(out), "-d", "4", "-p", "1", "-g", "2", "-c", "ctg0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert_same(C.read_hits(out.read_text()), oracle_rows(fa, guide, chrom="ctg0", d=4, p=1, g=2), "cli")
    r = subprocess.run([exe, "SearchReference", "-i", guide, "-I", "a", "-r", fa, "-c", "nope"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "Unknown chromosome" in r.stderr


def test_full_size_properties_ecoli_like(C):
    """BASELINE config 2 size (4.6 Mb): size-independent properties instead of a full oracle run --
    (1) every planted perfect site is found with 0 edits, (2) searching the reverse-complemented genome with the same
    guide flips strands and mirrors coordinates, (3) hits are sorted and non-redundant."""
    from calitas_amd import synth
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    rng = np.random.default_rng(0xC2)
    seq = synth.random_bases(rng, synth.ECOLI_LENGTH, gc=0.508)
    planted = []
    for k in range(50):
        pos = int(rng.integers(100, len(seq) - 100))
        minus = bool(k % 2)
        synth.plant_site(rng, seq, pos, "CTTGCCCCACAGGGCAGTAA", "nrg", False, 0, minus)
        planted.append((pos, minus))
    ctx = C.Context(0)
    ctx.set_reference(["ecoli_like"], [seq])
    params = C.make_params(max_guide_diffs=3)
    G = C.Guide(guide)
    alns = ctx.search([G], params)
    text, n = ctx.hits_tsv(G, "a", params, alns)
    hits = C.read_hits(text)
    perfect = {(int(h["coordinate_start"]), h["strand"]) for h in hits if h["total_mm_plus_gaps"] == "0"}
    for pos, minus in planted:
        start = pos + 3 if minus else pos
        assert (start, "-" if minus else "+") in perfect, (pos, minus)
    keys = [(int(h["coordinate_start"]), h["strand"], -int(h["score"])) for h in hits]
    assert keys == sorted(keys)
    # mirror property
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    rc = comp[seq[::-1]].copy()
    ctx.set_reference(["ecoli_like"], [rc])
    alns2 = ctx.search([G], params)
    text2, _ = ctx.hits_tsv(G, "a", params, alns2)
    hits2 = C.read_hits(text2)
    n_len = len(seq)
    best1 = {(int(h["coordinate_start"]), int(h["coordinate_end"]), h["strand"]) for h in hits if int(h["total_mm_plus_gaps"]) <= 1}
    best2 = {(n_len - int(h["coordinate_end"]), n_len - int(h["coordinate_start"]), "+" if h["strand"] == "-" else "-")
             for h in hits2 if int(h["total_mm_plus_gaps"]) <= 1}
    assert best1 == best2
    ctx.close()


def test_config2_ecoli_size_full_parity(C):
    """BASELINE config 2 at full size (one 4.64 Mb contig, GC 0.508, seed 0xC2, max-guide-diffs 3): every row against the
    oracle (a few seconds of CPU), through calitas_search_hits and through the two-stage path."""
    from calitas_amd import synth
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    rng = np.random.default_rng(0xC2)
    seq = synth.random_bases(rng, synth.ECOLI_LENGTH, gc=0.508)
    for k in range(60):
        synth.plant_site(rng, seq, int(rng.integers(100, len(seq) - 100)), "CTTGCCCCACAGGGCAGTAA", "nrg", False, k % 4, bool(k % 2))
    _, orows, nwin = O.search_memory(["ecoli_like"], [seq.tobytes()], guide, "a", d=3, threads=8)
    assert nwin > 4700 and len(orows) >= 30
    ctx = C.Context(0)
    ctx.set_reference(["ecoli_like"], [seq], genome_build="unknown")
    try:
        params = C.make_params(max_guide_diffs=3)
        text, n = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
        assert_same(C.read_hits(text), orows, "config 2")
        out, k = ctx.search_raw([C.Guide(guide)], params)
        try:
            text2, n2 = ctx.hits_tsv_raw(C.Guide(guide), "a", params, out, k, "v0", "stamp")
        finally:
            C._lib.lib.calitas_free(out)
        assert (n, text) == (n2, text2)
    finally:
        ctx.close()


def test_parity_at_scale_60mb(C):
    """60 Mb of the bench genome's recipe (two contigs, chunk-512 tiles, N blocks, soft-masking, tandem repeats, planted
    sites straddling window starts) -- every row against the oracle (BASELINE config 3 limits)."""
    from calitas_amd import synth
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    names, seqs = synth.make_genome([("chrA", 38_000_000), ("chrB", 22_000_000)], seed=0xC3, guides=[("CTTGCCCCACAGGGCAGTAA", "nrg", False)],
                                    sites_per_guide=400, n_run_ends=10_000, n_block=1_500_000, softmask=0.5, tandem_frac=0.01)
    ctx = C.Context(0)
    ctx.set_reference(names, seqs)
    G = C.Guide(guide)
    params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
    alns = ctx.search([G], params)
    text, n = ctx.hits_tsv(G, "a", params, alns)
    prod = C.read_hits(text)
    _, orac, nwin = O.search_memory(names, [s.tobytes() for s in seqs], guide, "a", d=5, p=1, g=2, threads=16)
    assert nwin > 55000 and len(orac) > 1500
    assert_same(prod, orac, "60mb")
    ctx.close()


def test_per_contig_passes_equal_one_pass(C, tmp_path, monkeypatch):
    """When a search does not fit the device (CALITAS_ENOMEM from an allocation, or the buffers exceed CALITAS_DEVICE_BUDGET_MB)
    calitas_search_hits runs one pass per contig and concatenates the texts; CALITAS_SEQUENTIAL forces that mode.  Same bytes as
    the one-pass call, with and without --chrom, and the context stays usable."""
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    extra = [("tiny", "ACGTTGCA" * 6 + "CTTGCCCCACAGGGCAGTAATGG" + "TTGACA" * 5), ("empty-ish", "N" * 300)]
    fa = synth_fasta(tmp_path, 47, [guide], lengths=(700000, 300000, 200000), extra=extra)
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        for kw in (dict(max_gaps_between_guide_and_pam=2), dict(max_gaps_between_guide_and_pam=2, chrom_index=1)):
            params = C.make_params(**kw)
            monkeypatch.setenv("CALITAS_CHUNKS", "1")
            want = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            assert want[1] > 10
            monkeypatch.setenv("CALITAS_SEQUENTIAL", "1")
            assert ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp") == want       # (the contigs' texts cross PCIe compact: round 5)
            monkeypatch.setenv("CALITAS_COMPACT_ROWS", "0")
            assert ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp") == want       # ... and as whole rows
            # calitas_search_hits_into in the per-contig mode (round 5): every contig's rows straight to their place in the caller's block,
            # whole rows and compact ones; a block that is too small is refused, nothing is written behind its capacity
            import ctypes
            wb = want[0].encode()
            cap = len(wb) + 4096
            addr = C.Context.alloc_host(cap)
            buf = np.ctypeslib.as_array((ctypes.c_uint8 * cap).from_address(addr))
            try:
                for compact in ("0", "1"):
                    monkeypatch.setenv("CALITAS_COMPACT_ROWS", compact)
                    buf[:] = 0x55
                    nb, rows = ctx.search_hits_into(C.Guide(guide), "a", params, addr, cap, "v0", "stamp")[:2]
                    assert (nb, rows) == (len(wb), want[1]) and bytes(buf[:nb]) == wb and buf[nb] == 0 and buf[nb + 1] == 0x55
                    buf[:] = 0x55
                    with pytest.raises(C.CalitasError, match="too small"):
                        ctx.search_hits_into(C.Guide(guide), "a", params, addr, len(wb) // 2, "v0", "stamp")
                    assert (buf[len(wb) // 2:] == 0x55).all()
            finally:
                del buf
                C.Context.free_host(addr)
            monkeypatch.delenv("CALITAS_COMPACT_ROWS")
            monkeypatch.delenv("CALITAS_SEQUENTIAL")
            monkeypatch.setenv("CALITAS_CHUNKS", "2")
            monkeypatch.setenv("CALITAS_DEVICE_BUDGET_MB", "1")          # nothing fits, not even a contig: the call fails with ENOMEM
            with pytest.raises(Exception, match="exceed CALITAS_DEVICE_BUDGET_MB"):
                ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            # the buffers of a pass over everything (1.2 Mb: 1.5e5 records x 2 KB of strip) do not fit, a contig's do
            monkeypatch.setenv("CALITAS_CHUNKS", "1")
            monkeypatch.setenv("CALITAS_DEVICE_BUDGET_MB", "250")
            got = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp")
            assert got == want
            monkeypatch.delenv("CALITAS_DEVICE_BUDGET_MB")
            assert ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp") == want
    finally:
        ctx.close()


def test_search_hits_stream(C, tmp_path, monkeypatch):
    """calitas_search_hits_stream hands the text to a callback: one piece when the search fits one call, header + per-contig pieces
    in the per-contig mode; the pieces concatenate to calitas_search_hits' text either way; a failing sink fails the call."""
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    fa = synth_fasta(tmp_path, 59, [guide], lengths=(60000, 25000, 40000))
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        params = C.make_params(max_gaps_between_guide_and_pam=2)
        want, n = ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp", decode="bytes")
        for sequential in (False, True):
            if sequential:
                monkeypatch.setenv("CALITAS_SEQUENTIAL", "1")
            pieces = []
            nbytes, rows = ctx.search_hits_stream(C.Guide(guide), "a", params, lambda mv: pieces.append(bytes(mv)), "v0", "stamp")
            assert b"".join(pieces) == want and (nbytes, rows) == (len(want), n)
            assert len(pieces) == (1 if not sequential else 1 + 3)        # header + one piece per contig with hits
        def broken(_piece):
            raise IOError("disk full")
        with pytest.raises(IOError):
            ctx.search_hits_stream(C.Guide(guide), "a", params, broken, "v0", "stamp")
        monkeypatch.delenv("CALITAS_SEQUENTIAL")
        assert ctx.search_hits(C.Guide(guide), "a", params, "v0", "stamp", decode="bytes") == (want, n)
    finally:
        ctx.close()


def test_most_searches_take_the_binned_tail():
    """Which tail finished the fused calls of this module's product_rows (calitas_timing_t.binned_lanes): the per-bin kernels must be
    what the parity suite exercises, not a path that always declines.  (The dense / crowded cases are expected to decline.)"""
    assert len(PATHS) >= 15
    taken = sum(1 for x in PATHS if x > 0)
    assert taken >= 0.5 * len(PATHS), (taken, len(PATHS), PATHS)
