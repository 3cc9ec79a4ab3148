"""GPU: BASELINE configs 4 and 5 at sizes the oracle covers, the persistent index on a device context, and the state a failed
reference upload leaves behind.

config 4 = "SearchReference: 96-guide batch vs hg38, same diff limits" (README.md:73-83 of the reference: one guide per
invocation, SearchReference.scala:452-453 -- the batch is the loop a caller writes around it): guide #0 + 95 random 20-mers
(seed 0xC4) through calitas_search_hits_batch, every guide's rows against the oracle.
config 5 = "hg38 + VCF via PrepareVcf, PAM-less search, max-guide-diffs=8" (SearchReference.scala:570-630, PrepareVcf.scala:63-85).
The full-size halves of both live in test_gpu_fullsize.py.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from fasta_util import write_fasta
from test_oracle_variants import write_vcf

pytestmark = pytest.mark.gpu
SKIP = {"aligner_version", "time_stamp"}
GUIDE0 = "CTTGCCCCACAGGGCAGTAAnrg"


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


def strip(rows, build=None):
    out = []
    for r in rows:
        r = {k: v for k, v in r.items() if k not in SKIP}
        if build is not None:
            r["genome_build"] = build
        if r.get("variant_vcf"):
            r["variant_vcf"] = r["variant_vcf"].split(":")[0]     # the oracle leaves the md5 out
        out.append(r)
    return out


@pytest.mark.timeout(1500)
def test_config4_96_guide_batch_against_the_oracle(C):
    """All 96 guides of BASELINE config 4 on 60 Mb of the bench genome's recipe, one calitas_search_hits_batch call (guides
    pipelined through the lanes); each guide's hits.txt against the oracle's, every column."""
    from calitas_amd import synth
    guides = [GUIDE0] + synth.random_guides(0xC4, 95)
    planted = [(g[:20], "nrg", False) for g in guides[:6]]
    names, seqs = synth.make_genome([("chrA", 38_000_000), ("chrB", 22_000_000)], seed=0xC4, guides=planted, sites_per_guide=120,
                                    n_run_ends=10_000, n_block=1_200_000, softmask=0.5, tandem_frac=0.01)
    ctx = C.Context(0)
    try:
        ctx.set_reference(names, seqs)
        params = C.make_params(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
        G = [C.Guide(g) for g in guides]
        ids = ["g%02d" % i for i in range(len(guides))]
        res = ctx.search_hits_batch(G, ids, params, "v0", "stamp")
        assert len(res) == 96 and ctx.timing()["lanes"] >= 2
        # a guide of the batch and the same guide alone: same bytes
        for i in (0, 17, 95):
            assert ctx.search_hits(G[i], ids[i], params, "v0", "stamp") == res[i]
    finally:
        ctx.close()
    raw = [s.tobytes() for s in seqs]
    total = 0
    for i, g in enumerate(guides):
        _, want, nwin = O.search_memory(names, raw, g, ids[i], d=5, p=1, g=2, threads=16)
        got = C.read_hits(res[i][0])
        assert nwin > 55000 and len(got) == res[i][1]
        assert strip(got) == strip(want), "guide %d (%s): %d rows, oracle %d" % (i, g, len(got), len(want))
        total += len(got)
    assert total > 96 * 500


def c5_genome(tmp_path, lengths, seed=3):
    from calitas_amd import synth
    guide = "GTGACTTGAAGTCTCAGTAT"
    rng = np.random.default_rng(5)
    names, seqs = synth.make_genome([("chr%d" % (i + 1), n) for i, n in enumerate(lengths)], seed=seed, guides=[(guide, "", False)],
                                    sites_per_guide=20, n_run_ends=20, n_block=200, softmask=0.2)
    fa = write_fasta(str(tmp_path / "c5.fa"), [(n, s.tobytes().decode()) for n, s in zip(names, seqs)])
    variants, afs = [], []
    for name, s in zip(names, seqs):
        U = s.tobytes().decode().upper()
        pos = 100
        while pos < len(U) - 100:
            pos += int(rng.integers(200, 1800))                  # about one variant per kilobase
            if pos >= len(U) - 10 or U[pos - 1] not in "ACGT":
                continue
            rb = U[pos - 1]
            others = [b for b in "ACGT" if b != rb]
            kind = int(rng.integers(0, 4))
            if kind <= 1:
                variants.append((name, pos, "rs%d" % len(variants), rb, [others[int(rng.integers(0, 3))]]))
            elif kind == 2:
                variants.append((name, pos, "rs%d" % len(variants), rb, [rb + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))]))
            else:
                ln = int(rng.integers(2, 5))
                ref = U[pos - 1:pos - 1 + ln]
                if any(c not in "ACGT" for c in ref):
                    continue
                variants.append((name, pos, "rs%d" % len(variants), ref, [ref[0]]))
            afs.append([round(float(rng.uniform(0.01, 0.5)), 3)])
    vcf = write_vcf(str(tmp_path / "c5.vcf"), variants, afs)
    return guide, fa, vcf, len(variants)


def test_config5_pamless_d8_with_vcf_against_the_oracle(C, tmp_path):
    """The shape of BASELINE config 5 at a size the oracle covers: PAM-less 20-mer, max-guide-diffs 8, --variants.  Rows as a
    multiset: ties between a variant group and the reference group are unordered in the reference (a hash map, SearchReference.scala:656)."""
    guide, fa, vcf, n_var = c5_genome(tmp_path, (30000, 9000))
    sr = C.SearchReference(guide=guide, guide_id="c5", ref=fa, variants=vcf, max_guide_diffs=8, max_pam_mismatches=0,
                           max_gaps_between_guide_and_pam=3)
    text, n = sr.run("v", "t")
    got = C.read_hits(text)
    _, want, _ = O.search_reference_vcf(fa, vcf, guide, "c5", d=8, p=0, g=3)
    key = lambda r: json.dumps(r, sort_keys=True)
    assert n_var > 30 and n == len(got) and len(got) > 500
    assert sum(1 for r in got if r["variant_id"]) > 20                   # hits that exist only with a variant allele
    assert sorted(map(key, strip(got))) == sorted(map(key, strip(want)))
    # the reference rows of the same search (no VCF) come in the reference's order, so they compare as a list
    text0, n0 = C.SearchReference(guide=guide, guide_id="c5", ref=fa, max_guide_diffs=8, max_pam_mismatches=0,
                                  max_gaps_between_guide_and_pam=3).run("v", "t")
    _, want0, _ = O.search_reference(fa, guide, "c5", d=8, p=0, g=3)
    assert n0 > 400 and strip(C.read_hits(text0)) == strip(want0)


def test_config5_per_contig_mode_through_the_memory_budget(C, tmp_path, monkeypatch):
    """PAM-less d = 8 keeps 2.4 KB of strip per scan record; with a device budget that one pass over the reference exceeds and a
    contig's pass does not, calitas_search_hits and calitas_search_hits_stream go to one pass per contig by themselves (the planner's
    estimate, not a failed allocation) and return the bytes of the unrestricted call."""
    guide, fa, _, _ = c5_genome(tmp_path, (400000, 250000, 90000), seed=11)
    ctx = C.Context(0)
    try:
        ctx.set_reference_fasta(fa)
        params = C.make_params(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
        G = C.Guide(guide)
        monkeypatch.setenv("CALITAS_CHUNKS", "1")
        want = ctx.search_hits(G, "c5", params, "v", "t", decode="bytes")
        assert want[1] > 5000 and ctx.timing()["contig_passes"] == 0
        monkeypatch.setenv("CALITAS_DEVICE_BUDGET_MB", "200")
        got = ctx.search_hits(G, "c5", params, "v", "t", decode="bytes")
        assert got == want and ctx.timing()["contig_passes"] == 3
        pieces = []
        nbytes, rows = ctx.search_hits_stream(G, "c5", params, lambda mv: pieces.append(bytes(mv)), "v", "t")
        assert b"".join(pieces) == want[0] and rows == want[1] and len(pieces) == 4
        monkeypatch.setenv("CALITAS_DEVICE_BUDGET_MB", "1")
        with pytest.raises(C.CalitasError) as e:
            ctx.search_hits(G, "c5", params, "v", "t")
        assert e.value.code == C._lib.ENOMEM
        monkeypatch.delenv("CALITAS_DEVICE_BUDGET_MB")
        assert ctx.search_hits(G, "c5", params, "v", "t", decode="bytes") == want
    finally:
        ctx.close()


def test_search_on_a_loaded_index(C, tmp_path):
    """calitas_save_index -> a fresh device context -> calitas_load_index -> the same hits.txt bytes as the FASTA path; a corrupted
    or truncated file is refused with CALITAS_EIO instead of reaching the kernels."""
    import test_gpu_parity as P
    fa = P.synth_fasta(tmp_path, 71, [GUIDE0], lengths=(90000, 40000, 700, 26))
    params = C.make_params(max_guide_diffs=4, max_gaps_between_guide_and_pam=2)
    a = C.Context(0)
    try:
        a.set_reference_fasta(fa)
        want = a.search_hits(C.Guide(GUIDE0), "a", params, "v0", "stamp")
        idx = str(tmp_path / "ref.calidx")
        a.save_index(idx)
        build = a.genome_build()
    finally:
        a.close()
    assert want[1] > 20
    b = C.Context(0)
    try:
        b.load_index(idx)
        assert b.contig_names == ["ctg%d" % i for i in range(4)] and b.genome_build() == build
        assert b.search_hits(C.Guide(GUIDE0), "a", params, "v0", "stamp") == want
        blob = bytearray(open(idx, "rb").read())
        blob[len(blob) // 2] ^= 0x40                                   # one flipped bit in the packed codes
        bad = str(tmp_path / "bad.calidx")
        open(bad, "wb").write(blob)
        with pytest.raises(C.CalitasError, match="checksum") as e:
            b.load_index(bad)
        assert e.value.code == C._lib.EIO
        open(bad, "wb").write(blob[:len(blob) - 4096])
        with pytest.raises(C.CalitasError, match="not a calitas index") as e:
            b.load_index(bad)
        # the context still holds the good index
        assert b.search_hits(C.Guide(GUIDE0), "a", params, "v0", "stamp") == want
    finally:
        b.close()


def test_failed_reference_upload_leaves_no_half_set_reference(C, tmp_path, monkeypatch):
    """An upload that fails (here: the packed reference exceeds the device budget) must not leave has_ref set with freed device
    pointers: the next search answers CALITAS_ESTATE, and a later successful upload works."""
    import test_gpu_parity as P
    fa = P.synth_fasta(tmp_path, 72, [GUIDE0], lengths=(3_000_000, 40000))
    params = C.make_params(max_guide_diffs=3)
    ctx = C.Context(0)
    try:
        ctx.set_reference_fasta(fa)
        want = ctx.search_hits(C.Guide(GUIDE0), "a", params, "v0", "stamp")
        monkeypatch.setenv("CALITAS_DEVICE_BUDGET_MB", "1")
        with pytest.raises(C.CalitasError) as e:
            ctx.set_reference_fasta(fa)
        assert e.value.code == C._lib.ENOMEM
        monkeypatch.delenv("CALITAS_DEVICE_BUDGET_MB")
        with pytest.raises(C.CalitasError) as e:
            ctx.search_hits(C.Guide(GUIDE0), "a", params, "v0", "stamp")
        assert e.value.code == C._lib.ESTATE
        ctx.set_reference_fasta(fa)
        assert ctx.search_hits(C.Guide(GUIDE0), "a", params, "v0", "stamp") == want
    finally:
        ctx.close()


def test_window_ranges_of_calitas_search_concatenate_to_the_whole_call(C, tmp_path):
    """calitas_params_t first_window / n_windows (the piece of a job one rank of a window-range partition runs, shard.window_partition):
    the alignments of consecutive ranges -- cuts inside contigs, inside an N block, one range a single window -- concatenate to the
    alignments of the whole call, record for record, and calitas_hits_tsv on them gives calitas_search_hits' bytes; calitas_search_hits
    on a range returns the rows whose coordinate_start lies in the range's stretch (their texts concatenate as well)."""
    import test_gpu_parity as P
    from calitas_amd import shard
    fa = P.synth_fasta(tmp_path, 73, [GUIDE0], lengths=(1_400_000, 300_000, 90_000, 700, 26), n_block=40_000)
    ctx = C.Context(0)
    try:
        ctx.set_reference_fasta(fa)
        G = C.Guide(GUIDE0)
        kw = dict(max_guide_diffs=5, max_pam_mismatches=1, max_gaps_between_guide_and_pam=2)
        params = C.make_params(**kw)
        step = 1000 - (G.cli_length + 5 + 2 - 1)
        want_text = ctx.search_hits(G, "a", params, "v0", "stamp")
        whole = ctx.search([G], params)
        key = lambda a: (a.contig_index, a.window_start, a.strand, a.start_offset, a.end_offset, a.score, a.pam_index, a.ops)
        total = sum(shard.window_counts(ctx.contig_lengths, step))
        for cuts in ([(0, total)], shard.window_partition(ctx.contig_lengths, 3, step), [(0, 700), (700, 1), (701, total - 701)],
                     shard.window_partition(ctx.contig_lengths, 8, step)):
            got = []
            for f, n in cuts:
                got += ctx.search([G], C.make_params(first_window=f, n_windows=n, **kw))
            assert [key(a) for a in got] == [key(a) for a in whole], cuts
        assert len(whole) > 60
        assert ctx.hits_tsv(G, "a", params, got, "v0", "stamp") == want_text
        # calitas_search_hits on a range returns the rows the range OWNS (coordinate_start inside its stretch): consecutive ranges'
        # texts concatenate to the whole text, cuts inside contigs and inside the N block included
        for cuts in (shard.window_partition(ctx.contig_lengths, 3, step), [(0, 700), (700, 1), (701, total - 701)],
                     shard.window_partition(ctx.contig_lengths, 8, step)):
            body, rows = "", 0
            for f, n in cuts:
                text, nr = ctx.search_hits(G, "a", C.make_params(first_window=f, n_windows=n, **kw), "v0", "stamp")
                body += text.split("\n", 1)[1]
                rows += nr
            assert want_text[0].split("\n", 1)[0] + "\n" + body == want_text[0] and rows == want_text[1], cuts
        # the batch call takes a range since round 4 (every guide's rows of the stretch); the stream call refuses one
        pr = C.make_params(first_window=0, n_windows=10, **kw)
        assert ctx.search_hits_batch([G, G], ["a", "b"], pr, "v0", "stamp")[0] == ctx.search_hits(G, "a", pr, "v0", "stamp")
        with pytest.raises(C.CalitasError, match="window range"):
            ctx.search_hits_stream(G, "a", pr, lambda piece: None, "v0", "stamp")
        with pytest.raises(C.CalitasError, match="outside the window table"):
            ctx.search_hits_batch([G, G], ["a", "b"], C.make_params(first_window=total - 5, n_windows=10, **kw), "v0", "stamp")
        with pytest.raises(C.CalitasError, match="outside the window table"):
            ctx.search([G], C.make_params(first_window=total - 5, n_windows=10, **kw))
    finally:
        ctx.close()


def test_search_hits_into_a_caller_buffer(C, tmp_path):
    """calitas_search_hits_into: the same bytes as calitas_search_hits, delivered into memory of the caller (page-locked with
    calitas_pin_host, or plain); a buffer that is too small is refused with CALITAS_EINVAL and nothing is written past it."""
    import ctypes
    import test_gpu_parity as P
    fa = P.synth_fasta(tmp_path, 74, [GUIDE0], lengths=(250000, 60000, 700))
    ctx = C.Context(0)
    try:
        ctx.set_reference_fasta(fa)
        G = C.Guide(GUIDE0)
        params = C.make_params(max_guide_diffs=5, max_gaps_between_guide_and_pam=2)
        want, rows = ctx.search_hits(G, "a", params, "v0", "stamp", decode="bytes")
        assert rows > 50
        cap = len(want) + 4096
        for pinned in (True, False):
            buf = ctypes.create_string_buffer(b"\xAA" * (cap + 64), cap + 64)
            addr = ctypes.addressof(buf)
            if pinned:
                ctx.pin_host(addr, cap)
            try:
                for chunks in ("1", "2"):
                    os.environ["CALITAS_CHUNKS"] = chunks
                    n, r = ctx.search_hits_into(G, "a", params, addr, cap, "v0", "stamp")
                    assert (n, r) == (len(want), rows) and buf.raw[:n] == want and buf.raw[n] == 0
                    assert buf.raw[cap:] == b"\xAA" * 64
                with pytest.raises(C.CalitasError, match="too small") as e:
                    ctx.search_hits_into(G, "a", params, addr, len(want) // 2, "v0", "stamp")
                assert e.value.code == C._lib.EINVAL and buf.raw[cap:] == b"\xAA" * 64
            finally:
                os.environ.pop("CALITAS_CHUNKS", None)
                if pinned:
                    ctx.unpin_host(addr)
    finally:
        ctx.close()


def test_record_estimate_with_a_chromosome_filter(C, tmp_path, monkeypatch):
    """The planner's sampled record estimate honours --chrom: a budget that the whole reference would exceed but the one contig does not
    lets the filtered search run as one pass, with the rows of that contig alone."""
    guide, fa, _, _ = c5_genome(tmp_path, (400000, 250000, 90000), seed=12)
    ctx = C.Context(0)
    try:
        ctx.set_reference_fasta(fa)
        G = C.Guide(guide)
        kw = dict(max_guide_diffs=8, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
        whole = C.read_hits(ctx.search_hits(G, "c5", C.make_params(**kw), "v", "t")[0])
        monkeypatch.setenv("CALITAS_DEVICE_BUDGET_MB", "200")
        text, rows = ctx.search_hits(G, "c5", C.make_params(chrom_index=2, **kw), "v", "t")
        assert ctx.timing()["contig_passes"] <= 1
        assert C.read_hits(text) == [r for r in whole if r["chromosome"] == "chr3"] and rows > 500
    finally:
        ctx.close()
