"""GPU parity of the variant branch (SearchReference.scala:570-630, BASELINE config 5 shape at test size): the reference's own
vector E4 and seeded random VCFs, every row against the oracle."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from fasta_util import write_fasta
from test_oracle_variants import write_vcf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = json.load(open(os.path.join(GOLD, "kat_variants.json")))
SKIP = {"aligner_version", "time_stamp"}


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


def _norm(rows):
    out = []
    for r in rows:
        r = {k: v for k, v in r.items() if k not in SKIP}
        if r.get("variant_vcf"):
            r["variant_vcf"] = r["variant_vcf"].split(":")[0]     # the oracle leaves the md5 out
        out.append(r)
    return out


def test_e4_through_gpu(C, tmp_path):
    e = V["e4"]
    fa = write_fasta(str(tmp_path / "e4.fa"), [("chr1", e["chr1"])], line_len=100)
    vcf = write_vcf(str(tmp_path / "e4.vcf"), [("chr1", p, i, r, a) for p, i, r, a in e["variants"]])
    out = str(tmp_path / "results.txt")
    C.SearchReference(guide=e["guide"], guide_id="test", ref=fa, variants=vcf, output=out, max_gaps_between_guide_and_pam=0,
                      max_guide_diffs=0).execute()
    hits = C.read_hits(out)
    x = e["expect"]
    assert len(hits) == x["n"]
    assert [int(h["coordinate_start"]) for h in hits] == x["coordinate_start"]
    for k in ("padded_extra_8_bases_5_prime", "padded_extra_8_bases_3_prime", "ten_bases_5_prime", "ten_bases_3_prime"):
        assert [h[k] for h in hits] == x[k], k
    _, want, _ = O.search_reference_vcf(fa, vcf, e["guide"], "test", d=0, g=0)
    assert _norm(hits) == _norm(want)


@pytest.mark.parametrize("guide,kw", [("CTTGCCCCACAGGGCAGTAAnrg", dict(d=4, p=1, g=2)), ("GTGACTTGAAGTCTCAGTATA", dict(d=5, p=1, g=3)),
                                      ("tttvAACCAACCAACCGGTTACGT", dict(d=3, p=1, g=1))])
def test_random_vcf_parity(C, guide, kw, tmp_path):
    from calitas_amd import synth
    G = C.Guide(guide)
    pam = G.pams[0] if G.pams else ""
    rng = np.random.default_rng(len(guide))
    names, seqs = synth.make_genome([("chr1", 40000), ("chr2", 15000)], seed=9, guides=[(G.guide, pam, G.pam_is_five_prime)],
                                    sites_per_guide=120, n_run_ends=100, n_block=800, softmask=0.3)
    contigs = [(n, s.tobytes().decode()) for n, s in zip(names, seqs)]
    fa = write_fasta(str(tmp_path / "v.fa"), contigs)
    # variants: ~1 per 150 bp, mixed kinds, some clustered, AF values, multi-allelic sites
    variants, afs = [], []
    for name, seq in contigs:
        pos, U = 200, seq.upper()
        while pos < len(seq) - 300:
            pos += int(rng.integers(5, 300))
            rb = U[pos - 1]
            if rb not in "ACGT":
                continue
            kind = int(rng.integers(0, 5))
            others = [b for b in "ACGT" if b != rb]
            if kind <= 1:
                ref, alts = rb, [others[int(rng.integers(0, 3))]]
            elif kind == 2:
                ref, alts = rb, [rb + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 6))))]
            elif kind == 3:
                ln = int(rng.integers(2, 6))
                ref = U[pos - 1:pos - 1 + ln]
                if any(c not in "ACGT" for c in ref):
                    continue
                alts = [rb]
            else:
                ref, alts = rb, others[:2]
            variants.append((name, pos, "rs%d" % len(variants) if rng.integers(0, 4) else "", ref, alts))
            afs.append([round(float(rng.uniform(0.0005, 0.5)), 4) for _ in alts])
            pos += len(ref)
    vcf = write_vcf(str(tmp_path / "v.vcf"), variants, afs)
    sr = C.SearchReference(guide=guide, guide_id="a", ref=fa, variants=vcf, max_guide_diffs=kw["d"], max_pam_mismatches=kw["p"],
                           max_gaps_between_guide_and_pam=kw["g"])
    text, n = sr.run("v0", "stamp")                            # calitas_search_variants (C++ window production and rows)
    import variants_twin                                        # the same branch written in Python: same bytes
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        assert variants_twin.search_reference_with_variants(sr, ctx, vcf, "v0", "stamp") == (text, n)
    finally:
        ctx.close()
    assert sr.variant_windows > 0
    got = C.read_hits(text)
    import gzip, shutil                                       # the same records gzipped: same rows (the file id differs)
    with open(vcf, "rb") as fi, gzip.open(str(tmp_path / "v.vcf.gz"), "wb") as fo:
        shutil.copyfileobj(fi, fo)
    sz = C.SearchReference(guide=guide, guide_id="a", ref=fa, variants=str(tmp_path / "v.vcf.gz"), max_guide_diffs=kw["d"],
                           max_pam_mismatches=kw["p"], max_gaps_between_guide_and_pam=kw["g"])
    blank = lambda rows: [dict(r, variant_vcf="") for r in rows]
    assert blank(C.read_hits(sz.run("v0", "stamp")[0])) == blank(got)
    _, want, _ = O.search_reference_vcf(fa, vcf, guide, "a", d=kw["d"], p=kw["p"], g=kw["g"])
    assert any(r["variant_id"] or r["variant_description"] for r in want)
    g2, w2 = _norm(got), _norm(want)
    if g2 != w2:
        gs = {json.dumps(r, sort_keys=True) for r in g2}
        ws = {json.dumps(r, sort_keys=True) for r in w2}
        raise AssertionError("product %d rows, oracle %d rows\nonly product: %s\nonly oracle: %s" % (
            len(g2), len(w2), [json.loads(x) for x in sorted(gs - ws)][:2], [json.loads(x) for x in sorted(ws - gs)][:2]))


def test_many_batches_equal_python_twin(C, tmp_path):
    """More variant windows than one alignment batch holds (16 384), dense clusters (reChunk, alleleCombos, the --max-variants
    cut) and multi-allelic records: calitas_search_variants against the Python implementation of the same branch, byte for byte,
    and against the oracle."""
    from calitas_amd import synth
    guide = "CTTGCCCCACAGGGCAGTAAnrg"
    G = C.Guide(guide)
    rng = np.random.default_rng(2024)
    names, seqs = synth.make_genome([("chr1", 1500000), ("chr2", 600000)], seed=12, guides=[(G.guide, G.pams[0], False)], sites_per_guide=400,
                                    n_run_ends=50, n_block=500, softmask=0.2)
    fa = write_fasta(str(tmp_path / "g.fa"), [(n, s.tobytes().decode()) for n, s in zip(names, seqs)])
    variants, afs = [], []
    for name, s in zip(names, seqs):
        U = s.tobytes().decode().upper()
        pos = 50
        while pos < len(U) - 50:
            pos += int(rng.integers(1, 40)) if rng.random() < 0.15 else int(rng.integers(40, 160))      # clusters and singletons
            if pos >= len(U) - 10:
                break
            rb = U[pos - 1]
            if rb not in "ACGT":
                continue
            others = [b for b in "ACGT" if b != rb]
            k = rng.random()
            if k < 0.75:
                ref, alts = rb, [others[int(rng.integers(0, 3))]] if rng.random() < 0.9 else others[:2]
            elif k < 0.88:
                ref, alts = rb, [rb + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 4))))]
            else:
                ln = int(rng.integers(2, 5))
                ref = U[pos - 1:pos - 1 + ln]
                if any(c not in "ACGT" for c in ref):
                    continue
                alts = [rb]
            variants.append((name, pos, "rs%d" % len(variants) if rng.integers(0, 4) else "", ref, alts))
            afs.append([round(float(rng.uniform(0.0005, 0.5)), 4) for _ in alts])
            pos += len(ref)
    vcf = write_vcf(str(tmp_path / "v.vcf"), variants, afs)
    sr = C.SearchReference(guide=guide, guide_id="a", ref=fa, variants=vcf, max_guide_diffs=4, max_pam_mismatches=1,
                           max_gaps_between_guide_and_pam=2, max_variants=3)
    text, n = sr.run("v0", "stamp")
    assert sr.variant_windows > 16384                          # more than one batch
    import variants_twin
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    try:
        assert variants_twin.search_reference_with_variants(sr, ctx, vcf, "v0", "stamp") == (text, n)
    finally:
        ctx.close()
    _, want, _ = O.search_reference_vcf(fa, vcf, guide, "a", d=4, p=1, g=2, max_variants=3)
    got = C.read_hits(text)
    assert sum(1 for r in got if r["variant_id"] or r["variant_description"]) > 10
    key = lambda r: json.dumps(r, sort_keys=True)
    assert sorted(map(key, _norm(got))) == sorted(map(key, _norm(want)))      # ties between groups may come in another order (SR:656)


def _dense_case(C, tmp_path, sizes, every, seed):
    """A PAM-less search permissive enough that every window holds alignments (BASELINE config 5's shape) + a VCF of SNVs and indels."""
    from calitas_amd import synth
    rng = np.random.default_rng(seed)
    names, seqs = synth.make_genome(sizes, seed=seed, guides=[("CTTGCCCCACAGGGCAGTAA", "", False)], sites_per_guide=40, n_run_ends=100,
                                    n_block=600, softmask=0.3)
    contigs = [(n, s.tobytes().decode()) for n, s in zip(names, seqs)]
    fa = write_fasta(str(tmp_path / "d.fa"), contigs)
    variants, afs = [], []
    for name, seq in contigs:
        U, pos = seq.upper(), 40
        while pos < len(U) - 60:
            pos += int(rng.integers(1, 2 * every))
            if pos >= len(U) - 10:
                break
            rb = U[pos - 1]
            if rb not in "ACGT":
                continue
            others = [b for b in "ACGT" if b != rb]
            k = rng.random()
            if k < 0.7:
                ref, alts = rb, [others[int(rng.integers(0, 3))]]
            elif k < 0.85:
                ref, alts = rb, [rb + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 4))))]
            else:
                ln = int(rng.integers(2, 5))
                ref = U[pos - 1:pos - 1 + ln]
                if any(c not in "ACGT" for c in ref):
                    continue
                alts = [rb]
            variants.append((name, pos, "rs%d" % len(variants), ref, alts))
            afs.append([round(float(rng.uniform(0.01, 0.5)), 4) for _ in alts])
            pos += len(ref)
    return fa, write_vcf(str(tmp_path / "d.vcf"), variants, afs)


@pytest.mark.parametrize("d,overlap", [(7, 10), (8, 10), (8, 25), (7, 1)])
def test_reference_rows_on_the_device_equal_the_host_merge(C, d, overlap, tmp_path, monkeypatch):
    """calitas_search_variants keeps the reference's own hits on the device: the hits of variant windows enter the device's
    removeOverlaps walk / order as key-only entries, the rows of the ones the walks keep are made on demand (hits.hpp, HitsExt::rows_for)
    and written into the holes the rows kernel leaves in the text once it is on the host (HitsExtRows::fill_on_host;
    CALITAS_VARIANTS_ROWS=device: the kept rows go to the device, =all: every entry comes with its finished row, as before round 5).  Same bytes as the host merge of
    alignment records (CALITAS_VARIANTS_HOST=1, the path the oracle comparisons of this file pinned), with and without a contig whose
    reference windows yield nothing, and the oracle's rows as a multiset (ties between groups: SR:656)."""
    fa, vcf = _dense_case(C, tmp_path, [("chr1", 60000), ("tiny", 90), ("chr2", 21000)], 80, seed=100 + d)
    kw = dict(guide="CTTGCCCCACAGGGCAGTAA", guide_id="c5", ref=fa, variants=vcf, max_guide_diffs=d, max_pam_mismatches=0,
              max_gaps_between_guide_and_pam=3, max_overlap=overlap)
    sr = C.SearchReference(**kw)
    text, n = sr.run("v0", "stamp")
    assert sr.timing["contig_passes"] == 3                      # one pass per contig, none declined
    monkeypatch.setenv("CALITAS_VARIANTS_COMPACT", "1")          # the reference passes' texts and the entries' rows compact (off by default)
    sc = C.SearchReference(**kw)
    assert sc.run("v0", "stamp") == (text, n) and sc.timing["contig_passes"] == 3
    monkeypatch.setenv("CALITAS_VARIANTS_ROWS", "all")           # ... and with every entry's row made up front, compact and whole
    sa = C.SearchReference(**kw)
    assert sa.run("v0", "stamp") == (text, n) and sa.timing["contig_passes"] == 3
    monkeypatch.delenv("CALITAS_VARIANTS_COMPACT")
    sa = C.SearchReference(**kw)
    assert sa.run("v0", "stamp") == (text, n) and sa.timing["contig_passes"] == 3
    monkeypatch.setenv("CALITAS_VARIANTS_ROWS", "device")        # ... and with the kept rows sent to the device instead of written into the text on the host
    sa = C.SearchReference(**kw)
    assert sa.run("v0", "stamp") == (text, n) and sa.timing["contig_passes"] == 3
    monkeypatch.delenv("CALITAS_VARIANTS_ROWS")
    monkeypatch.setenv("CALITAS_VARIANTS_HOST", "1")
    sh = C.SearchReference(**kw)
    text_h, n_h = sh.run("v0", "stamp")
    assert sh.timing["contig_passes"] == 0
    assert (text, n) == (text_h, n_h)
    got = C.read_hits(text)
    with_variant = sum(1 for r in got if r["variant_description"])
    assert with_variant > 20 and len(got) > 200, (with_variant, len(got))
    _, want, _ = O.search_reference_vcf(fa, vcf, kw["guide"], "c5", d=d, p=0, g=3, O=overlap)
    key = lambda r: json.dumps(r, sort_keys=True)
    assert sorted(map(key, _norm(got))) == sorted(map(key, _norm(want)))


@pytest.mark.parametrize("host_merge,block", [(False, "pinned"), (False, "alloc_host"), (False, "pageable"), (True, "pinned")])
def test_text_into_the_callers_page_locked_buffer(C, host_merge, block, tmp_path, monkeypatch):
    """calitas_search_variants_into: the same bytes as calitas_search_variants, delivered into memory of the caller's -- every contig's
    rows straight to their place over the bus when the device writes the rows, a copy of the library's block when the merge ran on the
    host --, and CALITAS_EINVAL, not an overrun, when the buffer is too small (at the header, and in the middle of the contigs)."""
    fa, vcf = _dense_case(C, tmp_path, [("chr1", 60000), ("tiny", 90), ("chr2", 21000)], 80, seed=107)
    kw = dict(guide="CTTGCCCCACAGGGCAGTAA", guide_id="c5", ref=fa, variants=vcf, max_guide_diffs=7, max_pam_mismatches=0,
              max_gaps_between_guide_and_pam=3, max_overlap=10)
    if host_merge:
        monkeypatch.setenv("CALITAS_VARIANTS_HOST", "1")
    text, n = C.SearchReference(**kw).run("v0", "stamp")
    want = text.encode()
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    params = C.make_params(max_guide_diffs=7, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3, max_overlap=10)
    g = C.Guide("CTTGCCCCACAGGGCAGTAA")
    if block == "alloc_host":                                   # a block of the runtime's own: the copy engines' favourite
        import ctypes
        addr = C.Context.alloc_host(len(want) + 4096)
        buf = np.ctypeslib.as_array((ctypes.c_uint8 * (len(want) + 4096)).from_address(addr))
        buf[:] = 0x55
    else:                                                       # memory of the caller's: locked in place, or -- slower, as good -- not at all
        buf = np.full(len(want) + 4096, 0x55, dtype=np.uint8)
        if block == "pinned":
            ctx.pin_host(buf.ctypes.data, buf.nbytes)
    try:
        for _ in range(2):                                      # (the buffer is reused from call to call)
            nb, rows, nwin = ctx.search_variants_into(g, "c5", params, vcf, buf.ctypes.data, buf.nbytes, "v0", "stamp")
            assert (nb, rows) == (len(want), n) and nwin > 0
            assert bytes(buf[:nb]) == want and buf[nb] == 0 and buf[nb + 1] == 0x55
        nb2, rows2, _ = ctx.search_variants_into(g, "c5", params, vcf, buf.ctypes.data, len(want) + 1, "v0", "stamp")   # exactly enough
        assert (nb2, rows2) == (len(want), n) and bytes(buf[:nb2]) == want
        monkeypatch.setenv("CALITAS_VARIANTS_COMPACT", "1")     # ... and with the per-contig texts compact on the bus, expanded into the buffer
        buf[:] = 0x55
        nb3, rows3, _ = ctx.search_variants_into(g, "c5", params, vcf, buf.ctypes.data, buf.nbytes, "v0", "stamp")
        assert (nb3, rows3) == (len(want), n) and bytes(buf[:nb3]) == want and buf[nb3] == 0 and buf[nb3 + 1] == 0x55
        monkeypatch.delenv("CALITAS_VARIANTS_COMPACT")
        for cap in (16, len(want) // 2, len(want)):             # too small: at the header, part of the way, by the final NUL
            buf[:] = 0x55
            with pytest.raises(C.CalitasError, match="too small|does not hold the header"):
                ctx.search_variants_into(g, "c5", params, vcf, buf.ctypes.data, cap, "v0", "stamp")
            assert (buf[cap:] == 0x55).all()                    # nothing behind the capacity was touched
    finally:
        if block == "alloc_host":
            del buf
            C.Context.free_host(addr)
        elif block == "pinned":
            ctx.unpin_host(buf.ctypes.data)
        ctx.close()


@pytest.mark.parametrize("which", [0, 1, 2, 4])
def test_a_failed_aligner_batch_fails_the_call_instead_of_hanging_it(C, which, tmp_path, monkeypatch):
    """The variant branch's stages (producer -> two aligners -> lifter) hand batches to the lifter in the order they were built.  A
    batch that fails in one aligner makes that stage drop the jobs behind it; the other aligner's jobs must still get their turn
    (StageThread's `skipped` handler passes it), so the call comes back with the error -- it used to wait forever.  Six contigs give
    six batches and six contig-end jobs, alternating between the two aligners; the k-th batch is made to fail (CALITAS_FAIL_ALIGN_BATCH)."""
    import threading
    fa, vcf = _dense_case(C, tmp_path, [("c%d" % i, 9000) for i in range(6)], 120, seed=31)
    kw = dict(guide="CTTGCCCCACAGGGCAGTAA", guide_id="c5", ref=fa, variants=vcf, max_guide_diffs=6, max_pam_mismatches=0,
              max_gaps_between_guide_and_pam=3)
    want = C.SearchReference(**kw).run("v0", "stamp")
    monkeypatch.setenv("CALITAS_FAIL_ALIGN_BATCH", str(which))
    box = {}

    def call():
        try:
            box["out"] = C.SearchReference(**kw).run("v0", "stamp")
        except Exception as e:                                   # noqa: BLE001 -- the error is what the test is about
            box["err"] = str(e)

    t = threading.Thread(target=call, daemon=True)
    t.start()
    t.join(60)
    assert not t.is_alive(), "calitas_search_variants did not return after a failed batch"
    assert "out" not in box and "injected failure" in box.get("err", ""), box
    monkeypatch.delenv("CALITAS_FAIL_ALIGN_BATCH")
    assert C.SearchReference(**kw).run("v0", "stamp") == want    # and the library is fine afterwards


def test_device_merge_declines_to_the_host(C, tmp_path):
    """-O 0 is beyond the device's row stage: the call merges on the host, same rows as the oracle."""
    fa, vcf = _dense_case(C, tmp_path, [("chr1", 30000)], 300, seed=77)
    sr = C.SearchReference(guide="CTTGCCCCACAGGGCAGTAA", guide_id="c5", ref=fa, variants=vcf, max_guide_diffs=5, max_pam_mismatches=0,
                           max_gaps_between_guide_and_pam=3, max_overlap=0)
    text, n = sr.run("v0", "stamp")
    assert sr.timing["contig_passes"] == 0
    _, want, _ = O.search_reference_vcf(fa, vcf, "CTTGCCCCACAGGGCAGTAA", "c5", d=5, p=0, g=3, O=0)
    key = lambda r: json.dumps(r, sort_keys=True)
    assert sorted(map(key, _norm(C.read_hits(text)))) == sorted(map(key, _norm(want)))


def test_cpp_cli_with_variants(C, tmp_path):
    """`calitas SearchReference --variants` (the C++ front end; the library computes the VCF's name:md5 identifier itself) writes the
    rows the Python API returns, whose identifier comes from hashlib."""
    import subprocess
    from calitas_amd import variants as VV
    e = V["e4"]
    fa = write_fasta(str(tmp_path / "e4.fa"), [("chr1", e["chr1"])], line_len=100)
    vcf = write_vcf(str(tmp_path / "e4.vcf"), [("chr1", p, i, r, a) for p, i, r, a in e["variants"]])
    exe = os.path.join(os.path.dirname(os.path.abspath(C.__file__)), "calitas")
    out = str(tmp_path / "cli.txt")
    r = subprocess.run([exe, "SearchReference", "-i", e["guide"], "-I", "test", "-r", fa, "-v", vcf, "-o", out, "-g", "0", "-d", "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = C.read_hits(out)
    sr = C.SearchReference(guide=e["guide"], guide_id="test", ref=fa, variants=vcf, max_gaps_between_guide_and_pam=0, max_guide_diffs=0)
    want = C.read_hits(sr.run()[0])
    strip = lambda rows: [{k: v for k, v in r.items() if k not in SKIP} for r in rows]
    assert strip(got) == strip(want) and len(got) == e["expect"]["n"]
    ids = {r["variant_vcf"] for r in got if r["variant_vcf"]}
    assert ids == {VV.vcf_identifier(vcf)}                     # "e4.vcf:<md5>" from the library's own MD5 == hashlib's


def test_variant_search_error_paths_and_an_empty_vcf(C, tmp_path):
    """What the stages of calitas_search_variants do when the VCF is not what it should be (the reference passes are under way by the time
    the file is looked at, round 5): a file that is not there, contigs out of the reference's order, a contig the reference does not have
    -- CALITAS errors, no hang, and the context searches on afterwards; a VCF without records gives the reference search's own text."""
    import threading
    fa, vcf = _dense_case(C, tmp_path, [("chr1", 30000), ("chr2", 12000)], 150, seed=41)
    kw = dict(guide="CTTGCCCCACAGGGCAGTAA", guide_id="c5", ref=fa, max_guide_diffs=6, max_pam_mismatches=0, max_gaps_between_guide_and_pam=3)
    lines = open(vcf).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    recs = [l for l in lines if not l.startswith("#")]
    swapped = tmp_path / "swapped.vcf"
    swapped.write_text("\n".join(head + [l for l in recs if l.startswith("chr2\t")] + [l for l in recs if l.startswith("chr1\t")]) + "\n")
    foreign = tmp_path / "foreign.vcf"
    foreign.write_text("\n".join(head + recs[:5] + ["chrZ\t10\t.\tA\tC\t.\tPASS\tAF=0.1"]) + "\n")
    empty = tmp_path / "empty.vcf"
    empty.write_text("\n".join(head) + "\n")
    ctx = C.Context(0)
    ctx.set_reference_fasta(fa)
    box = {}

    def run(path):
        try:
            box["out"] = C.SearchReference(variants=str(path), context=ctx, **kw).run("v0", "stamp")
        except Exception as e:                                  # noqa: BLE001 (the test looks at it)
            box["err"] = e

    try:
        for path, pattern in ((tmp_path / "absent.vcf", "cannot read"), (swapped, "not in reference order"), (foreign, "not in reference order")):
            box.clear()
            t = threading.Thread(target=run, args=(path,), daemon=True)
            t.start()
            t.join(60)
            assert not t.is_alive(), "calitas_search_variants hangs on %s" % path
            assert isinstance(box.get("err"), C.CalitasError) and pattern in str(box["err"]), box
        want = C.SearchReference(context=ctx, **kw).run("v0", "stamp")
        got = C.SearchReference(variants=str(empty), context=ctx, **kw).run("v0", "stamp")
        assert got == want and want[1] > 0
        assert C.SearchReference(variants=vcf, context=ctx, **kw).run("v0", "stamp")[1] >= want[1]   # ... and the context is as good as new
    finally:
        ctx.close()
