"""CPU tests of the Python cross-check of the variant branch (tests/variants_twin.py, what the GPU tests hold calitas_search_variants
against) and of the package's VCF helpers against the reference's own vectors
V1-V9 (SearchReferenceTest.scala:150-295) and against the oracle on random variant sets; PrepareVcf (PrepareVcfTest.scala)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import calitas_amd.variants as PV
from test_oracle_variants import write_vcf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = json.load(open(os.path.join(GOLD, "kat_variants.json")))


@pytest.fixture(scope="module")
def VA():
    import variants_twin
    return variants_twin


@pytest.mark.parametrize("case", V["allele_combos_counts"], ids=lambda c: "x".join(map(str, c["counts"])))
def test_allele_combos_counts(VA, case):
    assert VA.allele_combos_counts(case["counts"]) == case["expect"]


def _variants(VA, spec, chrom="chr1"):
    return [VA.Variant(chrom, pos, vid, ref, alts) for pos, vid, ref, alts in spec]


@pytest.mark.parametrize("case", V["build_variant_window"], ids=lambda c: c["id"])
def test_build_variant_window(VA, case):
    vs = _variants(VA, case["variants"])
    w = VA.build_variant_window(vs, case["alleles"], "chr1", case["ref"].upper().encode(), case["padding"])
    assert w.bases.decode() == case["bases"] and w.cigar_string == case["cigar"]
    for off, prec, want in case["offsets"]:
        assert w.ref_offset_at_base_offset(off, prec) == want


@pytest.mark.parametrize("case", V["allele_combos_variants"], ids=lambda c: "%s-max%d" % (c["lines"], c["max"]))
def test_allele_combos_variants(VA, case):
    sets = [["%s=%d" % (v.id, a) for v, a in zip(vs, al)] for vs, al in VA.allele_combos(_variants(VA, case["variants"]), case["max"])]
    if "expect" in case:
        assert sorted(map(tuple, sets)) == sorted(map(tuple, case["expect"]))
    else:
        assert len(sets) == case["expect_size"]


def test_random_variant_windows_match_oracle(VA):
    """Random clusters of SNVs / insertions / deletions / multi-allelic sites: same windows (bases, cigar, start, lift-back)
    as the oracle for every allele combination."""
    rng = np.random.default_rng(5)
    ref = "".join(rng.choice(list("ACGT"), size=400))
    for trial in range(40):
        n = int(rng.integers(1, 5))
        pos, spec = int(rng.integers(5, 40)), []
        for k in range(n):
            pos += int(rng.integers(1, 30))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                r, alts = ref[pos - 1], [rng.choice([b for b in "ACGT" if b != ref[pos - 1]])]
            elif kind == 1:
                r, alts = ref[pos - 1], [ref[pos - 1] + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5))))]
            elif kind == 2:
                ln = int(rng.integers(2, 6)); r, alts = ref[pos - 1:pos - 1 + ln], [ref[pos - 1]]
            else:
                r, alts = ref[pos - 1], [b for b in "ACGT" if b != ref[pos - 1]][:2]
            spec.append([pos, "v%d" % k, r, alts])
            pos += len(r)
        vs = _variants(VA, spec)
        padding = int(rng.integers(5, 35))
        for variants, alleles in VA.allele_combos(vs, 16):
            sel = [alleles[variants.index(v)] if v in variants else 0 for v in vs]
            w = VA.build_variant_window(variants, alleles, "chr1", ref.encode(), padding)
            queries = [(o, p) for o in range(0, len(w.bases) + 1, 7) for p in (True, False)]
            try:
                ob, oc, ostart, ooffs = O.build_variant_window(ref, spec, sel, padding, queries)
            except RuntimeError:
                continue
            assert (w.bases.decode(), w.cigar_string, w.start) == (ob, oc, ostart)
            assert [w.ref_offset_at_base_offset(o, p) for o, p in queries] == ooffs


def test_prepare_vcf(VA, tmp_path):
    """PrepareVcfTest.scala:9-39 shape: 10 PASS variants with AF 0.5 and genotypes -> 10 variants, no samples; plus AF filtering."""
    src = tmp_path / "in.vcf"
    with open(src, "w") as f:
        f.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"x\">\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"x\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample1\tsample2\n")
        for i in range(10):
            f.write("1\t%d\t.\tA\tC\t.\tPASS\tAF=0.5;DP=10\tGT\t0/1\t./.\n" % (1000 * (i + 1)))
        f.write("1\t20000\t.\tA\tC,G\t.\tPASS\tAF=0.001,0.2\tGT\t0/1\t./.\n")
        f.write("1\t21000\t.\tA\tC\t.\tq10\tAF=0.5\tGT\t0/1\t./.\n")
        f.write("1\t22000\t.\tA\tC\t.\tPASS\tAF=0.0001\tGT\t0/1\t./.\n")
    out = tmp_path / "out.vcf"
    n = PV.prepare_vcf([str(src)], str(out), min_af=0.01)
    hdr, vs = PV.read_vcf(str(out))
    assert n == 11 and len(vs) == 11
    assert [h for h in hdr if h.startswith("#CHROM")][0].split("\t")[-1] == "INFO"     # no samples
    assert all(v.chrom == "chr1" for v in vs)
    assert vs[10].alts == ["G"] and vs[10].afs == [0.2]


def test_prepare_vcf_with_a_sequence_dictionary(VA, tmp_path):
    """PrepareVcf --dict (PrepareVcf.scala:36, 46-56): contig lines from the dictionary's sequences (length, assembly), every old
    `reference` line replaced by the first sequence's assembly, records untouched; the same through `python -m calitas_amd
    PrepareVcf -d`; a dictionary next to a FASTA is found the way htsjdk finds it; a .fai (no assembly) is refused with a reason."""
    src = tmp_path / "in.vcf"
    with open(src, "w") as f:
        f.write("##fileformat=VCFv4.2\n##reference=file:///old/b37.fa\n##contig=<ID=1,length=5>\n##contig=<ID=2,length=7>\n")
        f.write("##INFO=<ID=AF,Number=A,Type=Float,Description=\"x\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\n")
        for i in range(10):
            f.write("1\t%d\trs%d\tA\tC\t.\tPASS\tAF=0.5\tGT\t0/1\n" % (1000 * (i + 1), i))
    d = tmp_path / "ref.dict"
    with open(d, "w") as f:
        f.write("@HD\tVN:1.5\tSO:unsorted\n@SQ\tSN:chr1\tLN:248956422\tM5:abc\tAS:hg38\tUR:file:/x\n@SQ\tSN:chr2\tLN:242193529\tAS:hg38\n@SQ\tSN:chrM\tLN:16569\n")
    out = tmp_path / "out.vcf"
    assert PV.prepare_vcf([str(src)], str(out), dict_path=str(d)) == 10
    hdr, vs = PV.read_vcf(str(out))
    assert [h for h in hdr if h.startswith("##contig=")] == ["##contig=<ID=chr1,length=248956422,assembly=hg38>",
                                                             "##contig=<ID=chr2,length=242193529,assembly=hg38>", "##contig=<ID=chrM,length=16569>"]
    assert [h for h in hdr if h.startswith("##reference=")] == ["##reference=hg38"]
    assert hdr[0] == "##fileformat=VCFv4.2" and hdr[-1].split("\t")[-1] == "INFO" and any(h.startswith("##INFO=<ID=AF") for h in hdr)
    assert len(vs) == 10 and all(v.chrom == "chr1" for v in vs) and [v.id for v in vs] == ["rs%d" % i for i in range(10)]
    # without --dict the header is the input's (PrepareVcf.scala:45)
    plain = tmp_path / "plain.vcf"
    PV.prepare_vcf([str(src)], str(plain))
    assert [h for h in PV.read_vcf(str(plain))[0] if h.startswith(("##contig=", "##reference="))] == [
        "##reference=file:///old/b37.fa", "##contig=<ID=1,length=5>", "##contig=<ID=2,length=7>"]
    # the command line tool, and a FASTA whose dictionary lies next to it
    (tmp_path / "ref.fa").write_text(">chr1\nACGT\n")
    import calitas_amd.__main__ as M
    cli = tmp_path / "cli.vcf.gz"
    assert M.main(["PrepareVcf", "-i", str(src), "-o", str(cli), "-d", str(tmp_path / "ref.fa")]) == 0
    assert PV.read_vcf(str(cli))[0] == hdr
    # a header without contig lines gets them in front of #CHROM; a dictionary without an assembly cannot name the reference
    bare = tmp_path / "bare.vcf"
    bare.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n1\t5\t.\tA\tC\t.\tPASS\tAF=0.5\n")
    PV.prepare_vcf([str(bare)], str(out), dict_path=str(d))
    h2 = PV.read_vcf(str(out))[0]
    assert h2[1:4] == hdr[1:4] and h2[4] == "##reference=hg38" and h2[5].startswith("#CHROM")
    (tmp_path / "x.fai").write_text("chr1\t100\t6\t60\t61\n")
    with pytest.raises(ValueError, match="assembly"):
        PV.prepare_vcf([str(bare)], str(out), dict_path=str(tmp_path / "x.fai"))


# ---- the library's own VCF reader and MD5 (calitas_vcf_records / calitas_vcf_identifier) on a host-only context ---------------------

def _hostctx():
    import calitas_amd as C
    return C, C.Context(-1)


def _float32(x):
    with np.errstate(over="ignore"):                            # (1.8e308 is infinity as a float, in both readers)
        return float(np.float32(float(x)))


@pytest.mark.parametrize("n_bytes", [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 128, (1 << 20) - 1, (1 << 20) + 3, 3 * (1 << 20) + 64])
def test_library_md5_equals_hashlib(tmp_path, n_bytes):
    """ReferenceHit's VCF identifier "name:md5" (ReferenceHit.scala:175-183): the library's MD5 (64 steps written out, RFC 1321 padding at
    every length class around the 56- and 64-byte boundaries and across its 1 MB read buffer) against hashlib's."""
    C, ctx = _hostctx()
    p = tmp_path / ("f%d.vcf" % n_bytes)
    p.write_bytes(np.random.default_rng(n_bytes).integers(0, 256, n_bytes, dtype=np.uint8).tobytes())
    try:
        assert ctx.vcf_identifier(p) == PV.vcf_identifier(p)
        with pytest.raises(C.CalitasError):
            ctx.vcf_identifier(tmp_path / "absent.vcf")
    finally:
        ctx.close()


def _compare_with_python_reader(ctx, path, chrom=None):
    got = ctx.vcf_records(path, chrom)
    _, want = PV.read_vcf(str(path), chrom)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        # (an AF comes back as the %.9g of the float the search keeps: nine digits name a float exactly)
        assert g[:6] + ([_float32(a) for a in g[6]],) == (w.chrom, w.pos, w.end, w.id, w.ref, w.alts, [_float32(a) for a in w.afs]), (g, w.line)
    return got


def test_library_vcf_reader_on_edge_cases(tmp_path):
    """calitas_vcf_records (what calitas_search_variants reads: mapped file, records parsed in place, one ALT / AF held in place, AF through
    Clinger's fast path) against the package's Python reader: headers, short lines, '.' ids, several ALTs, AF lists with '.' and empty
    entries, exponents, long digit strings, signs, END, no INFO column, no trailing newline, the --chrom filter, gzip."""
    C, ctx = _hostctx()
    lines = [
        "##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO",
        "chr1\t10\t.\tA\tC\t.\tPASS\tAF=0.25",
        "chr1\t20\trs1\tAC\tA,ACG,T\t.\t.\tDP=3;AF=0.5,.,1e-3;END=25",
        "chr1\t30\trs2\tG\tT\t.\tPASS\tAF=",
        "chr1\t40\trs3\tG\tT\t.\tPASS\tAF=.",
        "chr1\t50\trs4\tG\tT\t.\tPASS\tAF=0.000001234,5,-0.5,+0.75,00.5,.5,5.",
        "chr1\t60\trs5\tG\tT\t.\tPASS\tAF=0.1234567890123456789,123456789012345678,1.7976931348623157e308,4.9e-324,1e-46",
        "chr1\t70\trs6\tG\tT\t.\tPASS\tAF=0.1;AF=0.2",
        "chr1\t80\trs7\tG\tT\t.\tPASS",
        "chr1\t90\trs8\tG\tT",
        "short\tline",
        "chr2\t5\t.\tTTT\tT\t.\tPASS\tEND=7;AF=0.33333334",
        "chr2\t15\tlong_identifier_of_more_than_fifteen_characters\tACGTACGTACGTACGTACGT\tA,ACGTACGTACGTACGTACGTACGT\t.\tPASS\tAF=0.016,0.984",
        "chr2\t25\t.\tA\t\t.\tPASS\tAF=0.5",
    ]
    p = tmp_path / "edge.vcf"
    p.write_text("\n".join(lines))                              # (no newline behind the last record)
    try:
        got = _compare_with_python_reader(ctx, p)
        assert len(got) == 12 and got[-1][5] == [""] and [_float32(a) for a in got[1][6]] == [0.5, _float32(1e-3)]
        assert len(_compare_with_python_reader(ctx, p, "chr2")) == 3
        assert _compare_with_python_reader(ctx, p, "chrX") == []
        import gzip
        gz = tmp_path / "edge.vcf.gz"
        with gzip.open(gz, "wt") as f:
            f.write("\n".join(lines) + "\n")
        assert _compare_with_python_reader(ctx, gz) == got
        empty = tmp_path / "empty.vcf"
        empty.write_text("")
        assert ctx.vcf_records(empty) == []
        with pytest.raises(C.CalitasError):
            ctx.vcf_records(tmp_path / "absent.vcf")
    finally:
        ctx.close()


def test_library_vcf_reader_in_waves(tmp_path):
    """A file of several waves (the reader parses 1-16 MB at a time and publishes the records wave by wave, every worker taking the lines
    that start in its share): 120 000 records with random AF spellings, record for record against the Python reader."""
    C, ctx = _hostctx()
    rng = np.random.default_rng(77)
    p = tmp_path / "big.vcf"
    with open(p, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        pos = 0
        for i in range(120000):
            pos += int(rng.integers(1, 50))
            k = int(rng.integers(0, 6))
            af = ("%.*f" % (int(rng.integers(1, 12)), rng.random()) if k < 3 else "%.3e" % rng.random() if k == 3 else
                  "%d" % rng.integers(0, 3) if k == 4 else "%.17g" % rng.random())
            alts = ",".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 3))))
            f.write("chr%d\t%d\trs%d\t%s\t%s\t.\tPASS\tDP=%d;AF=%s%s\n" % (1 + i // 40000, pos, i, "ACGT"[i % 4], alts, i % 97, af,
                                                                      ",%s" % af if "," in alts else ""))
    assert os.path.getsize(p) > 4 << 20
    try:
        got = _compare_with_python_reader(ctx, p)
        assert len(got) == 120000
        assert len(_compare_with_python_reader(ctx, p, "chr2")) == 40000
    finally:
        ctx.close()
