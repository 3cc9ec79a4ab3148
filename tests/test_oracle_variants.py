"""Pins the oracle's variant path against the reference's own vectors (SearchReferenceTest.scala:94-295: V1-V9 pure
functions and the end-to-end E4 with two insertion variants)."""
import json
import os

import pytest

import oracle_lib as O
from fasta_util import write_fasta

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = json.load(open(os.path.join(GOLD, "kat_variants.json")))


def write_vcf(path, chrom_variants, afs=None):
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"ALT allele frequency\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for i, (chrom, pos, vid, ref, alts) in enumerate(chrom_variants):
            info = "." if afs is None else "AF=" + ",".join(str(x) for x in afs[i])
            f.write("%s\t%d\t%s\t%s\t%s\t.\tPASS\t%s\n" % (chrom, pos, vid or ".", ref, ",".join(alts), info))
    return path


@pytest.mark.parametrize("case", V["allele_combos_counts"], ids=lambda c: "x".join(map(str, c["counts"])))
def test_allele_combos_counts(case):
    assert O.allele_combos(case["counts"]) == case["expect"]


@pytest.mark.parametrize("case", V["build_variant_window"], ids=lambda c: c["id"])
def test_build_variant_window(case):
    queries = [(o, p) for o, p, _ in case["offsets"]]
    bases, cigar, start, offs = O.build_variant_window(case["ref"], case["variants"], case["alleles"], case["padding"], queries)
    assert bases == case["bases"] and cigar == case["cigar"]
    assert offs == [e for _, _, e in case["offsets"]]


@pytest.mark.parametrize("case", V["allele_combos_variants"], ids=lambda c: "%s-max%d" % (c["lines"], c["max"]))
def test_allele_combos_variants(case):
    sets = O.allele_sets(case["variants"], case["max"])
    if "expect" in case:
        assert sorted(map(tuple, sets)) == sorted(map(tuple, case["expect"]))
    else:
        assert len(sets) == case["expect_size"]


def test_e4_flanks_for_ref_and_variant_windows(tmp_path):
    e = V["e4"]
    fa = write_fasta(str(tmp_path / "e4.fa"), [("chr1", e["chr1"])], line_len=100)
    vcf = write_vcf(str(tmp_path / "e4.vcf"), [("chr1", p, i, r, a) for p, i, r, a in e["variants"]])
    _, rows, _ = O.search_reference_vcf(fa, vcf, e["guide"], "test", d=e["params"]["d"], g=e["params"]["g"])
    x = e["expect"]
    assert len(rows) == x["n"]
    assert [int(r["coordinate_start"]) for r in rows] == x["coordinate_start"]
    for k in ("padded_extra_8_bases_5_prime", "padded_extra_8_bases_3_prime", "ten_bases_5_prime", "ten_bases_3_prime"):
        assert [r[k] for r in rows] == x[k], k
    # the two hits that start inside an insertion carry the variant annotation
    assert [r["variant_id"] for r in rows] == ["", "", "insGAGGCGT", "insTCGCCCC"]
    assert rows[2]["genome_build"].endswith("+variants") and not rows[0]["genome_build"].endswith("+variants")
    assert rows[2]["variant_description"] == "insGAGGCGT:238:A>AGAGGCGT:0.000"
