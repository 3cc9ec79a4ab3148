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


def test_hits_text_property_checker(tmp_path):
    """oracle/check_hits.cpp (what the full-size GPU tests hold a 21.8 GB hits.txt against): the oracle's own output of a dense search
    with a VCF has no row out of ReferenceHit.sort order (RH:284) and no two consecutive kept hits of a group overlapping by maxOverlap
    or more (SR:653-675); swapped rows, a duplicated row and a cut column are counted; any thread count gives the same answer."""
    import numpy as np
    rng = np.random.default_rng(5)
    contigs = [("c1", "".join(rng.choice(list("ACGT"), size=30000))), ("c2", "".join(rng.choice(list("ACGT"), size=9000)))]
    fa = write_fasta(str(tmp_path / "p.fa"), contigs)
    variants = []
    for name, seq in contigs:
        for pos in range(150, len(seq) - 100, 130):
            rb = seq[pos - 1]
            variants.append((name, pos, "rs%d" % len(variants), rb, ["ACGT"[("ACGT".index(rb) + 1) % 4]]))
    vcf = write_vcf(str(tmp_path / "p.vcf"), variants, [[0.1]] * len(variants))
    header, rows, _ = O.search_reference_vcf(fa, vcf, "CTTGCCCCACAGGGCAGTAA", "a", d=8, p=0, g=3)
    assert len(rows) > 300 and sum(1 for r in rows if r["variant_description"]) > 10
    lines = ["\t".join(r[h] for h in header) for r in rows]
    text = ("\t".join(header) + "\n" + "\n".join(lines) + "\n").encode()
    names = [n for n, _ in contigs]
    for threads in (1, 3, 16):
        got = O.check_hits_text(text, names, 10, threads)
        assert got == dict(rows=len(rows), rows_with_variant=sum(1 for r in rows if r["variant_description"]), out_of_order=0, overlapping=0,
                           malformed=0, threads=threads), threads
    # planted faults
    k = next(i for i in range(len(rows) - 1) if rows[i]["coordinate_start"] != rows[i + 1]["coordinate_start"] and rows[i]["chromosome"] == rows[i + 1]["chromosome"])
    swapped = lines[:k] + [lines[k + 1], lines[k]] + lines[k + 2:]
    bad = O.check_hits_text(("\t".join(header) + "\n" + "\n".join(swapped) + "\n").encode(), names, 10, 4)
    assert bad["out_of_order"] >= 1
    dup = lines[:k + 1] + [lines[k]] + lines[k + 1:]                    # the same hit twice: overlaps itself by its whole length
    bad = O.check_hits_text(("\t".join(header) + "\n" + "\n".join(dup) + "\n").encode(), names, 10, 4)
    assert bad["overlapping"] >= 1 and bad["out_of_order"] == 0
    cutcol = lines[:k] + ["\t".join(lines[k].split("\t")[:-1])] + lines[k + 1:]
    bad = O.check_hits_text(("\t".join(header) + "\n" + "\n".join(cutcol) + "\n").encode(), names, 10, 4)
    assert bad["malformed"] == 1
    bad = O.check_hits_text(text, names[::-1], 10, 4)                    # another dictionary order: the contigs are out of order
    assert bad["out_of_order"] >= 1
