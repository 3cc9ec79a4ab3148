#!/usr/bin/env python3
"""Transcribes the known-answer vectors of the reference's own unit tests into JSON fixtures.

Run once in the build container (the reference tree is not available on the GPU box):

    python tests/golden/make_kat_fixtures.py

Only DATA is taken from the reference tests: the sequence literals of the in-memory FASTA that
SequentialGuideAlignerTest.scala:12-44 builds are pulled out with a regex (so they cannot be mistyped); the
inputs / expected values of every test case were transcribed by hand, each with the line range it comes from.
Outputs: kat_sga.json (K1-K26), kat_ga.json (G1-G6), kat_sr.json (E1-E3, E5), kat_variants.json (V1-V9, E4).
"""
import json
import os
import re

REF = "/root/reference/calitas/src/test/scala/com/editasmedicine/aligner"
HERE = os.path.dirname(os.path.abspath(__file__))


def rc(s):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    return "".join(comp[c] for c in reversed(s))


def sga_reference_contigs():
    text = open(os.path.join(REF, "SequentialGuideAlignerTest.scala")).read()
    block = text[text.index('builder.add("chr1")'):text.index("val path = builder.toTempFile()")]
    chr1_part, chr2_part = block.split('builder.add("chr2")')
    lines = lambda part: re.findall(r'\.add\("([A-Za-z]+)"\)', part)
    chr1 = "".join(lines(chr1_part))
    chr2 = "".join(lines(chr2_part))
    assert len(chr1) == 2400 and len(chr2) == 42, (len(chr1), len(chr2))
    return chr1, chr2


def main():
    chr1, chr2 = sga_reference_contigs()
    q = "GATACGTCTCGTACTGTnrg"
    sga = {
        "source": "calitas/src/test/scala/com/editasmedicine/aligner/SequentialGuideAlignerTest.scala",
        "contigs": {"chr1": chr1, "chr2": chr2},
        "align": [
            {"id": "K1", "lines": "51-65", "guide": "AACCAACC", "target": "TTTTAACCAACCGGGG", "d": 0, "p": 0, "g": 0, "D": 0,
             "expect": {"size": 1, "strand": "+", "start": 4, "end": 12, "gstart": 4, "gend": 12, "cigar": "8=",
                        "padded_guide": "AACCAACC", "padded_target": "AACCAACC"}},
            {"id": "K2", "lines": "67-81", "guide": "GGTTGGTT", "target": "TTAACCAACCGGGG", "d": 0, "p": 0, "g": 0, "D": 0,
             "expect": {"size": 1, "strand": "-", "start": 2, "end": 10, "gstart": 2, "gend": 10, "cigar": "8=",
                        "padded_guide": "GGTTGGTT", "padded_target": "GGTTGGTT"}},
            {"id": "K3", "lines": "83-97", "guide": "GGTTGGTT", "target": "AGCCAACC", "d": 1, "p": 0, "g": 0, "D": 1,
             "expect": {"size": 1, "strand": "-", "start": 0, "end": 8, "gstart": 0, "gend": 8, "cigar": "6=1X1=",
                        "padded_guide": "GGTTGGTT", "padded_target": "GGTTGGCT"}},
            {"id": "K4", "lines": "99-112", "guide": "AACCAACCAACCnrg", "target": "CCAACCAACCAACCGAGGGGGG", "d": 0, "p": 0, "g": 1, "D": 1,
             "expect": {"size": 1, "strand": "+", "start": 2, "end": 17, "gstart": 2, "gend": 14, "cigar": "15=",
                        "padded_guide": "AACCAACCAACCnrg", "padded_target": "AACCAACCAACCGAG"}},
            {"id": "K5", "lines": "114-127", "guide": "AACCAACCAACCnrg", "target": "CCCTGGGTTGGTTGGTTGGGGGG", "d": 0, "p": 0, "g": 1, "D": 1,
             "expect": {"size": 1, "strand": "-", "start": 2, "end": 17, "gstart": 5, "gend": 17, "cigar": "15=",
                        "padded_guide": "AACCAACCAACCnrg", "padded_target": "AACCAACCAACCCAG"}},
            {"id": "K6", "lines": "129-142", "guide": "tttvAACCAACCAACC", "target": "CCTTTGAACCAACCAACCGAGG", "d": 0, "p": 0, "g": 1, "D": 1,
             "expect": {"size": 1, "strand": "+", "start": 2, "end": 18, "gstart": 6, "gend": 18, "cigar": "16=",
                        "padded_guide": "tttvAACCAACCAACC", "padded_target": "TTTGAACCAACCAACC"}},
            {"id": "K7", "lines": "144-157", "guide": "tttvAACCAACCAACC", "target": "CC" + rc("TTTGAACCAACCAACC") + "GAGG",
             "d": 0, "p": 0, "g": 1, "D": 1,
             "expect": {"size": 1, "strand": "-", "start": 2, "end": 18, "gstart": 2, "gend": 14, "cigar": "16=",
                        "padded_guide": "tttvAACCAACCAACC", "padded_target": "TTTGAACCAACCAACC"}},
            {"id": "K8", "lines": "159-172", "guide": "tttvAACCAACCAACC", "target": "CCTTTGAACCAACCAAGCGAGG", "d": 1, "p": 0, "g": 0, "D": 1,
             "expect": {"size": 1, "strand": "+", "start": 2, "end": 18, "gstart": 6, "gend": 18, "cigar": "14=1X1=",
                        "padded_guide": "tttvAACCAACCAACC", "padded_target": "TTTGAACCAACCAAGC"}},
            {"id": "K9", "lines": "174-187", "guide": "tttvAACCAACCAACC", "target": "CC" + rc("TTTGAACCAACCAAGC") + "GAGG",
             "d": 1, "p": 0, "g": 0, "D": 1,
             "expect": {"size": 1, "strand": "-", "start": 2, "end": 18, "gstart": 2, "gend": 14, "cigar": "14=1X1=",
                        "padded_guide": "tttvAACCAACCAACC", "padded_target": "TTTGAACCAACCAAGC"}},
            # K10 (189-220): targetOffset=1000, all limits 0; only start/end of r1..r4 (+ r1's guide offsets) are asserted
            {"id": "K10-r1", "lines": "197-201", "guide": "gggTTTTT", "target": "AGAGAGAGAGGGTTTTTGGGAGAGAGAGAGAGAG", "d": 0, "p": 0, "g": 0,
             "D": 0, "off": 1000, "expect": {"head": True, "start": 1009, "end": 1017, "gstart": 1012, "gend": 1017}},
            {"id": "K10-r2", "lines": "203-205", "guide": "TTTTTggg", "target": "AGAGAGAGAGGGTTTTTGGGAGAGAGAGAGAGAG", "d": 0, "p": 0, "g": 0,
             "D": 0, "off": 1000, "expect": {"head": True, "start": 1012, "end": 1020}},
            {"id": "K10-r3", "lines": "209-211", "guide": "gggTTTTT", "target": "AGAGAGAGACCCAAAAACCCAGAGAGAGAGAGAG", "d": 0, "p": 0, "g": 0,
             "D": 0, "off": 1000, "expect": {"head": True, "start": 1012, "end": 1020}},
            {"id": "K10-r4", "lines": "215-217", "guide": "TTTTTggg", "target": "AGAGAGAGACCCAAAAACCCAGAGAGAGAGAGAG", "d": 0, "p": 0, "g": 0,
             "D": 0, "off": 1000, "expect": {"head": True, "start": 1009, "end": 1017}},
            {"id": "K13", "lines": "242-248", "guide": "yttnAGGAAACTTCTGGCAGGACC",
             "target": "GTTAGTTCCAGATCTTGAGGAAGCTATCCCAGGACCCTGTCGCCACAGCCA", "d": 5, "g": 1, "p": 1, "D": 7, "O": 10,
             "expect": {"size": 1, "start": 13}},
            {"id": "K26-a", "lines": "379-385", "guide": q, "target": "GAAACGTTTCGTACTGTAAC", "d": 2, "g": 0, "p": 1, "D": 3,
             "expect": {"size": 1}},
            {"id": "K26-b", "lines": "387-388", "guide": q, "target": "GAAACGTTTCGTACTGTAAC", "d": 2, "g": 0, "p": 1, "D": 2,
             "expect": {"size": 0}},
        ],
        # K11 (222-233): alignBest(guide, t) vs alignBest(rc(guide), rc(t)) agree on score and the four counters
        "revcomp_symmetry": {"id": "K11", "lines": "222-233", "guide": "AATTCcgg",
                             "targets": ["AATTCCGG", "AGTTCCGG", "AAATTCCGG", "AATTCCGAG", "AATTCCTG"]},
        "align_best": [
            {"id": "K12", "lines": "235-240", "guide": "AACCGGTTnrg", "target": "nnnnnnnnnnn",
             "expect": {"score": 8 * -60 + 3 * -130}},
            {"id": "K14", "lines": "250-256", "guide": "AACCGGTTACGTnrg", "aux": ["ntg"], "target": "AACCGGTTACGTTTG",
             "expect": {"guide": "AACCGGTTACGTntg", "pam_mms_plus_gaps": 0}},
            {"id": "K15", "lines": "258-263", "guide": "AACCGGTTACGTnnn", "aux": ["nnnn", "nn"], "target": "AACCGGTTACGTAAAAAAA",
             "expect": {"guide": "AACCGGTTACGTnnnn"}},
            {"id": "K16", "lines": "265-271", "guide": "AACCGGTTACGTacc", "aux": ["cccc"], "target": "AACCGGTTACGTACCCC",
             "expect": {"guide": "AACCGGTTACGTcccc", "cigar": "12=1D4="}},
            {"id": "K24", "lines": "361-368", "guide": q,
             "target": q.replace("GATA", "GATT").replace("nrg", "AAG") + "TTTTT" + q.replace("TCTC", "TCTCC").replace("nrg", "AAG"),
             "expect": {"start": 0, "mismatches": 1, "gap_bases": 0}},
            {"id": "K25", "lines": "370-377", "guide": q,
             "target": q.replace("TCTC", "TCTCC").replace("nrg", "AAG") + "NNNNN" + q.replace("TCTC", "TCT").replace("nrg", "AAG"),
             "expect": {"start": 0, "mismatches": 0, "gap_bases": 1}},
        ],
        "align_to_ref_best": [
            {"id": "K17", "lines": "274-285", "guide": chr1[49:69], "chrom": "chr1", "pos": 65,
             "expect": {"start": 49, "end": 69, "strand": "+", "all_match": True, "padded_guide_equals_target": True, "score_ge": 0}},
            {"id": "K18", "lines": "287-296", "guide": chr1[49:69].replace("T", "U"), "chrom": "chr1", "pos": 65,
             "expect": {"same_score_and_alignment_as": "K17"}},
            {"id": "K19", "lines": "298-308", "guide": rc(chr1[49:69]), "chrom": "chr1", "pos": 65,
             "expect": {"start": 49, "end": 69, "strand": "-", "all_match": True, "score_ge": 0}},
            {"id": "K20", "lines": "310-321", "guide": "GAGAATTGTTTGAACCCAGGNGG", "chrom": "chr1", "pos": 515,
             "expect": {"start": 500, "end": 523, "strand": "+", "padded_alignment": "||||||||.||||||||||||||", "mismatches": 1}},
            {"id": "K21", "lines": "323-337", "guide": "TCAGTGCCTGCGCCGCGCTCGCTCCCnrycwshdm", "chrom": "chr1", "pos": 1820,
             "expect": {"start": 1800, "end": 1835, "gstart": 1800, "gend": 1826, "strand": "+",
                        "padded_alignment": "||||||||||||||||||||||||||||||.||||", "mismatches": 1}},
            {"id": "K22", "lines": "339-349", "guide": "AGGCTGGGGCGGTCGCTCGCNGG", "chrom": "chr1", "pos": 1510,
             "expect": {"start": 1500, "end": 1523, "strand": "-", "padded_alignment": "|||||||~|||||||||~||||||"}},
            {"id": "K23", "lines": "351-359", "guide": q, "chrom": "chr2", "pos": 22,
             "expect": {"start": 0, "end": 20, "gap_bases": 0, "mismatches": 2}},
        ],
    }
    json.dump(sga, open(os.path.join(HERE, "kat_sga.json"), "w"), indent=1)

    ga = {
        "source": "calitas/src/test/scala/com/editasmedicine/aligner/GuideAlignmentTest.scala",
        "columns": ["guide_mm", "guide_gaps", "guide_mm_plus_gaps", "pam_mm", "pam_gaps", "pam_mm_plus_gaps", "mismatches", "gap_bases",
                    "edits", "gstart", "gend"],
        "cases": [
            {"id": "G1", "lines": "11-28", "pg": "GCTGACTGCATGACTATAnrg", "pa": "|||||||||||||||||||||", "pt": "GCTGACTGCATGACTATAnrg",
             "start": 1, "end": 21, "strand": "+", "expect": [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 18]},
            {"id": "G2", "lines": "30-47", "pg": "GCTGACT-GCATGACTATAnrg", "pa": "||.||||~|||.||~|||||||", "pt": "GCAGACTCGCACGA-TATAnrg",
             "start": 1, "end": 21, "strand": "+", "expect": [2, 2, 4, 0, 0, 0, 2, 2, 4, 1, 18]},
            {"id": "G3", "lines": "49-66", "pg": "GCTGACTGCATGACTATAnngrrn", "pa": "|||||||||||||||||||~||.|", "pt": "GCTGACTGCATGACTATAC-GATT",
             "start": 1, "end": 23, "strand": "+", "expect": [0, 0, 0, 1, 1, 2, 1, 1, 2, 1, 18]},
            {"id": "G4", "lines": "68-85", "pg": "GCTGAC---TGCATGACTATAnrg", "pa": "||||||~~~||||~~|||||||||", "pt": "GCTGACGGGTGCA--ACTATACGG",
             "start": 1, "end": 22, "strand": "-", "expect": [0, 5, 5, 0, 0, 0, 0, 5, 5, 4, 22]},
            {"id": "G5", "lines": "87-104", "pg": "---GCTGACTGCATGACTATAnrg--", "pa": "~~~|||||||||||||||||||||~~", "pt": "TGTGCTGACTGCATGACTATACGGCC",
             "start": 1, "end": 26, "strand": "+", "expect": [0, 3, 3, 0, 2, 2, 0, 5, 5, 4, 21]},
            {"id": "G6", "lines": "106-123", "pg": "GCTGACTGCATGACTATA--nrg", "pa": "||||||||||||||||||~~|||", "pt": "GCTGACTGCATGACTATATTCGG",
             "start": 1, "end": 23, "strand": "+", "expect": [0, 2, 2, 0, 0, 0, 0, 2, 2, 1, 18]},
        ],
    }
    json.dump(ga, open(os.path.join(HERE, "kat_ga.json"), "w"), indent=1)

    perfect = "ACGTACATGCTCGATACGACGccgaat".upper()
    mismatched = "ACGcACAcGCcCGAcACGACGccgaat".upper()
    sr = {
        "source": "calitas/src/test/scala/com/editasmedicine/aligner/SearchReferenceTest.scala",
        # contigs as lists of [unit, repeat] exactly as the ReferenceSetBuilder calls at SRT:17-33 / 76-77 / 45
        "fasta_main": {"lines": "17-33", "contigs": [
            ["chr1", [["N", 5000], ["AATAT", 1000], ["N", 5000]]],
            ["chr2", [["N", 3000], [perfect, 1], ["GT", 500], [mismatched, 1], ["CA", 500], ["N", 3000]]]]},
        "fasta_short": {"lines": "76-77", "contigs": [
            ["ref", [["GTGCGTGACTTGAAGTCTCAGTATACCTTGCCACACGTTGCAGGTTGCCC", 1]]],
            ["alt", [["GTGCGTGACTTGAAGTCTCAGTATgaaaTTGCCACACGTTGCAGGTTGCCC", 1]]]]},
        "fasta_windows": {"lines": "44-46", "contigs": [["chr1", [["ACGTC", 5000]]]]},
        "cases": [
            {"id": "E1", "lines": "51-62", "fasta": "fasta_main", "guide": "ACGTACATGCTCGATACGACGnngrrn",
             "expect": {"n": 2, "chromosome": ["chr2", "chr2"], "coordinate_start": [3000, 4000 + len(perfect)],
                        "total_mm_plus_gaps": [0, 4]}},
            {"id": "E2", "lines": "64-69", "fasta": "fasta_main", "guide": "ACGTACATGCTCGATACGACG", "expect": {"n": 2}},
            {"id": "E3", "lines": "71-92", "fasta": "fasta_short", "guide": "GTGACTTGAAGTCTCAGTATA",
             "expect": {"n": 2, "chromosome": ["ref", "alt"], "coordinate_start": [4, 4],
                        "padded_alignment": ["|||||||||||||||||||||", "||||||||||||||||||||."]}},
        ],
        "window_iterator": {"id": "E5", "lines": "43-49", "fasta": "fasta_windows", "window": 451, "step": 426},
    }
    json.dump(sr, open(os.path.join(HERE, "kat_sr.json"), "w"), indent=1)
    ref50 = "CTAGACTGACTGACTAGCACTAGCCGCTTTATATATGCTATGGGACACCG"
    ref79 = "CTAGACTGACTGACTAGCACTAGCCGCTTTATATATGCTAGGCGCTACTGAATGCTATAGCTCTGAGACTGGGACACCG"
    e4_lines = [
        "ACACACACACACACACACACACACACACACACACACACAgcgtcacggtcgagcgattggggAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
        "ACACACACACACACACACACACACACACACACACACACAccccaatcgctcgaccgtgacgcAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
        "ACACACACACACACACACACACACACACACACACACACAcacggtcgagcgattggggAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
        "ACACACACACACACACACACACACACACACACACACACAaatcgctcgaccgtgacgcAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"]
    assert all(len(x) == 100 for x in e4_lines)
    var = {
        "source": "calitas/src/test/scala/com/editasmedicine/aligner/SearchReferenceTest.scala",
        "allele_combos_counts": [   # V1-V3, lines 150-181
            {"lines": "150-153", "counts": [2], "expect": [[0], [1]]},
            {"lines": "150-153", "counts": [3], "expect": [[0], [1], [2]]},
            {"lines": "155-158", "counts": [2, 2], "expect": [[0, 0], [0, 1], [1, 0], [1, 1]]},
            {"lines": "155-158", "counts": [3, 2], "expect": [[0, 0], [0, 1], [1, 0], [1, 1], [2, 0], [2, 1]]},
            {"lines": "160-181", "counts": [3, 2, 3],
             "expect": [[a, b, c] for a in range(3) for b in range(2) for c in range(3)]},
        ],
        "build_variant_window": [   # V4-V7, lines 183-247; variants as [pos, id, ref, alts]
            {"id": "V4", "lines": "183-196", "ref": ref50, "variants": [[20, "rs123", "C", ["G"]]], "alleles": [1], "padding": 15,
             "bases": "ACTGACTGACTAGCAgTAGCCGCTTTATATA".upper(), "cigar": "31M",
             "offsets": [[0, True, 4], [15, True, 19], [20, True, 24], [31, True, 35]]},
            {"id": "V5", "lines": "198-215", "ref": ref50, "variants": [[20, "rs123", "C", ["CGT"]]], "alleles": [1], "padding": 15,
             "bases": "ACTGACTGACTAGCAcgtTAGCCGCTTTATATA".upper(), "cigar": "16M2I15M",
             "offsets": [[0, True, 4], [14, True, 18], [15, True, 19], [16, True, 19], [17, True, 19], [15, False, 19], [16, False, 20],
                         [17, False, 20]]},
            {"id": "V6", "lines": "217-230", "ref": ref50, "variants": [[20, "rs123", "CTA", ["C"]]], "alleles": [1], "padding": 15,
             "bases": "ACTGACTGACTAGCAcGCCGCTTTATATATG".upper(), "cigar": "16M2D15M",
             "offsets": [[0, True, 4], [15, True, 19], [16, True, 22]]},
            {"id": "V7", "lines": "232-247", "ref": ref79,
             "variants": [[10, "snp", "C", ["T"]], [20, "ins", "C", ["CG"]], [30, "del", "TAT", ["T"]]], "alleles": [1, 1, 1], "padding": 15,
             "bases": "CTAGACTGAtTGACTAGCAcgTAGCCGCTTtATATGCTAGGCGCTA".upper(), "cigar": "20M1I10M2D15M", "offsets": []},
        ],
        "allele_combos_variants": [   # V8-V9, lines 249-295; expected sets as lists of "id=allele" (order of the sets is not asserted)
            {"lines": "249-255", "variants": [[20, "snp", "A", ["C"]]], "max": 10, "expect": [["snp=1"]]},
            {"lines": "257-266", "variants": [[20, "snp", "A", ["C", "G", "T"]]], "max": 10, "expect": [["snp=1"], ["snp=2"], ["snp=3"]]},
            {"lines": "268-284", "variants": [[20, "a", "A", ["C"]], [25, "b", "C", ["T"]], [30, "c", "G", ["A"]]], "max": 10,
             "expect": [["a=1"], ["b=1"], ["c=1"], ["a=1", "b=1"], ["a=1", "c=1"], ["b=1", "c=1"], ["a=1", "b=1", "c=1"]]},
            {"lines": "286-295", "variants": [[20, "a", "A", ["C"]], [25, "b", "C", ["T"]], [30, "c", "G", ["A"]]], "max": 2, "expect_size": 1},
            {"lines": "286-295", "variants": [[20, "a", "A", ["C"]], [25, "b", "C", ["T"]], [30, "c", "G", ["A"]]], "max": 3, "expect_size": 7},
        ],
        "e4": {"id": "E4", "lines": "94-147", "guide": "GCGTCACGGTCGAGCGATTGnrg", "chr1": "".join(e4_lines).upper(),
               "variants": [[239, "insGAGGCGT", "A", ["AGAGGCGT"]], [339, "insTCGCCCC", "A", ["ATCGCCCC"]]],
               "params": {"g": 0, "d": 0},
               "expect": {"n": 4, "coordinate_start": [39, 142, 238, 338],
                          "padded_extra_8_bases_5_prime": ["CACACACA", "TTTTTTTT", "ACACAGAG", "TTTTTTTT"],
                          "padded_extra_8_bases_3_prime": ["AAAAAAAA", "TGTGTGTG", "AAAAAAAA", "CGATGTGT"],
                          "ten_bases_5_prime": ["CACACACACA", "TTTTTTTTTT", "ACACACAGAG", "TTTTTTTTTT"],
                          "ten_bases_3_prime": ["GGGAAAAAAA", "GGGTGTGTGT", "GGGAAAAAAA", "GGGCGATGTG"]}},
    }
    json.dump(var, open(os.path.join(HERE, "kat_variants.json"), "w"), indent=1)
    print("wrote kat_sga.json kat_ga.json kat_sr.json kat_variants.json")


if __name__ == "__main__":
    main()
