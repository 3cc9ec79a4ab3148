"""Pins the CPU oracle against every known-answer vector the reference's own unit tests hold for the hot path
(K1-K26 SequentialGuideAlignerTest, G1-G6 GuideAlignmentTest, E1-E3/E5 SearchReferenceTest).
Runs under both settings of the two switches the vectors cannot distinguish (U1, U2)."""
import json
import os

import pytest

import oracle_lib as O
from fasta_util import expand, write_fasta

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SGA = json.load(open(os.path.join(GOLD, "kat_sga.json")))
GA = json.load(open(os.path.join(GOLD, "kat_ga.json")))
SR = json.load(open(os.path.join(GOLD, "kat_sr.json")))

SWITCHES = [0, 1, 2, 3, 4, 6]          # bit 0: U1-b one alignment per matrix, bit 1: U2 by score, bit 2: U1-c no shared traceback cells (oracle only: a probe, tests/golden/u_probe.json)


def rc(s):
    comp = dict(zip("ACGTacgtNn", "TGCAtgcaNn"))
    return "".join(comp[c] for c in reversed(s))


@pytest.mark.parametrize("sw", SWITCHES)
@pytest.mark.parametrize("case", SGA["align"], ids=lambda c: c["id"])
def test_align_kats(case, sw):
    alns = O.align(case["guide"], case["target"], case["d"], case["g"], case["p"], case["D"], O=case.get("O", 0),
                   off=case.get("off", 0), switches=sw)
    e = case["expect"]
    if "size" in e:
        assert len(alns) == e["size"]
        if e["size"] == 0:
            return
    a = alns[0]
    for k in ("strand", "start", "end", "gstart", "gend", "cigar", "padded_guide", "padded_target"):
        if k in e:
            assert a[k] == e[k], (case["id"], k, a)


@pytest.mark.parametrize("sw", SWITCHES)
def test_revcomp_symmetry_k11(sw):
    c = SGA["revcomp_symmetry"]
    for t in c["targets"]:
        f = O.align_best(c["guide"], t, switches=sw)
        r = O.align_best(rc(c["guide"]), rc(t), switches=sw)
        for k in ("score", "guide_mm", "guide_gaps", "pam_mm", "pam_gaps"):
            assert f[k] == r[k], (t, k, f, r)


@pytest.mark.parametrize("sw", SWITCHES)
@pytest.mark.parametrize("case", SGA["align_best"], ids=lambda c: c["id"])
def test_align_best_kats(case, sw):
    a = O.align_best(case["guide"], case["target"], aux=case.get("aux", ()), switches=sw)
    e = case["expect"]
    for k in ("score", "guide", "cigar", "start", "mismatches", "gap_bases"):
        if k in e:
            assert a[k] == e[k], (case["id"], k, a)
    if "pam_mms_plus_gaps" in e:
        assert a["pam_mm"] + a["pam_gaps"] == e["pam_mms_plus_gaps"]


@pytest.mark.parametrize("sw", SWITCHES)
def test_align_to_ref_best_kats(sw):
    got = {}
    for case in SGA["align_to_ref_best"]:
        contig = SGA["contigs"][case["chrom"]]
        a = O.align_to_ref_best(case["guide"], case["chrom"], contig, case["pos"], switches=sw)
        got[case["id"]] = a
        e = case["expect"]
        for k in ("start", "end", "gstart", "gend", "strand", "padded_alignment", "mismatches", "gap_bases"):
            if k in e:
                assert a[k] == e[k], (case["id"], k, a)
        if e.get("all_match"):
            assert set(a["padded_alignment"]) == {"|"}
        if e.get("padded_guide_equals_target"):
            assert a["padded_guide"] == a["padded_target"]
        if "score_ge" in e:
            assert a["score"] >= e["score_ge"]
        if "same_score_and_alignment_as" in e:
            o = got[e["same_score_and_alignment_as"]]
            assert a["score"] == o["score"] and a["padded_alignment"] == o["padded_alignment"]
        assert a["chrom"] == case["chrom"]


@pytest.mark.parametrize("case", GA["cases"], ids=lambda c: c["id"])
def test_guide_alignment_counters(case):
    assert O.guide_alignment(case["pg"], case["pa"], case["pt"], case["start"], case["end"], case["strand"]) == case["expect"]


def _fasta(tmp_path, key):
    contigs = [(name, expand(spec)) for name, spec in SR[key]["contigs"]]
    return write_fasta(str(tmp_path / (key + ".fa")), contigs)


@pytest.mark.parametrize("sw", SWITCHES)
@pytest.mark.parametrize("case", SR["cases"], ids=lambda c: c["id"])
def test_search_reference_kats(case, sw, tmp_path):
    fa = _fasta(tmp_path, case["fasta"])
    header, rows, _ = O.search_reference(fa, case["guide"], switches=sw)
    e = case["expect"]
    assert len(header) == 34
    assert len(rows) == e["n"]
    for k in ("chromosome", "padded_alignment"):
        if k in e:
            assert [r[k] for r in rows] == e[k]
    for k in ("coordinate_start", "total_mm_plus_gaps"):
        if k in e:
            assert [int(r[k]) for r in rows] == e[k]


def test_window_iterator_e5(tmp_path):
    w = SR["window_iterator"]
    fa = _fasta(tmp_path, w["fasta"])
    wins = O.windows(fa, w["window"], w["step"])
    assert len(wins) == 59  # Range(0, 24999, 426)
    assert wins[0] == ("chr1", 1, 451, 451)
    assert wins[-1][2] == 25000


def test_align_to_reference_tool_restatement_is_consistent(tmp_path):
    """The file-level AlignToReference restatement (A2R:95-146; no reference test pins it) must agree with the pinned
    alignToRefBest restatement (K23-K26) task by task, and print the Option-typed flags like Scala does."""
    from fasta_util import write_fasta
    contig = SGA["contigs"]
    name = sorted(contig)[0]
    fa = write_fasta(str(tmp_path / "a2r.fa"), [(name, contig[name])])
    tasks = [("q%d" % i, g, name, pos) for i, (g, pos) in enumerate([("CTTGCCCCACAGGGCAGTAAnrg", 40), ("GATACGTCTCGTACTGTnrg", 100),
                                                                       ("tttvAACCAACCAACCGGTTACGT", 60)])]
    inp = tmp_path / "tasks.tsv"
    inp.write_text("id\tquery\tchrom\tposition\n" + "".join("%s\t%s\t%s\t%d\n" % t for t in tasks))
    header, rows = O.align_to_reference(fa, str(inp))
    assert len(rows) == len(tasks) and header[0] == "guide_id" and len(header) == 34
    by_id = {r["guide_id"]: r for r in rows}
    for tid, g, chrom, pos in tasks:
        w = O.align_to_ref_best(g, chrom, contig[name], pos)
        r = by_id[tid]
        assert (int(r["score"]), r["cigar"], r["strand"], int(r["coordinate_start"]), int(r["coordinate_end"])) == \
               (w["score"], w["cigar"], w["strand"], w["gstart"], w["gend"])
        assert r["aligner"] == "CALITAS:AlignToReference" and "max-guide-diffs=None" in r["aligner_other_parameters"]
    _, rows = O.align_to_reference(fa, str(inp), limits=(5, 1, 10))
    assert all("max-guide-diffs=Some(5)" in r["aligner_other_parameters"] and "max-overlap=Some(10)" in r["aligner_other_parameters"] for r in rows)
    keys = [(int(r["coordinate_start"]), r["strand"], -int(r["score"])) for r in rows]
    assert keys == sorted(keys)


def test_u_probe_vectors():
    """tests/golden/u_probe.json (made by make_u_probe.py): the inputs an integrator with fgbio at hand runs once to settle U1 / U2.
    The oracle reproduces each listed output under the matching switch, the listed outputs differ pairwise, and the product's two
    switches map to the first two readings (the third exists in the oracle only)."""
    P = json.load(open(os.path.join(GOLD, "u_probe.json")))
    u1, u2 = P["U1"], P["U2"]
    sw_of = {"U1-a": 0, "U1-b": 1, "U1-c": 4, "U2-a": 0, "U2-b": 2}
    for u in (u1, u2):
        seen = []
        for label, want in u["expect"].items():
            sw = sw_of[label.split(" ")[0]]
            assert O.glocal(u["query"], u["target"], u["min_score"], switches=sw) == want, label
            assert want not in seen
            seen.append(want)
        rows = []
        for key in [k for k in u["align"] if k.startswith("U")]:
            d = u["max_guide_diffs"]
            got = [{k: r[k] for k in ("strand", "start", "end", "score", "cigar")}
                   for r in O.align(u["query"], u["target"], d, 0, 0, d, O=u["align"]["max_overlap"], switches=sw_of[key])]
            assert got == u["align"][key], key
            assert got not in rows
            rows.append(got)
    assert len(u1["expect"]) == 3 and len(u2["expect"]) == 2
    assert u1["min_score"] == 60 * len(u1["query"]) - 122 * u1["max_guide_diffs"]      # minGuideScore, SGA:239-243
