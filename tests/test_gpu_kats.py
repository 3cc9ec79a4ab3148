"""The reference's own known-answer vectors for SequentialGuideAligner (SequentialGuideAlignerTest.scala:51-389, K1-K26)
run through the GPU kernels via calitas_align_windows -- not only through the oracle -- plus randomized agreement with the
oracle on explicit (guide, target) pairs (PairwiseAlignSequences / AlignToReference shape)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SGA = json.load(open(os.path.join(GOLD, "kat_sga.json")))


def rc(s):
    comp = dict(zip("ACGTacgtNn", "TGCAtgcaNn"))
    return "".join(comp[c] for c in reversed(s))


@pytest.fixture(scope="module")
def aligner():
    import calitas_amd as C
    ref = {k: v.encode() for k, v in SGA["contigs"].items()}
    return C.SequentialGuideAligner(ref=ref)


@pytest.fixture(scope="module")
def C():
    import calitas_amd
    return calitas_amd


@pytest.mark.parametrize("case", SGA["align"], ids=lambda c: c["id"])
def test_align_kats_on_gpu(C, aligner, case):
    alns = aligner.align(C.Guide(case["guide"]), case["target"], case["d"], case["g"], case["p"], case["D"], case.get("O", 0),
                         target_offset=case.get("off", 0))
    e = case["expect"]
    if "size" in e:
        assert len(alns) == e["size"]
        if e["size"] == 0:
            return
    a = alns[0]
    got = dict(strand=a.strand, start=a.start_offset, end=a.end_offset, gstart=a.guide_start_offset, gend=a.guide_end_offset,
               cigar=a.cigar, padded_guide=a.padded_guide, padded_target=a.padded_target)
    for k in got:
        if k in e:
            assert got[k] == e[k], (case["id"], k, got)


def test_revcomp_symmetry_k11_on_gpu(C, aligner):
    c = SGA["revcomp_symmetry"]
    for t in c["targets"]:
        f = aligner.align_best(C.Guide(c["guide"]), t)
        r = aligner.align_best(C.Guide(rc(c["guide"])), rc(t))
        assert (f.score, f.guide_mismatches, f.guide_gap_bases, f.pam_mismatches, f.pam_gap_bases) == \
               (r.score, r.guide_mismatches, r.guide_gap_bases, r.pam_mismatches, r.pam_gap_bases), t


@pytest.mark.parametrize("case", SGA["align_best"], ids=lambda c: c["id"])
def test_align_best_kats_on_gpu(C, aligner, case):
    a = aligner.align_best(C.Guide(case["guide"], case.get("aux", ())), case["target"])
    e = case["expect"]
    got = dict(score=a.score, guide=a.guide, cigar=a.cigar, start=a.start_offset, mismatches=a.mismatches, gap_bases=a.gap_bases)
    for k in got:
        if k in e:
            assert got[k] == e[k], (case["id"], k, got)
    if "pam_mms_plus_gaps" in e:
        assert a.pam_mms_plus_gaps == e["pam_mms_plus_gaps"]


def test_align_to_ref_best_kats_on_gpu(C, aligner):
    got = {}
    for case in SGA["align_to_ref_best"]:
        a = aligner.align_to_ref_best(C.Guide(case["guide"]), case["chrom"], case["pos"])
        got[case["id"]] = a
        e = case["expect"]
        vals = dict(start=a.start_offset, end=a.end_offset, gstart=a.guide_start_offset, gend=a.guide_end_offset, strand=a.strand,
                    padded_alignment=a.padded_alignment, mismatches=a.mismatches, gap_bases=a.gap_bases)
        for k in vals:
            if k in e:
                assert vals[k] == e[k], (case["id"], k, vals)
        if e.get("all_match"):
            assert set(a.padded_alignment) == {"|"}
        if e.get("padded_guide_equals_target"):
            assert a.padded_guide == a.padded_target
        if "score_ge" in e:
            assert a.score >= e["score_ge"]
        if "same_score_and_alignment_as" in e:
            o = got[e["same_score_and_alignment_as"]]
            assert (a.score, a.padded_alignment) == (o.score, o.padded_alignment)
        assert a.chrom == case["chrom"]


def _rows(alns):
    return [(a.strand, a.start_offset, a.end_offset, a.guide_start_offset, a.guide_end_offset, a.score, a.cigar, a.guide,
             a.padded_guide, a.padded_alignment, a.padded_target) for a in alns]


def _orows(rows):
    return [(r["strand"], r["start"], r["end"], r["gstart"], r["gend"], r["score"], r["cigar"], r["guide"], r["padded_guide"],
             r["padded_alignment"], r["padded_target"]) for r in rows]


def test_random_pairs_match_oracle(C, aligner):
    """Many (guide, target) tasks in one launch: alignBest-style limits and explicit limits, mixed guide shapes, targets with
    lower case, N, IUPAC, shorter than the guide, empty."""
    from calitas_amd import synth
    rng = np.random.default_rng(2024)
    guides = ["CTTGCCCCACAGGGCAGTAAnrg", "tttvAACCAACCAACCGGTT", "GTGACTTGAAGTCTCAGTATA", "AACCGGTTACGTnnn", "GAGAATTGNTTGAACCCRGGngg",
              "ACGTACATGCTCGATACGACGnngrrn"]
    tasks = []
    for k in range(160):
        g = guides[k % len(guides)]
        G = C.Guide(g)
        n = int(rng.integers(0, 140))
        t = synth.random_bases(rng, n).tobytes().decode() if n else ""
        if n > 40 and k % 3:
            pam = synth.realise(rng, G.pams[0]) if G.pams else ""
            site = synth.mutate(rng, synth.realise(rng, G.guide), int(rng.integers(0, 5)))
            full = (pam + site) if G.pam_is_five_prime else (site + pam)
            if k % 2:
                full = synth.revcomp(full)
            p0 = int(rng.integers(0, max(1, n - len(full))))
            t = (t[:p0] + full + t[p0 + len(full):])[:n]
        if k % 7 == 0 and n > 10:
            t = t[:5] + "NNnn" + t[9:]
        if k % 11 == 0 and n > 20:
            t = t[:12].lower() + "R" + t[13:]
        tasks.append((g, t))
    Gs = [C.Guide(g) for g, _ in tasks]
    # alignBest limits
    res = aligner.align_many(Gs, [t for _, t in tasks])
    for (g, t), alns in zip(tasks, res):
        G = C.Guide(g)
        D = G.protospacer_length + 3 + G.pam_length
        want = O.align(g, t, G.protospacer_length, 3, G.pam_length, D, O=0)
        assert _rows(alns) == _orows(want), (g, t)
    # explicit limits, offsets
    res = aligner.align_many(Gs, [t for _, t in tasks], offsets=[1000 + i for i in range(len(tasks))], max_guide_diffs=4,
                             max_gaps_between_guide_and_pam=2, max_pam_diffs=1, max_total_diffs=6, max_overlap=10)
    for i, ((g, t), alns) in enumerate(zip(tasks, res)):
        want = O.align(g, t, 4, 2, 1, 6, O=10, off=1000 + i)
        assert _rows(alns) == _orows(want), (g, t)


def test_pairwise_align_sequences_tool(C, aligner, tmp_path):
    """PairwiseAlignSequences (PairwiseAlignSequences.scala:42-85): the 11-column TSV."""
    pairs = [("AATTCcgg", "aattccgg"), ("AATTCcgg", "AGTTCCGG"), ("AACCGGTTnrg", "nnnnnnnnnnn"),
             ("GATACGTCTCGTACTGTnrg", "GATTCGTCTCGTACTGTAAGTTTTTGATACGTCTCCGTACTGTAAG")]
    inp = tmp_path / "pairs.txt"
    inp.write_text("\n".join("%s %s" % p for p in pairs) + "\n\n")
    out = tmp_path / "out.txt"
    C.pairwise_align_sequences(str(inp), str(out), aligner=aligner)
    lines = out.read_text().splitlines()
    assert lines[0].split("\t") == ["query", "target", "score", "query_start", "target_start", "cigar", "mismatches", "gap_bases",
                                    "padded_query", "alignment", "padded_target"]
    for (q, t), ln in zip(pairs, lines[1:]):
        f = ln.split("\t")
        w = O.align_best(q, t.upper())
        assert f == [q, t.upper(), str(w["score"]), "1", str(w["start"]), w["cigar"], str(w["mismatches"]), str(w["gap_bases"]),
                     w["padded_guide"], w["padded_alignment"], w["padded_target"]]


def _a2r_inputs(tmp_path, seed):
    """A small FASTA (mixed case, an N block) with planted sites and a task file around them."""
    from fasta_util import write_fasta
    rng = np.random.default_rng(seed)
    guides = ["CTTGCCCCACAGGGCAGTAAnrg", "tttvAACCAACCAACCGGTTACGT", "GTGACTTGAAGTCTCAGTATA", "GATACGTCTCGTACTGTnrg"]
    contigs = []
    tasks = []
    for ci in range(3):
        n = int(rng.integers(1500, 4000))
        seq = np.array(list("ACGT"))[rng.integers(0, 4, n)]
        lower = rng.random(n) < 0.2
        seq = np.where(lower, np.char.lower(seq), seq)
        seq[600:640] = "N"
        s = "".join(seq)
        for k, g in enumerate(guides):
            site = g.upper().replace("N", "A").replace("R", "G").replace("V", "C")
            if k % 2:
                site = rc(site)
            pos = int(rng.integers(50, n - 100))
            site = list(site)
            for _ in range(int(rng.integers(0, 3))):
                site[int(rng.integers(0, len(site)))] = "ACGT"[int(rng.integers(0, 4))]
            s = s[:pos] + "".join(site) + s[pos + len(site):]
            tasks.append(("t%d_%d" % (ci, k), g, "c%d" % ci, pos + int(rng.integers(0, 25))))
        tasks.append(("edge%d" % ci, guides[0], "c%d" % ci, 5))            # window clipped at the contig start
        tasks.append(("", guides[2], "c%d" % ci, n - 3))                   # no id -> the query is the id; clipped at the end
        contigs.append(("c%d" % ci, s))
    fa = write_fasta(str(tmp_path / ("a2r%d.fa" % seed)), contigs)
    inp = tmp_path / ("a2r%d.tsv" % seed)
    inp.write_text("id\tquery\tchrom\tposition\n" + "".join("%s\t%s\t%s\t%d\n" % t for t in tasks))
    return fa, str(inp)


@pytest.mark.parametrize("mode", ["best", "limits", "limits-window"])
def test_align_to_reference_tool(C, tmp_path, mode):
    """AlignToReference (AlignToReference.scala:95-146) file to file: every ReferenceHit column against the oracle's restatement
    (no reference test pins the tool itself; its aligner calls are pinned by K23-K26)."""
    fa, inp = _a2r_inputs(tmp_path, 11)
    kw, okw = {}, {}
    if mode != "best":
        kw = dict(max_guide_diffs=4, max_pam_mismatches=1, max_overlap=5, max_gaps_between_guide_and_pam=2)
        okw = dict(limits=(4, 1, 5), g=2)
    if mode == "limits-window":
        kw["window_size"] = 120; okw["window_size"] = 120
        kw["max_total_diffs"] = 5; okw["D"] = 5
    out = tmp_path / "a2r_out.txt"
    text = C.align_to_reference(inp, fa, str(out), version="unknown", time_stamp="n/a", **kw)
    assert out.read_text() == text
    header, want = O.align_to_reference(fa, inp, **okw)
    lines = text.splitlines()
    assert lines[0].split("\t") == header
    got = [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]
    assert len(want) > 10
    assert got == want


def test_module_cli_align_to_reference(C, tmp_path):
    """python -m calitas_amd AlignToReference with the reference's flags (AlignToReference.scala:34-51)."""
    from calitas_amd.__main__ import main
    fa, inp = _a2r_inputs(tmp_path, 12)
    out = tmp_path / "cli_out.txt"
    assert main(["AlignToReference", "-i", inp, "-r", fa, "-o", str(out), "-d", "3", "-p", "1", "-O", "8", "-g", "1"]) == 0
    header, want = O.align_to_reference(fa, inp, limits=(3, 1, 8), g=1)
    lines = out.read_text().splitlines()
    got = [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]
    skip = {"aligner_version", "time_stamp"}
    assert [{k: v for k, v in r.items() if k not in skip} for r in got] == [{k: v for k, v in r.items() if k not in skip} for r in want]
    with pytest.raises(ValueError):
        main(["AlignToReference", "-i", inp, "-r", fa, "-o", str(out), "-d", "3"])   # all or none of -d/-p/-O (A2R:81-85)
