"""Generates tests/golden/u_probe.json: the shortest inputs found on which the readings of fgbio's Glocal enumeration that the
reference's own known-answer tests cannot tell apart (U1, U2) give different results.  Whoever has fgbio 2.0.0 at hand
runs the Scala snippet of INTEGRATION.md on these inputs once and knows which switch the product has to run with.

The call probed is com.fulcrumgenomics.alignment.Aligner(scorer, useEqualsAndX = true, Mode.Glocal).align(query, target, minScore) --
SequentialGuideAligner.scala:210, 261, 278, 295, 299 -- with the default scorer (SGA:192-208: match 60, mismatch -60, query gap -121,
target gap -62).  Readings (oracle/calitas_oracle.cpp glocal_align):
  U1-a  one alignment per end column of the bottom row, the best of the three matrices      (oracle switches 0; product default)
  U1-b  one alignment per (end column, matrix) whose bottom-row score reaches minScore        (oracle switch 1; product eqx_by_score bit 1)
  U1-c  end cells best score first; an alignment whose traceback meets a cell an earlier one used is dropped   (oracle switch 4; probe only)
  U2-a  '=' where Sequences.compatible(query base, target base)                                (oracle default)
  U2-b  '=' where the pairing scored as a match                                                 (oracle switch 2; product eqx_by_score bit 0)
Data only; run from the repository root:  python tests/golden/make_u_probe.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

MATCH, WORST = 60, -122          # SGA:192-208 with the default net costs; minGuideScore = match * L + worst * d (SGA:239-243)


def three_way(query, target, min_score):
    a = O.glocal(query, target, min_score, switches=0)
    b = O.glocal(query, target, min_score, switches=1)
    c = O.glocal(query, target, min_score, switches=4)
    return a, b, c


def align_rows(guide, target, d, overlap, sw):
    """SequentialGuideAligner.align for a PAM-less guide, as (strand, start, end, score, cigar) rows."""
    return [{k: r[k] for k in ("strand", "start", "end", "score", "cigar")} for r in O.align(guide, target, d, 0, 0, d, O=overlap, switches=sw)]


def search_u1(seed=7, tries=400000):
    """Random queries of 8-12 bases against targets made of overlapping near-copies of them, at most two guide differences; the smallest
    (len(query) + len(target)) on which the three U1 readings return three different lists from Aligner.align AND three different
    results from SequentialGuideAligner.align for some --max-overlap."""
    rng = np.random.default_rng(seed)
    best = None
    for _ in range(tries):
        L = int(rng.integers(8, 13))
        q = "".join(rng.choice(list("ACGT"), size=L))
        # a target with two or three copies of the query, edited and overlapping
        parts = []
        for _k in range(int(rng.integers(1, 4))):
            s = list(q)
            for _e in range(int(rng.integers(0, 3))):
                p = int(rng.integers(0, len(s)))
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    s[p] = "ACGT"[int(rng.integers(0, 4))]
                elif kind == 1 and len(s) > 2:
                    del s[p]
                else:
                    s.insert(p, "ACGT"[int(rng.integers(0, 4))])
            cut = int(rng.integers(0, 3))
            parts.append("".join(s)[cut:])
        t = "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 3)))) + "".join(parts) + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 3))))
        if best is not None and len(q) + len(t) >= best[0]:
            continue
        if len(t) < L:
            continue
        for d in (1, 2):
            min_score = MATCH * L + WORST * d
            a, b, c = three_way(q, t, min_score)
            if not (a != b and a != c and b != c):
                continue
            for overlap in (0, 2, 5, 10):
                ra, rb, rc = (align_rows(q, t, d, overlap, sw) for sw in (0, 1, 4))
                if ra != rb and ra != rc and rb != rc:
                    best = (len(q) + len(t), q, t, d, min_score, overlap)
                    break
            if best is not None and best[1] == q and best[2] == t:
                break
    return best


def shrink(q, t, d, overlap):
    """Greedy: drop target bases from either end while the three readings still differ pairwise at both levels."""
    min_score = MATCH * len(q) + WORST * d
    changed = True
    while changed:
        changed = False
        for cand in (t[1:], t[:-1]):
            if len(cand) < len(q):
                continue
            a, b, c = three_way(q, cand, min_score)
            ra, rb, rc = (align_rows(q, cand, d, overlap, sw) for sw in (0, 1, 4))
            if a != b and a != c and b != c and ra != rb and ra != rc and rb != rc:
                t, changed = cand, True
                break
    return t


def main():
    found = search_u1()
    assert found is not None, "no input separates the three readings"
    _, q, t, d, min_score, overlap = found
    t = shrink(q, t, d, overlap)
    a, b, c = three_way(q, t, min_score)
    assert a != b and a != c and b != c
    u1 = {"query": q, "target": t, "min_score": min_score, "max_guide_diffs": d,
          "expect": {"U1-a (one per end column, best of three matrices)": a, "U1-b (one per end column and matrix)": b,
                     "U1-c (best first, no shared traceback cells)": c}}
    # U2: a target N inside a match run -- compatible() holds, the pairing scores as a mismatch (SGA:144)
    q2, t2, d2 = "AACCAACC", "TTAACNAACCGG", 1
    ms2 = MATCH * len(q2) + WORST * d2
    ua, ub = O.glocal(q2, t2, ms2, switches=0), O.glocal(q2, t2, ms2, switches=2)
    assert ua != ub
    u2 = {"query": q2, "target": t2, "min_score": ms2, "max_guide_diffs": d2,
          "expect": {"U2-a ('=' by Sequences.compatible)": ua, "U2-b ('=' by pairing score > 0)": ub}}
    # the same two inputs through SequentialGuideAligner.align (PAM-less guides, -O 0): what the switch changes in hits.txt terms
    call = "align(guide = query, target, maxGuideDiffs = d, maxGapsBetweenGuideAndPam = 0, maxPamDiffs = 0, maxTotalDiffs = d, maxOverlap = %d)"
    u1["align"] = {"call": call % overlap, "max_overlap": overlap, "U1-a": align_rows(q, t, d, overlap, 0), "U1-b": align_rows(q, t, d, overlap, 1),
                   "U1-c": align_rows(q, t, d, overlap, 4)}
    u2["align"] = {"call": call % 0, "max_overlap": 0, "U2-a": align_rows(q2, t2, d2, 0, 0), "U2-b": align_rows(q2, t2, d2, 0, 2)}
    out = {"format": "targetStart-targetEnd:score:cigar per alignment, in the order Aligner.align returns them (1-based, inclusive)",
           "scorer": {"match": 60, "mismatch": -60, "query_gap": -121, "target_gap": -62}, "U1": u1, "U2": u2}
    with open(os.path.join(HERE, "u_probe.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
