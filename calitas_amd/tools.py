"""Mirrors of SequentialGuideAligner's per-target entry points and of the two small tools built on them, on top of
calitas_align_windows (GPU).  Reference: SequentialGuideAligner.scala:228-418, PairwiseAlignSequences.scala:24-87,
AlignToReference.scala:34-148.  Padded strings keep the case of the target bytes, as the reference does."""
import ctypes

from . import _lib
from ._lib import AlnT, GuideT, lib
from .aligner import Alignment, Context, Defaults, Guide, make_params


class GuideAlignment:
    """GuideAlignment (GuideAlignment.scala:72-183) with the padded strings materialised."""

    def __init__(self, rec, guide, target, target_offset, chrom="n/a"):
        self.chrom = chrom
        self.start_offset, self.end_offset = rec.start_offset, rec.end_offset
        self.guide_start_offset, self.guide_end_offset = rec.guide_start_offset, rec.guide_end_offset
        self.strand, self.score, self.cigar, self.ops = rec.strand, rec.score, rec.cigar, rec.ops
        pam = guide.pams[rec.pam_index] if rec.pam_index >= 0 and guide.pams else ""
        self.guide = (pam + guide.guide) if guide.pam_is_five_prime else (guide.guide + pam)  # GuideAlignment.guide
        g, a = guide.to_c(), rec.to_c()
        bufs = [ctypes.create_string_buffer(_lib.MAX_OPS + 1) for _ in range(3)]
        rc = lib.calitas_padded_strings_target(ctypes.byref(g), ctypes.byref(a), target, len(target), target_offset, *bufs)
        if rc != _lib.OK:
            raise _lib.CalitasError(rc, "calitas_padded_strings_target")
        self.padded_guide, self.padded_alignment, self.padded_target = (b.value.decode() for b in bufs)

    # GuideAlignment.scala:99-108
    @property
    def mismatches(self):
        return self.padded_alignment.count(".")

    @property
    def gap_bases(self):
        return self.padded_alignment.count("~")

    @property
    def edits(self):
        return self.mismatches + self.gap_bases

    def _count(self, lower, both_sides, mms, gaps):  # GuideAlignment.scala:139-163
        n, pg, pa = 0, self.padded_guide, self.padded_alignment
        for i, ch in enumerate(pa):
            if mms and ch == "." and pg[i].islower() == lower:
                n += 1
            elif gaps and ch == "~":
                gb = pg[i]
                me = gb != "-" and gb.islower() == lower
                if not me:
                    pi = i
                    while pi > 0 and pg[pi] == "-":
                        pi -= 1
                    ni = i
                    while ni < len(pg) - 1 and pg[ni] == "-":
                        ni += 1
                    prev, nxt = pg[pi], pg[ni]
                    if both_sides:
                        me = (prev == "-" or prev.islower() == lower) and (nxt == "-" or nxt.islower() == lower)
                    else:
                        me = (prev.isalpha() and prev.islower() == lower) or (nxt.isalpha() and nxt.islower() == lower)
                if me:
                    n += 1
        return n

    guide_mismatches = property(lambda s: s._count(False, False, True, False))
    guide_gap_bases = property(lambda s: s._count(False, False, False, True))
    pam_mismatches = property(lambda s: s._count(True, True, True, False))
    pam_gap_bases = property(lambda s: s._count(True, True, False, True))
    pam_mms_plus_gaps = property(lambda s: s._count(True, True, True, True))


class SequentialGuideAligner:
    """SequentialGuideAligner(refFile, costs) (SequentialGuideAligner.scala:170-175).  `ref` maps contig name -> bases
    (bytes, case preserved) for alignToRef."""

    def __init__(self, context=None, ref=None, mismatch_net_cost=Defaults.MismatchNetCost, genome_gap_net_cost=Defaults.GenomeGapNetCost,
                 guide_gap_net_cost=Defaults.GuideGapNetCost, pam_mismatch_net_cost=Defaults.PamMismatchNetCost, device=0, eqx_by_score=0):
        self.ctx = context or Context(device)
        self.ref = ref or {}
        self._costs = dict(guide_mismatch_net_cost=mismatch_net_cost, pam_mismatch_net_cost=pam_mismatch_net_cost,
                           genome_gap_net_cost=genome_gap_net_cost, guide_gap_net_cost=guide_gap_net_cost, eqx_by_score=eqx_by_score)

    def align_many(self, guides, targets, offsets=None, names=None, max_guide_diffs=None, max_gaps_between_guide_and_pam=
                   Defaults.MaxGapsBetweenGuideAndPam, max_pam_diffs=None, max_total_diffs=None, max_overlap=0):
        """One calitas_align_windows call for a list of (guide, target) tasks. max_guide_diffs=None selects alignBest's
        limits.  Returns a list (per task) of lists of GuideAlignment in the order align() returns them."""
        n = len(guides)
        tb = [t if isinstance(t, bytes) else t.encode() for t in targets]
        offsets = list(offsets) if offsets is not None else [0] * n
        keep = [g.to_c() for g in guides]
        garr = (GuideT * max(1, n))(*keep)
        tptr = (ctypes.c_void_p * max(1, n))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in tb])
        tlen = (ctypes.c_uint32 * max(1, n))(*[len(b) for b in tb])
        toff = (ctypes.c_int32 * max(1, n))(*offsets)
        if max_guide_diffs is None:
            params = make_params(max_guide_diffs=-1, max_gaps_between_guide_and_pam=max_gaps_between_guide_and_pam, **self._costs)
        else:
            params = make_params(max_guide_diffs=max_guide_diffs, max_pam_mismatches=max_pam_diffs,
                                 max_gaps_between_guide_and_pam=max_gaps_between_guide_and_pam, max_total_diffs=max_total_diffs,
                                 max_overlap=max_overlap, **self._costs)
        out, cnt, counts = ctypes.POINTER(AlnT)(), ctypes.c_uint64(), ctypes.POINTER(ctypes.c_uint32)()
        _lib.check(self.ctx._h, lib.calitas_align_windows(self.ctx._h, n, garr, tptr, tlen, toff, ctypes.byref(params), ctypes.byref(out),
                                                          ctypes.byref(cnt), ctypes.byref(counts)))
        try:
            res, k = [], 0
            for t in range(n):
                lst = []
                for _ in range(counts[t]):
                    lst.append(GuideAlignment(Alignment(out[k]), guides[t], tb[t], offsets[t], names[t] if names else "n/a"))
                    k += 1
                res.append(lst)
            return res
        finally:
            lib.calitas_free(out)
            lib.calitas_free(counts)

    def align(self, guide, target, max_guide_diffs, max_gaps_between_guide_and_pam, max_pam_diffs, max_total_diffs, max_overlap=0,
              target_name="n/a", target_offset=0):
        """SequentialGuideAligner.align (SequentialGuideAligner.scala:228-323)."""
        return self.align_many([guide], [target], [target_offset], [target_name], max_guide_diffs, max_gaps_between_guide_and_pam,
                               max_pam_diffs, max_total_diffs, max_overlap)[0]

    def align_best(self, guide, target, max_gaps_between_guide_and_pam=Defaults.MaxGapsBetweenGuideAndPam):
        """alignBest (SequentialGuideAligner.scala:333-345): maxBy(score) keeps the first maximum."""
        alns = self.align_many([guide], [target], max_gaps_between_guide_and_pam=max_gaps_between_guide_and_pam)[0]
        if not alns:
            raise ValueError("empty.maxBy")
        return max(alns, key=lambda a: a.score)  # max() returns the first maximal element

    def _region(self, guide, chrom, pos, window_size):
        if chrom not in self.ref:
            raise ValueError("Unknown chromosome: %s" % chrom)
        contig = self.ref[chrom]
        padding = window_size // 2 if window_size else guide.length * 2            # SequentialGuideAligner.scala:372
        start, end = max(pos - padding, 1), min(pos + padding, len(contig))         # :373
        return contig[start - 1:end], start - 1

    def align_to_ref(self, guide, chrom, pos, window_size=None, max_guide_diffs=None, max_gaps_between_guide_and_pam=
                     Defaults.MaxGapsBetweenGuideAndPam, max_pam_diffs=None, max_total_diffs=None, max_overlap=0):
        """alignToRef (SequentialGuideAligner.scala:359-387): result sorted by (score desc, gap bases asc)."""
        target, off = self._region(guide, chrom, pos, window_size)
        alns = self.align_many([guide], [target], [off], [chrom], max_guide_diffs, max_gaps_between_guide_and_pam, max_pam_diffs,
                               max_total_diffs, max_overlap)[0]
        return sorted(alns, key=lambda a: (-a.score, a.gap_bases))                 # stable, GuideAlignment.scala:125-129

    def align_to_ref_best(self, guide, chrom, pos, window_size=None, max_gaps_between_guide_and_pam=Defaults.MaxGapsBetweenGuideAndPam):
        """alignToRefBest (SequentialGuideAligner.scala:402-418)."""
        return self.align_to_ref(guide, chrom, pos, window_size, None, max_gaps_between_guide_and_pam)[0]


def pairwise_align_sequences(input_path, output_path, max_gaps_between_guide_and_pam=Defaults.MaxGapsBetweenGuideAndPam, aligner=None,
                             **costs):
    """PairwiseAlignSequences.execute (PairwiseAlignSequences.scala:42-85): `query target` per line -> 11-column TSV.
    (`-O` is declared but unused by the reference, so it is not a parameter here.)"""
    aligner = aligner or SequentialGuideAligner(**costs)
    tasks = []
    with open(input_path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            fields = line.split()
            if len(fields) != 2:
                raise ValueError("Line found with %d fields: %s" % (len(fields), " ".join(fields)))
            tasks.append((fields[0], fields[1].upper()))
    cols = ["query", "target", "score", "query_start", "target_start", "cigar", "mismatches", "gap_bases", "padded_query", "alignment",
            "padded_target"]
    with open(output_path, "w") as out:
        out.write("\t".join(cols) + "\n")
        for b in range(0, len(tasks), 10000):
            batch = tasks[b:b + 10000]
            res = aligner.align_many([Guide(q) for q, _ in batch], [t for _, t in batch],
                                     max_gaps_between_guide_and_pam=max_gaps_between_guide_and_pam)
            for (q, t), alns in zip(batch, res):
                a = max(alns, key=lambda x: x.score)
                out.write("\t".join(str(x) for x in (q, t, a.score, 1, a.start_offset, a.cigar, a.mismatches, a.gap_bases, a.padded_guide,
                                                     a.padded_alignment, a.padded_target)) + "\n")
