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
        cache = {}                                          # the variant branch passes one Guide object thousands of times
        keep = [cache[id(g)] if id(g) in cache else cache.setdefault(id(g), g.to_c()) for g in guides]
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


def read_fasta(path):
    """Contig name -> bases (bytes, case preserved), in file order."""
    ref, name, parts = {}, None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if not line:
                continue
            if line.startswith(b">"):
                if name is not None:
                    ref[name] = b"".join(parts)
                name, parts = line[1:].split()[0].decode(), []
            elif name is not None:
                parts.append(line)
    if name is not None:
        ref[name] = b"".join(parts)
    return ref


_COMP = bytes.maketrans(b"ACGTUMKRYVBHDWSNacgtumkryvbhdwsn", b"TGCAAKMYRBVDHWSNtgcaakmyrbvdhwsn")

HIT_COLUMNS = ["guide_id", "unpadded_guide_sequence", "genome_build", "chromosome", "coordinate_start", "coordinate_end", "strand",
               "unpadded_target_sequence", "ten_bases_5_prime", "ten_bases_3_prime", "pam_used", "variant_id", "variant_description",
               "variant_vcf", "allele_frequency", "score", "guide_mm", "guide_gaps", "guide_mm_plus_gaps", "pam_mm", "total_mm_plus_gaps",
               "padded_guide", "padded_alignment", "padded_target", "padded_extra_8_bases_5_prime", "padded_extra_8_bases_3_prime", "cigar",
               "unpadded_guide_sequence_length", "unpadded_target_sequence_length", "aligner", "aligner_version", "aligner_search_pam",
               "aligner_other_parameters", "time_stamp"]


def reference_hit_row(aln, guide, guide_id, contig, genome_build, aligner_id, version, arguments, time_stamp):
    """ReferenceHit.Builder.build without variants (ReferenceHit.scala:210-254) for a tools.GuideAlignment on `contig` (bytes)."""
    neg = aln.strand == "-"

    def fetch(start, end):                                      # fetchBases ReferenceHit.scala:261-266, 1-based closed
        a_s, a_e = max(1, start), min(len(contig), end)
        bases = b"N" * (a_s - start) + contig[a_s - 1:a_e] + b"N" * (end - a_e)
        if neg:
            bases = bases.translate(_COMP)[::-1]
        return bases.decode().upper()

    ten_left, ten_right = fetch(aln.guide_start_offset + 1 - 10, aln.guide_start_offset), fetch(aln.guide_end_offset + 1, aln.guide_end_offset + 10)
    eight_left, eight_right = fetch(aln.start_offset + 1 - 8, aln.start_offset), fetch(aln.end_offset + 1, aln.end_offset + 8)
    pg, pt = aln.padded_guide, aln.padded_target
    ups = [i for i, ch in enumerate(pg) if ch.isupper()]        # unpaddedTargetWithoutPam GuideAlignment.scala:111-115
    unpadded_target = "".join(ch for ch in pt[ups[0]:ups[-1] + 1] if ch != "-")
    row = [guide_id, guide.guide, genome_build, aln.chrom, aln.guide_start_offset, aln.guide_end_offset, aln.strand, unpadded_target,
           ten_right if neg else ten_left, ten_left if neg else ten_right, "".join(ch for ch in aln.guide if ch.islower()), "", "", "", "",
           aln.score, aln.guide_mismatches, aln.guide_gap_bases, aln.guide_mismatches + aln.guide_gap_bases, aln.pam_mismatches, aln.edits,
           pg, aln.padded_alignment, pt, eight_right if neg else eight_left, eight_left if neg else eight_right, aln.cigar, len(guide.guide),
           len(unpadded_target), aligner_id, version, ",".join(guide.pams), arguments, time_stamp]
    return "\t".join(str(x) for x in row)


def align_to_reference(input_path, ref, output_path=None, window_size=None, max_guide_diffs=None, max_pam_mismatches=None,
                       max_gaps_between_guide_and_pam=Defaults.MaxGapsBetweenGuideAndPam, max_total_diffs=None, max_overlap=None,
                       guide_mismatch_net_cost=Defaults.MismatchNetCost, pam_mismatch_net_cost=Defaults.PamMismatchNetCost,
                       genome_gap_net_cost=Defaults.GenomeGapNetCost, guide_gap_net_cost=Defaults.GuideGapNetCost, threads=8, context=None,
                       device=0, version=None, time_stamp=None, eqx_by_score=0):
    """AlignToReference.execute (AlignToReference.scala:95-146): a tab-delimited file with a header (id optional, query, chrom,
    position) -> ReferenceHit rows.  All of -d/-p/-O given: every alignment meeting them (alignToRef); none given: the single best
    alignment per query (alignToRefBest).  Batches of 10000 tasks go to the GPU in one calitas_align_windows call each and are
    sorted on their own (AlignToReference.scala:104,140).  `threads` is accepted and ignored.  Returns the TSV text."""
    import time as _time
    given = [x is not None for x in (max_guide_diffs, max_pam_mismatches, max_overlap)]
    if any(given) and not all(given):
        raise ValueError("Must specify all or none of: --max-guide-diffs, --max-pam-mismatches, --max-overlap")
    limits = all(given)
    contigs = read_fasta(ref)
    ctx = context or Context(device)
    own = context is None
    try:
        if not ctx.contig_names:
            ctx.set_reference_fasta(ref)
        genome_build = ctx.genome_build()
        order = {n: i for i, n in enumerate(ctx.contig_names)}
        aligner = SequentialGuideAligner(context=ctx, ref=contigs, mismatch_net_cost=guide_mismatch_net_cost,
                                         genome_gap_net_cost=genome_gap_net_cost, guide_gap_net_cost=guide_gap_net_cost,
                                         pam_mismatch_net_cost=pam_mismatch_net_cost, eqx_by_score=eqx_by_score)
        opt = lambda v: "Some(%d)" % v if v is not None else "None"          # Scala prints the Option-typed flags this way
        arguments = ";".join(sorted("%s=%s" % kv for kv in {
            "max-guide-diffs": opt(max_guide_diffs), "max-pam-mismatches": opt(max_pam_mismatches),
            "max-gaps-between-guide-and-pam": max_gaps_between_guide_and_pam, "max-overlap": opt(max_overlap),
            "guide-mismatch-net-cost": guide_mismatch_net_cost, "pam-mismatch-net-cost": pam_mismatch_net_cost,
            "genome-gap-net-cost": genome_gap_net_cost, "guide-gap-net-cost": guide_gap_net_cost}.items()))   # AlignToReference.scala:70-79
        version = version or _time.strftime("unknown-%Y-%m-%d", _time.gmtime())
        time_stamp = time_stamp or _time.strftime("%a %b %d %H:%M:%S UTC %Y", _time.gmtime())

        tasks = []
        with open(input_path) as f:
            header = f.readline().rstrip("\r\n").split("\t")
            for need in ("query", "chrom", "position"):
                if need not in header:
                    raise ValueError("input needs the columns query, chrom and position")
            for line in f:
                line = line.rstrip("\r\n")
                if not line:
                    continue
                row = dict(zip(header, line.split("\t")))
                tasks.append((row.get("id") or row["query"], row["query"], row["chrom"], int(row["position"])))

        lines = ["\t".join(HIT_COLUMNS)]
        for b0 in range(0, len(tasks), 10000):
            batch = tasks[b0:b0 + 10000]
            guides = [Guide(q) for _, q, _, _ in batch]
            regions = [aligner._region(g, chrom, pos, window_size) for g, (_, _, chrom, pos) in zip(guides, batch)]
            D = None
            if limits:
                D = max_total_diffs if max_total_diffs is not None else max_guide_diffs + max_gaps_between_guide_and_pam + max_pam_mismatches
            res = aligner.align_many(guides, [t for t, _ in regions], [o for _, o in regions], [c for _, _, c, _ in batch],
                                     max_guide_diffs if limits else None, max_gaps_between_guide_and_pam,
                                     max_pam_mismatches if limits else None, D, max_overlap if limits else 0)
            hits = []
            for (tid, _, chrom, _), g, alns in zip(batch, guides, res):
                alns = sorted(alns, key=lambda a: (-a.score, a.gap_bases))    # alignToRef's order, SequentialGuideAligner.scala:386
                if not limits:
                    if not alns:
                        raise ValueError("head of empty list")
                    alns = alns[:1]
                for a in alns:
                    key = (order[chrom], a.guide_start_offset, a.strand, -a.score)               # ReferenceHit.scala:284
                    hits.append((key, reference_hit_row(a, g, tid, contigs[chrom], genome_build, "CALITAS:AlignToReference", version,
                                                        arguments, time_stamp)))
            hits.sort(key=lambda h: h[0])                                     # stable
            lines += [r for _, r in hits]
        text = "\n".join(lines) + "\n"
        if output_path is not None:
            with open(output_path, "w") as f:
                f.write(text)
        return text
    finally:
        if own:
            ctx.close()
