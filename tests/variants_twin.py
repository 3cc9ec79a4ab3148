"""TEST INFRASTRUCTURE: the variant branch of SearchReference (SearchReference.scala:101-400, 570-630) written a second time, in Python,
as a cross-check of the product's implementation (calitas_amd/csrc/variants.cpp behind calitas_search_variants).  Window production
(nextChunk / reChunk / alleleCombos / buildVariantWindow), lift-back and the variant columns follow the Scala; the windows are aligned
on the GPU through calitas_align_windows like the product does.  Only tests/ and tools/fuzz_variants.py import this module; it used to
live in the package (calitas_amd/variants.py) and was moved out so that the package holds one implementation."""
import ctypes
import hashlib
import os

from calitas_amd import _lib
from calitas_amd._lib import lib
from calitas_amd.aligner import Guide, make_params, read_hits
from calitas_amd.tools import SequentialGuideAligner
from calitas_amd.variants import Variant, format_metric_double, read_vcf, vcf_identifier


class VariantAllele:                      # SearchReference.scala:105-110
    __slots__ = ("id", "pos", "ref", "alt", "af")

    def __init__(self, vid, pos, ref, alt, af):
        self.id, self.pos, self.ref, self.alt, self.af = vid, pos, ref, alt, af

    def display_string(self):
        import struct
        af32 = struct.unpack("f", struct.pack("f", self.af))[0]     # the reference holds AF as a Float
        return "%s:%d:%s>%s:%.3f" % (self.id or ".", self.pos - 1, self.ref, self.alt, af32)


class VariantWindow:                      # SearchReference.scala:118-157
    def __init__(self, chrom, start, variants, cigar, bases):
        self.chrom, self.start, self.variants, self.cigar, self.bases = chrom, start, variants, cigar, bases

    def ref_offset_at_base_offset(self, offset, preceding):
        on_q = lambda e: e[1] if e[0] in "MI" else 0
        on_t = lambda e: e[1] if e[0] in "MD" else 0
        if offset == len(self.bases):
            return self.start - 1 + sum(on_t(e) for e in self.cigar)
        ref_off, base_off, k = self.start - 1, 0, 0
        while offset >= base_off + on_q(self.cigar[k]):
            ref_off += on_t(self.cigar[k])
            base_off += on_q(self.cigar[k])
            k += 1
        op = self.cigar[k][0]
        if op == "I":
            return ref_off - 1 if preceding else ref_off
        if op == "M":
            return ref_off + (offset - base_off)
        raise RuntimeError("Query bases can't be present at operator %s." % op)

    @property
    def cigar_string(self):
        return "".join("%d%s" % (n, op) for op, n in self.cigar)


def allele_combos_counts(counts):         # SearchReference.scala:377-399
    total = 1
    for c in counts:
        total *= c
    results = [[0] * len(counts) for _ in range(total)]
    denom = 1
    for i, n in enumerate(counts):
        denom *= n
        group = total // denom
        j, allele = 0, 0
        while j < total:
            for _ in range(group):
                results[j][i] = allele
                j += 1
            allele = (allele + 1) % n
    return results


def _is_valid(variants):                  # VariantSet.isValid SearchReference.scala:182-193
    if len(variants) == 1:
        return True
    for a, b in zip(variants, variants[1:]):
        s1, e1, s2, e2 = a.pos, a.pos + len(a.ref) - 1, b.pos, b.pos + len(b.ref) - 1
        if a.chrom == b.chrom and s1 <= e2 and e1 >= s2:
            return False
    return True


def allele_combos(vs, max_variants):      # SearchReference.scala:351-369 -> list of (variants, alleles)
    if len(vs) > max_variants:
        v = vs[0]
        return [([v], [a + 1]) for a in range(len(v.alts))]
    out = []
    for alleles in allele_combos_counts([1 + len(v.alts) for v in vs]):
        sel = [(v, a) for v, a in zip(vs, alleles) if a != 0]
        if sel and _is_valid([v for v, _ in sel]):
            out.append(([v for v, _ in sel], [a for _, a in sel]))
    return out


def build_variant_window(variants, alleles, chrom, ref_upper, padding):     # SearchReference.scala:263-323
    window_start = max(1, variants[0].pos - padding)
    window_end = min(len(ref_upper), variants[-1].end + padding)
    bases = bytearray(ref_upper[window_start - 1:window_end])
    vas = [VariantAllele(v.id, v.pos, v.ref, v.alts[a - 1], v.afs[a - 1] if a - 1 < len(v.afs) else 0.0) for v, a in zip(variants, alleles)]
    for al in reversed(vas):
        i = al.pos - window_start
        bases[i:i + len(al.ref)] = al.alt.encode()
    cigar, ref_pos, base_off = [], window_start, 0
    for al in vas:
        pm = al.pos - ref_pos
        if pm > 0:
            cigar.append(["M", pm]); ref_pos += pm; base_off += pm
        rl, alen = len(al.ref), len(al.alt)
        if rl == alen:
            cigar.append(["M", rl])
        elif rl == 1 and alen > 1:
            cigar += [["M", 1], ["I", alen - 1]]
        elif rl > 1 and alen == 1:
            cigar += [["M", 1], ["D", rl - 1]]
        else:
            cigar += [["D", rl], ["I", alen]]
        ref_pos += rl; base_off += alen
    cigar.append(["M", len(bases) - base_off])
    merged = []
    for op, n in cigar:                                                      # Cigar.coalesce
        if merged and merged[-1][0] == op:
            merged[-1][1] += n
        else:
            merged.append([op, n])
    if sum(n for op, n in merged if op in "MI") != len(bases):
        raise RuntimeError("requirement failed: cigar length on query != bases")
    return VariantWindow(chrom, window_start, vas, [tuple(e) for e in merged], bytes(bases))


def variant_windows(contig_names, get_upper, vcf_variants, chrom, padding, max_variants):
    """variantWindowIterator (SearchReference.scala:217-256) with nextChunk / reChunk (326-347)."""
    vs = [v for v in vcf_variants if chrom is None or v.chrom == chrom]
    order = [n for n in contig_names if chrom is None or n == chrom]
    ci, i = 0, 0
    while i < len(vs):
        chunk, last = [vs[i]], vs[i]
        i += 1
        while i < len(vs) and vs[i].chrom == last.chrom and vs[i].pos <= last.end + padding:
            last = vs[i]; chunk.append(last); i += 1
        chunks = []
        for s in range(len(chunk)):
            sub = []
            for k in range(s, len(chunk)):
                if chunk[k].pos - chunk[s].end > padding:
                    break
                sub.append(chunk[k])
            chunks.append(sub)
        while ci < len(order) and order[ci] != chunk[0].chrom:
            ci += 1
        if ci >= len(order):
            raise RuntimeError("next on empty iterator (VCF contig %s not in reference order)" % chunk[0].chrom)
        ref_upper = get_upper(order[ci])
        for c in chunks:
            for variants, alleles in allele_combos(c, max_variants):
                yield build_variant_window(variants, alleles, order[ci], ref_upper, padding)


_COMP = bytes.maketrans(b"ACGTUMKRYVBHDWSNacgtumkryvbhdwsn", b"TGCAAKMYRBVDHWSNtgcaakmyrbvdhwsn")


def _revcomp(b):
    return b.translate(_COMP)[::-1]


def search_reference_with_variants(sr, ctx, vcf_path, version=None, time_stamp=None):
    """SearchReference.execute with --variants on a resident reference.  `sr` is the SearchReference mirror object."""
    import struct
    kw = sr._kw
    query = sr.query
    d, p_, g = kw["max_guide_diffs"], kw["max_pam_mismatches"], kw["max_gaps_between_guide_and_pam"]
    D = kw["max_total_diffs"] if kw["max_total_diffs"] is not None else d + g + p_
    O = kw["max_overlap"]
    chrom_index = ctx.contig_names.index(sr.chrom) if sr.chrom is not None else -1
    params = make_params(chrom_index=chrom_index, **kw)

    # reference windows on the GPU
    out, n = ctx.search_raw([query], params)
    try:
        # variant windows: built here, aligned on the GPU
        _, vcf = read_vcf(vcf_path)
        md5 = hashlib.md5(open(vcf_path, "rb").read()).hexdigest()
        vcf_id = "%s:%s" % (os.path.basename(str(vcf_path)), md5)                          # ReferenceHit.scala:175-183
        padding = query.length - 1 + d + g                                                 # SearchReference.scala:575
        upper_cache = {}

        def get_upper(name):
            if name not in upper_cache:
                upper_cache.clear()
                ci = ctx.contig_names.index(name)
                upper_cache[name] = ctx.fetch_bases(ci, 0, ctx.contig_lengths[ci]).encode()
            return upper_cache[name]

        windows = list(variant_windows(ctx.contig_names, get_upper, vcf, sr.chrom, padding, kw["max_variants"]))
        aligner = SequentialGuideAligner(context=ctx, mismatch_net_cost=kw["guide_mismatch_net_cost"], genome_gap_net_cost=kw["genome_gap_net_cost"],
                                         guide_gap_net_cost=kw["guide_gap_net_cost"], pam_mismatch_net_cost=kw["pam_mismatch_net_cost"],
                                         eqx_by_score=kw["eqx_by_score"])
        genome_build = ctx.genome_build()
        search_pam = ",".join(query.pams)
        args = core_parameters(kw, D)
        ext_rows = []
        B = 4096
        for b0 in range(0, len(windows), B):
            ws = windows[b0:b0 + B]
            res = aligner.align_many([query] * len(ws), [w.bases for w in ws], names=[w.chrom for w in ws], max_guide_diffs=d,
                                     max_gaps_between_guide_and_pam=g, max_pam_diffs=p_, max_total_diffs=D, max_overlap=O)
            for w, alns in zip(ws, res):
                wl = len(w.bases)
                ci = ctx.contig_names.index(w.chrom)
                clen = ctx.contig_lengths[ci]
                for a in alns:
                    pos_strand = a.strand == "+"
                    l10 = w.bases[a.guide_start_offset - 10:a.guide_start_offset] if a.guide_start_offset >= 10 else None
                    r10 = w.bases[a.guide_end_offset:a.guide_end_offset + 10] if wl - a.guide_end_offset >= 10 else None
                    l8 = w.bases[a.start_offset - 8:a.start_offset] if a.start_offset >= 8 else None
                    r8 = w.bases[a.end_offset:a.end_offset + 8] if wl - a.end_offset >= 8 else None
                    if not pos_strand:                                                      # SearchReference.scala:605-611
                        l10, r10, l8, r8 = (_revcomp(r10) if r10 is not None else None, _revcomp(l10) if l10 is not None else None,
                                            _revcomp(r8) if r8 is not None else None, _revcomp(l8) if l8 is not None else None)
                    start = w.ref_offset_at_base_offset(a.start_offset, True)               # SearchReference.scala:615-620
                    end = w.ref_offset_at_base_offset(a.end_offset, False)
                    gstart = w.ref_offset_at_base_offset(a.guide_start_offset, True)
                    gend = w.ref_offset_at_base_offset(a.guide_end_offset, False)

                    def fetch(s1, e1):                                                      # fetchBases ReferenceHit.scala:261-266
                        as_, ae = max(1, s1), min(clen, e1)
                        mid = ctx.fetch_bases(ci, as_ - 1, ae - as_ + 1) if ae >= as_ else ""
                        bases = "N" * (as_ - s1) + mid + "N" * max(0, e1 - ae)
                        return (_revcomp(bases.encode()).decode() if not pos_strand else bases).upper()
                    ten_left = lambda: fetch(gstart + 1 - 10, gstart)
                    ten_right = lambda: fetch(gend + 1, gend + 10)
                    eight_left = lambda: fetch(start + 1 - 8, start)
                    eight_right = lambda: fetch(end + 1, end + 8)
                    vs = [v for v in w.variants if start <= v.pos - 1 <= end]               # ReferenceHit.scala:211
                    pg, pt = a.padded_guide, a.padded_target
                    ups = [i for i, ch in enumerate(pg) if ch.isupper()]
                    unpadded_target = "".join(ch for ch in pt[ups[0]:ups[-1] + 1] if ch.isalpha())
                    tlen = sum(1 for ch in a.ops if ch in "=XD")
                    af = None
                    if vs:
                        mn = min(vs, key=lambda v: struct.unpack("f", struct.pack("f", v.af))[0])
                        af = format_metric_double(struct.unpack("f", struct.pack("f", mn.af))[0])
                    row = [sr.guide_id, query.guide, genome_build + ("+variants" if vs else ""), w.chrom, str(gstart), str(gend), a.strand,
                           unpadded_target,
                           (l10.decode() if l10 is not None else (ten_left() if pos_strand else ten_right())),
                           (r10.decode() if r10 is not None else (ten_right() if pos_strand else ten_left())),
                           "".join(ch for ch in a.guide if ch.islower()),
                           ";".join(v.id for v in vs) if vs else "", ";".join(v.display_string() for v in vs) if vs else "",
                           vcf_id if vs else "", af if vs else "",
                           str(a.score), str(a.guide_mismatches), str(a.guide_gap_bases), str(a.guide_mismatches + a.guide_gap_bases),
                           str(a.pam_mismatches), str(a.edits), pg, a.padded_alignment, pt,
                           (l8.decode() if l8 is not None else (eight_left() if pos_strand else eight_right())),
                           (r8.decode() if r8 is not None else (eight_right() if pos_strand else eight_left())),
                           a.cigar, str(len(query.guide)), str(len(unpadded_target)), "CALITAS:SearchReference", version or "",
                           search_pam, args, time_stamp or ""]
                    ext_rows.append((ci, gstart, gstart + tlen - 1, a.score, a.strand, row[12], "\t".join(row)))
        # merge: removeOverlaps + sort over everything (SearchReference.scala:641-648)
        n_ext = len(ext_rows)
        ExtArr = _lib.ExtHitT * max(1, n_ext)
        ext = ExtArr()
        keep = []
        for i, (ci, gs, en, sc, st, desc, row) in enumerate(ext_rows):
            ext[i].contig_index, ext[i].coordinate_start, ext[i].end, ext[i].score, ext[i].strand = ci, gs, en, sc, ord(st)
            kd, kr = desc.encode(), row.encode()
            keep += [kd, kr]
            ext[i].variant_description, ext[i].row = kd, kr
        gq = query.to_c()
        tsv, rows = ctypes.c_void_p(), ctypes.c_uint64()
        _lib.check(ctx._h, lib.calitas_hits_tsv_ext(ctx._h, ctypes.byref(gq), sr.guide_id.encode(), ctypes.byref(params), out, n, ext, n_ext,
                                                    version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                    ctypes.byref(tsv), ctypes.byref(rows)))
        text = ctypes.string_at(tsv).decode()
        lib.calitas_free(tsv)
        return text, rows.value
    finally:
        lib.calitas_free(out)


def core_parameters(kw, max_total):       # SearchReference.scala:496-508
    d = {"max-variants": kw["max_variants"], "window-size": kw["window_size"], "max-guide-diffs": kw["max_guide_diffs"],
         "max-pam-mismatches": kw["max_pam_mismatches"], "max-gaps-between-guide-and-pam": kw["max_gaps_between_guide_and_pam"],
         "max-total-diffs": max_total, "max-overlap": kw["max_overlap"], "guide-mismatch-net-cost": kw["guide_mismatch_net_cost"],
         "pam-mismatch-net-cost": kw["pam_mismatch_net_cost"], "genome-gap-net-cost": kw["genome_gap_net_cost"],
         "guide-gap-net-cost": kw["guide_gap_net_cost"]}
    return ";".join(sorted("%s=%s" % kv for kv in d.items()))
