"""The Python face of the variant branch of SearchReference (SearchReference.scala:570-630) and PrepareVcf (PrepareVcf.scala:31-92).

The branch itself -- variantWindowIterator and its helpers (SearchReference.scala:217-399), the per-window align calls, lift-back,
window flanks, variant columns and the merge with the reference hits -- is calitas_search_variants in the library
(calitas_amd/csrc/variants.cpp); search_variants() below calls it.  This module adds what sits around it: a VCF reader for the
subset the path needs (fgbio vcf.api: CHROM POS ID REF ALT FILTER INFO(AF, END)), the `name:md5` identifier of a VCF
(ReferenceHit.scala:175-183) and PrepareVcf.  (A second, pure-Python implementation of the branch used as a cross-check lives with
the tests: tests/variants_twin.py.)"""
import ctypes
import gzip
import hashlib
import os

from . import _lib
from ._lib import lib
from .aligner import make_params


class Variant:
    __slots__ = ("chrom", "pos", "id", "ref", "alts", "afs", "end", "filters", "line")

    def __init__(self, chrom, pos, vid, ref, alts, afs=None, end=None, filters=("PASS",), line=None):
        self.chrom, self.pos, self.id, self.ref, self.alts = chrom, pos, vid or "", ref, list(alts)
        self.afs = list(afs) if afs else []
        self.end = end if end is not None else pos + len(ref) - 1      # fgbio Variant.end
        self.filters, self.line = tuple(filters), line


def _open_text(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path)


def read_vcf(path, chrom=None):
    header, out = [], []
    with _open_text(path) as f:
        for line in f:
            if line.startswith("#"):
                header.append(line.rstrip("\n"))
                continue
            fld = line.rstrip("\n").split("\t")
            if len(fld) < 5 or (chrom is not None and fld[0] != chrom):
                continue
            afs, end = [], None
            if len(fld) > 7:
                for kv in fld[7].split(";"):
                    if kv.startswith("AF="):
                        afs = [float(x) for x in kv[3:].split(",") if x not in (".", "")]
                    elif kv.startswith("END="):
                        end = int(kv[4:])
            filters = tuple(fld[6].split(";")) if len(fld) > 6 and fld[6] not in (".", "") else ("PASS",)
            out.append(Variant(fld[0], int(fld[1]), "" if fld[2] == "." else fld[2], fld[3], fld[4].split(","), afs, end, filters, fld))
    return header, out


def format_metric_double(d):
    """fgbio Metric.formatValue(Double): DecimalFormat("0.######"), scientific below 1e-5 (recalled; not pinned by a reference test)."""
    if d == 0:
        return "0"
    if abs(d) < 0.00001:
        import math
        ex = math.floor(math.log10(abs(d)))
        m = ("%.5f" % (d / 10 ** ex)).rstrip("0").rstrip(".")
        return "%sE%d" % (m, ex)
    return ("%.6f" % d).rstrip("0").rstrip(".")


def vcf_identifier(vcf_path):
    """ReferenceHit.scala:175-183: file name and md5 of the VCF."""
    h = hashlib.md5()
    with open(vcf_path, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            h.update(block)
    return "%s:%s" % (os.path.basename(str(vcf_path)), h.hexdigest())


def search_variants(sr, ctx, vcf_path, chrom_index=-1, version=None, time_stamp=None):
    """SearchReference.execute with --variants through calitas_search_variants (window production, lift-back and rows in C++;
    the windows are aligned on the GPU).  Returns (tsv_text, n_rows); sr.variant_windows = number of variant windows."""
    params = make_params(chrom_index=chrom_index, **sr._kw)
    gq = sr.query.to_c()
    tsv, nbytes, rows, nwin = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(ctx._h, lib.calitas_search_variants(ctx._h, ctypes.byref(gq), sr.guide_id.encode(), ctypes.byref(params), str(vcf_path).encode(),
                                                   sr.chrom.encode() if sr.chrom is not None else None, None,   # NULL: the library computes name:md5
                                                   version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                   ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows), ctypes.byref(nwin)))
    try:
        text = ctypes.string_at(tsv, nbytes.value).decode()
    finally:
        lib.calitas_free(tsv)
    sr.variant_windows = nwin.value
    sr.timing = ctx.timing()                                   # contig_passes > 0: the reference rows were built on the device
    return text, rows.value


def read_sequence_dictionary(path):
    """The sequences of a sequence dictionary as [(name, length, assembly or None)], the way SAMSequenceDictionaryExtractor reads them
    for PrepareVcf.scala:46-49: a `.dict` / SAM header text (`@SQ` lines with SN, LN and an optional AS), a FASTA (its companion
    `.dict` -- `ref.dict` or `ref.fa.dict` -- as htsjdk looks for it) or a `.fai` index (names and lengths only)."""
    path = str(path)
    low = path.lower()
    if low.endswith((".fa", ".fasta", ".fna", ".fa.gz", ".fasta.gz")):
        base = path[:-3] if low.endswith(".gz") else path
        for cand in (os.path.splitext(base)[0] + ".dict", base + ".dict"):
            if os.path.exists(cand):
                return read_sequence_dictionary(cand)
        raise ValueError("no sequence dictionary (.dict) next to %s" % path)
    seqs = []
    with _open_text(path) as f:
        if low.endswith(".fai"):
            for line in f:
                fld = line.rstrip("\n").split("\t")
                if len(fld) >= 2:
                    seqs.append((fld[0], int(fld[1]), None))
        else:
            for line in f:
                if not line.startswith("@"):
                    break                                          # (a SAM file: the header is over)
                if not line.startswith("@SQ"):
                    continue
                tags = dict(t.split(":", 1) for t in line.rstrip("\n").split("\t")[1:] if ":" in t)
                if "SN" not in tags or "LN" not in tags:
                    raise ValueError("@SQ line without SN / LN in %s" % path)
                seqs.append((tags["SN"], int(tags["LN"]), tags.get("AS")))
    if not seqs:
        raise ValueError("no sequences in the sequence dictionary %s" % path)
    return seqs


def _header_with_dictionary(header, seqs):
    """PrepareVcf.scala:46-55: the header's contig lines become the dictionary's sequences (index order; length and, when the
    dictionary has one, assembly), every `##reference=` line goes and one `##reference=<assembly of the first sequence>` is added
    behind the other general lines.  Where the lines stand in the written header is fgbio's / htsjdk's business and pinned by no
    reference test: the contig lines take the place of the first old one (or stand in front of `#CHROM`), the reference line stands
    in front of `#CHROM`."""
    assembly = seqs[0][2]
    if assembly is None:
        raise ValueError("the first sequence of the dictionary has no assembly (AS) -- PrepareVcf --dict writes it as the VCF's `reference`")
    contigs = ["##contig=<ID=%s,length=%d%s>" % (n, ln, ",assembly=%s" % a if a is not None else "") for n, ln, a in seqs]
    out, placed = [], False
    for h in header:
        if h.startswith("##contig="):
            if not placed:
                out.extend(contigs)
                placed = True
            continue
        if h.startswith("##reference="):
            continue
        if h.startswith("#CHROM"):
            if not placed:
                out.extend(contigs)
                placed = True
            out.append("##reference=%s" % assembly)
        out.append(h)
    return out


def prepare_vcf(inputs, output, min_af=0.01, add_chr_prefix=True, dict_path=None):
    """PrepareVcf.execute (PrepareVcf.scala:39-88): PASS variants with an ALT at or above min_af, alleles below it dropped, INFO
    reduced to AF, genotypes and samples removed, optional chr prefix; with dict_path (`-d / --dict`, PrepareVcf.scala:36, 46-56)
    the header's contig lines and `reference` line are rewritten from that sequence dictionary."""
    fix = set([str(i) for i in range(1, 23)] + ["X", "Y"])
    first_header, _ = read_vcf(inputs[0])
    if dict_path is not None:
        first_header = _header_with_dictionary(first_header, read_sequence_dictionary(dict_path))
    opener = gzip.open if str(output).endswith(".gz") else open
    n = 0
    with opener(output, "wt") as out:
        for h in first_header:
            if h.startswith("#CHROM"):
                out.write("\t".join(h.split("\t")[:8]) + "\n")
            else:
                out.write(h + "\n")
        for path in inputs:
            _, vs = read_vcf(path)
            for v in vs:
                if v.filters != ("PASS",):
                    continue
                if not any(af >= min_af for af in v.afs):
                    continue
                if any(a.startswith("<") or "[" in a or "]" in a or a == "*" for a in v.alts):
                    continue
                pairs = [(a, af) for a, af in zip(v.alts, v.afs) if af >= min_af]
                chrom = "chr" + v.chrom if add_chr_prefix and v.chrom in fix else v.chrom
                out.write("\t".join([chrom, str(v.pos), v.id or ".", v.ref, ",".join(a for a, _ in pairs), v.line[5] if len(v.line) > 5 else ".",
                                     "PASS", "AF=" + ",".join(repr(float(af)) if af != int(af) else str(af) for _, af in pairs)]) + "\n")
                n += 1
    return n
