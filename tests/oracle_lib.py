"""ctypes binding for oracle/liboracle.so -- TEST INFRASTRUCTURE (see oracle/calitas_oracle.cpp header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

DEFAULT_COSTS = (-120, -260, -122, -121)  # guideMismatch, pamMismatch, genomeGap, guideGap net costs (SGA:17-22)

GA_COLUMNS = ["strand", "start", "end", "gstart", "gend", "score", "cigar", "guide", "padded_guide", "padded_alignment",
              "padded_target", "mismatches", "gap_bases", "guide_mm", "guide_gaps", "pam_mm", "pam_gaps", "chrom"]
_GA_INT = {"start", "end", "gstart", "gend", "score", "mismatches", "gap_bases", "guide_mm", "guide_gaps", "pam_mm", "pam_gaps"}

_lib = None


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("calitas_oracle.cpp", "check_hits.cpp")]
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        for name in ["oracle_align", "oracle_align_best", "oracle_align_to_ref", "oracle_guide_alignment", "oracle_windows",
                     "oracle_search_reference", "oracle_search_memory", "oracle_search_reference_vcf", "oracle_allele_combos",
                     "oracle_variant_window", "oracle_align_to_reference", "oracle_glocal"]:
            getattr(L, name).restype = ctypes.c_void_p
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_check_hits_text.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        L.oracle_check_hits_text.restype = ctypes.c_int
        _lib = L
    return _lib


def _take(ptr):
    s = ctypes.string_at(ptr).decode()
    lib().oracle_free(ptr)
    if s.startswith("ERROR\t"):
        raise RuntimeError(s[6:])
    return s


def _costs(costs):
    return (ctypes.c_int * 4)(*costs)


def _rows(text):
    out = []
    for line in text.splitlines():
        f = line.split("\t")
        d = dict(zip(GA_COLUMNS, f))
        for k in _GA_INT:
            d[k] = int(d[k])
        out.append(d)
    return out


def align(guide, target, d, g, p, D, O=0, off=0, aux=(), name="n/a", costs=DEFAULT_COSTS, switches=0):
    t = target.encode() if isinstance(target, str) else bytes(target)
    ptr = lib().oracle_align(guide.encode(), ",".join(aux).encode(), t, len(t), name.encode(), off, d, g, p, D, O, _costs(costs), switches)
    return _rows(_take(ptr))


def glocal(query, target, min_score, costs=DEFAULT_COSTS, switches=0):
    """fgbio Aligner(Glocal).align(query, target, minScore) as the oracle restates it: ["targetStart-targetEnd:score:cigar", ...]."""
    return _take(lib().oracle_glocal(query.encode(), target.encode(), min_score, _costs(costs), switches)).split()


def align_best(guide, target, aux=(), g=3, costs=DEFAULT_COSTS, switches=0):
    t = target.encode()
    ptr = lib().oracle_align_best(guide.encode(), ",".join(aux).encode(), t, len(t), g, _costs(costs), switches)
    return _rows(_take(ptr))[0]


def align_to_ref_best(guide, chrom, contig, pos, window_size=0, g=3, costs=DEFAULT_COSTS, switches=0):
    c = contig.encode()
    ptr = lib().oracle_align_to_ref(guide.encode(), chrom.encode(), c, len(c), pos, window_size, 1, 0, g, 0, 0, 0, _costs(costs), switches)
    return _rows(_take(ptr))[0]


def guide_alignment(pg, pa, pt, start, end, strand):
    ptr = lib().oracle_guide_alignment(pg.encode(), pa.encode(), pt.encode(), start, end, ctypes.c_char(strand.encode()))
    return [int(x) for x in _take(ptr).split()]


def windows(fasta, window, step, chrom=""):
    ptr = lib().oracle_windows(fasta.encode(), window, step, chrom.encode())
    out = []
    for line in _take(ptr).splitlines():
        n, s, e, ln = line.split("\t")
        out.append((n, int(s), int(e), int(ln)))
    return out


def _iparams(window_size=1000, d=5, p=1, g=3, D=-1, O=10, m=-120, M=-260, b=-122, B=-121, max_variants=16, threads=1, switches=0):
    return (ctypes.c_int * 13)(window_size, d, p, g, D, O, m, M, b, B, max_variants, threads, switches)


def search_reference(fasta, guide, guide_id="a", aux=(), chrom="", **kw):
    """Returns (header, rows) of the hits.txt the reference algorithm produces for this FASTA."""
    nwin = ctypes.c_long(0)
    ptr = lib().oracle_search_reference(fasta.encode(), guide.encode(), guide_id.encode(), ",".join(aux).encode(), _iparams(**kw),
                                        chrom.encode(), ctypes.byref(nwin))
    lines = _take(ptr).splitlines()
    header = lines[0].split("\t")
    rows = [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]
    return header, rows, nwin.value


def search_memory(names, seqs, guide, guide_id="a", aux=(), **kw):
    """seqs: list of bytes objects (ASCII bases)."""
    n = len(names)
    c_names = (ctypes.c_char_p * n)(*[s.encode() for s in names])
    c_seqs = (ctypes.c_char_p * n)(*seqs)
    c_lens = (ctypes.c_long * n)(*[len(s) for s in seqs])
    nwin = ctypes.c_long(0)
    ptr = lib().oracle_search_memory(n, c_names, c_seqs, c_lens, guide.encode(), guide_id.encode(), ",".join(aux).encode(),
                                     _iparams(**kw), ctypes.byref(nwin))
    lines = _take(ptr).splitlines()
    header = lines[0].split("\t")
    rows = [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]
    return header, rows, nwin.value


def search_reference_vcf(fasta, vcf, guide, guide_id="a", aux=(), chrom="", **kw):
    nwin = ctypes.c_long(0)
    ptr = lib().oracle_search_reference_vcf(fasta.encode(), guide.encode(), guide_id.encode(), ",".join(aux).encode(), _iparams(**kw),
                                            chrom.encode(), vcf.encode(), ctypes.byref(nwin))
    lines = _take(ptr).splitlines()
    header = lines[0].split("\t")
    rows = [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]
    return header, rows, nwin.value


def check_hits_text(text, names, max_overlap=10, threads=16, nbytes=None):
    """Properties of a finished hits.txt (oracle/check_hits.cpp): ReferenceHit.sort order (RH:284), consecutive kept hits of a
    (chromosome, strand, variant_description) group overlapping by less than maxOverlap (SR:653-675, RH:135-144), 34 columns.
    text: bytes, or an address with nbytes (a text of gigabytes stays where the library put it).  Returns a dict of counts."""
    out = (ctypes.c_uint64 * 6)()
    if isinstance(text, (bytes, bytearray)):
        held = bytes(text)                                     # (stays alive for the call)
        buf, n = ctypes.cast(ctypes.c_char_p(held), ctypes.c_void_p), len(held)
    else:
        buf, n = ctypes.c_void_p(text), nbytes
    rc = lib().oracle_check_hits_text(buf, n, "\n".join(names).encode(), max_overlap, threads, out)
    if rc != 0:
        raise RuntimeError("not a hits.txt (no header line)")
    return dict(zip(("rows", "rows_with_variant", "out_of_order", "overlapping", "malformed", "threads"), [int(x) for x in out]))


def align_to_reference(fasta, input_tsv, limits=None, g=3, D=-1, window_size=0, costs=DEFAULT_COSTS, switches=0):
    """AlignToReference on files; limits = (d, p, O) or None for the best alignment per query. Returns (header, rows)."""
    d, p, O = limits if limits else (0, 0, 0)
    ip = (ctypes.c_int * 12)(1 if limits else 0, d, p, g, D, O, window_size, costs[0], costs[1], costs[2], costs[3], switches)
    lines = _take(lib().oracle_align_to_reference(fasta.encode(), input_tsv.encode(), ip)).splitlines()
    header = lines[0].split("\t")
    return header, [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]


def allele_combos(counts):
    arr = (ctypes.c_int * len(counts))(*counts)
    return [[int(x) for x in ln.split(",")] for ln in _take(lib().oracle_allele_combos(arr, len(counts))).splitlines()]


def _vspec(variants):
    return ",".join("%d:%s:%s:%s" % (pos, vid or ".", ref, "/".join(alts)) for pos, vid, ref, alts in variants)


def allele_sets(variants, max_variants, chrom="chr1"):
    out = _take(lib().oracle_variant_window(chrom.encode(), b"", _vspec(variants).encode(), b"", 0, max_variants, b""))
    return [ln.split(",") for ln in out.splitlines()]


def build_variant_window(ref, variants, alleles, padding, queries=(), chrom="chr1"):
    q = ",".join("%d:%d" % (off, 1 if prec else 0) for off, prec in queries)
    out = _take(lib().oracle_variant_window(chrom.encode(), ref.encode(), _vspec(variants).encode(), ",".join(map(str, alleles)).encode(),
                                            padding, 16, q.encode())).splitlines()
    bases, cigar, start = out[0].split("\t")
    return bases, cigar, int(start), [int(x) for x in out[1:]]
