"""Host-side mirror of the reference's operator interface for the SearchReference path, on top of the C ABI.

Names, argument meaning and error behaviour follow
  calitas/src/main/scala/com/editasmedicine/aligner/SequentialGuideAligner.scala (Guide, Defaults)
  calitas/src/main/scala/com/editasmedicine/aligner/SearchReference.scala (SearchReference flags, execute)
so the parity tests read like the reference's own tests.  All alignment work happens in libcalitas_hip.so.
"""
import contextlib
import ctypes
import time

from . import _lib
from ._lib import AlnT, CalitasError, GuideT, ParamsT, TimingT, lib


class Defaults:  # SequentialGuideAligner.scala:17-28
    MismatchNetCost = -120
    GuideGapNetCost = -121
    GenomeGapNetCost = -122
    PamMismatchNetCost = -260
    MaxGuideDiffs = 5
    MaxPamMismatches = 1
    MaxGapsBetweenGuideAndPam = 3
    MaxOverlap = 10
    MaxVariantsInCluster = 16


def _split_by_case(s):  # SequentialGuideAligner.scala:110-121
    parts, i = [], 0
    while i < len(s):
        first = s[i].islower()
        j = i
        while j < len(s) and s[j].islower() == first:
            j += 1
        parts.append(s[i:j])
        i = j
    return parts


class Guide:
    """SequentialGuideAligner.Guide (SequentialGuideAligner.scala:32-107): protospacer in upper case, optional PAM in
    lower case at either end, optional auxiliary PAMs."""

    def __init__(self, sequence, aux_pams=()):
        aux_pams = list(aux_pams)
        self.sequence = sequence
        parts = _split_by_case(sequence.strip())
        if not (1 <= len(parts) <= 2):
            raise ValueError("Invalid Guide sequence %s." % sequence)
        if not (len(parts) == 2 or parts[0][0].isupper()):
            raise ValueError("Guide sequence cannot be all lower case.")
        if aux_pams and len(parts) != 2:
            raise ValueError("Cannot provide auxiliary PAMs without providing a PAM in the guide sequence.")
        if any(p != p.lower() for p in aux_pams):
            raise ValueError("All PAMs must be lower case. PAMs given: %s" % ", ".join(aux_pams))
        if len(parts) == 1:
            guide, pam, five = parts[0], None, False
        elif parts[0][0].isupper():
            guide, pam, five = parts[0], parts[1], False
        else:
            guide, pam, five = parts[1], parts[0], True
        self.guide = guide.upper()
        self.pams = ([pam] if pam is not None else []) + aux_pams
        self.pams = [p.lower() for p in self.pams]
        self.pam_is_five_prime = five
        self.pam_is_three_prime = pam is not None and not five
        self.protospacer_length = len(self.guide)
        self.pam_length = max([len(p) for p in self.pams], default=0)
        self.length = self.protospacer_length + self.pam_length
        self.cli_length = len(sequence)  # SearchReference.scala:528 uses the raw `-i` string

    def to_c(self):
        g = GuideT()
        g.protospacer = self.guide.encode()
        g.n_pams = len(self.pams)
        self._pam_arr = (ctypes.c_char_p * max(1, len(self.pams)))(*[p.encode() for p in self.pams])
        g.pams = self._pam_arr
        g.pam_is_5prime = 1 if self.pam_is_five_prime else 0
        g.cli_length = self.cli_length
        return g


def make_params(window_size=1000, max_guide_diffs=Defaults.MaxGuideDiffs, max_pam_mismatches=Defaults.MaxPamMismatches,
                max_gaps_between_guide_and_pam=Defaults.MaxGapsBetweenGuideAndPam, max_total_diffs=None,
                max_overlap=Defaults.MaxOverlap, guide_mismatch_net_cost=Defaults.MismatchNetCost,
                pam_mismatch_net_cost=Defaults.PamMismatchNetCost, genome_gap_net_cost=Defaults.GenomeGapNetCost,
                guide_gap_net_cost=Defaults.GuideGapNetCost, chrom_index=-1, eqx_by_score=0, per_matrix=0,
                max_variants=Defaults.MaxVariantsInCluster, first_window=0, n_windows=0):
    p = ParamsT()
    p.window_size = window_size
    p.max_guide_diffs = max_guide_diffs
    p.max_pam_mismatches = max_pam_mismatches
    p.max_gaps_between_guide_and_pam = max_gaps_between_guide_and_pam
    p.max_total_diffs = -1 if max_total_diffs is None else max_total_diffs
    p.max_overlap = max_overlap
    p.guide_mismatch_net_cost = guide_mismatch_net_cost
    p.pam_mismatch_net_cost = pam_mismatch_net_cost
    p.genome_gap_net_cost = genome_gap_net_cost
    p.guide_gap_net_cost = guide_gap_net_cost
    p.chrom_index = chrom_index
    p.eqx_by_score = (eqx_by_score & 3) | (2 if per_matrix else 0)   # bit flags, see calitas_hip.h
    p.max_variants = max_variants
    p.first_window, p.n_windows = first_window, n_windows   # calitas_search only: a window range of the job (shard.window_partition)
    return p


class Alignment:
    """One GuideAlignment (GuideAlignment.scala:72-88) as returned through the C ABI."""
    __slots__ = ("guide_index", "contig_index", "window_start", "start_offset", "end_offset", "guide_start_offset",
                 "guide_end_offset", "score", "strand", "pam_index", "ops")

    def __init__(self, a):
        self.guide_index = a.guide_index
        self.contig_index = a.contig_index
        self.window_start = a.window_start
        self.start_offset = a.start_offset
        self.end_offset = a.end_offset
        self.guide_start_offset = a.guide_start_offset
        self.guide_end_offset = a.guide_end_offset
        self.score = a.score
        self.strand = chr(a.strand)
        self.pam_index = a.pam_index
        self.ops = bytes(a.ops[:a.n_ops]).decode()

    @property
    def cigar(self):
        out, i = [], 0
        while i < len(self.ops):
            j = i
            while j < len(self.ops) and self.ops[j] == self.ops[i]:
                j += 1
            out.append("%d%s" % (j - i, self.ops[i]))
            i = j
        return "".join(out)

    def to_c(self):
        a = AlnT()
        for f in ("guide_index", "contig_index", "window_start", "start_offset", "end_offset", "guide_start_offset",
                  "guide_end_offset", "score", "pam_index"):
            setattr(a, f, getattr(self, f))
        a.strand = ord(self.strand)
        a.n_ops = len(self.ops)
        for i, c in enumerate(self.ops.encode()):
            a.ops[i] = c
        return a


class Context:
    """One GPU (device >= 0) or a host-only context (device = -1) holding a packed reference."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        rc = lib.calitas_create(device, ctypes.byref(h))
        if rc != _lib.OK:
            msg = lib.calitas_last_error(None)
            raise CalitasError(rc, msg.decode() if msg else "?")
        self._h = h
        self.device = device
        self.contig_names = []
        self.contig_lengths = []

    def close(self):
        if self._h:
            lib.calitas_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _load_contig_table(self):
        n = ctypes.c_int32()
        _lib.check(self._h, lib.calitas_reference_info(self._h, ctypes.byref(n), None, None))
        self.contig_names, self.contig_lengths = [], []
        for i in range(n.value):
            nm, ln = ctypes.c_char_p(), ctypes.c_uint64()
            _lib.check(self._h, lib.calitas_contig_name(self._h, i, ctypes.byref(nm), ctypes.byref(ln)))
            self.contig_names.append(nm.value.decode())
            self.contig_lengths.append(ln.value)

    def set_reference(self, names, seqs, genome_build="unknown", lengths=None):
        """seqs: bytes-like objects or numpy uint8 arrays holding ASCII bases; borrowed only for the call.  seqs[i] None with
        lengths[i] given: contig i is absent (its name and length count, its bases are not held: a process of a multi-GPU job
        and the contigs its window range does not touch)."""
        n = len(names)
        c_names = (ctypes.c_char_p * n)(*[s.encode() for s in names])
        c_lens = (ctypes.c_uint64 * n)(*[(int(lengths[i]) if s is None else len(s)) for i, s in enumerate(seqs)])
        ptrs, keep = [], []
        for s in seqs:
            if s is None:
                ptrs.append(ctypes.c_void_p(None))
            elif hasattr(s, "ctypes"):  # numpy array
                keep.append(s)
                ptrs.append(ctypes.c_void_p(s.ctypes.data))
            else:
                b = bytes(s) if not isinstance(s, bytes) else s
                keep.append(b)
                ptrs.append(ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p))
        c_ptrs = (ctypes.c_void_p * n)(*ptrs)
        _lib.check(self._h, lib.calitas_set_reference(self._h, n, c_names, c_lens, c_ptrs, genome_build.encode()))
        self._load_contig_table()

    def set_reference_fasta(self, path):
        _lib.check(self._h, lib.calitas_set_reference_fasta(self._h, str(path).encode()))
        self._load_contig_table()

    def save_index(self, path):
        _lib.check(self._h, lib.calitas_save_index(self._h, str(path).encode()))

    def load_index(self, path):
        _lib.check(self._h, lib.calitas_load_index(self._h, str(path).encode()))
        self._load_contig_table()

    def genome_build(self):
        return lib.calitas_genome_build(self._h).decode()

    def reference_info(self):
        n, tb, pb = ctypes.c_int32(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_reference_info(self._h, ctypes.byref(n), ctypes.byref(tb), ctypes.byref(pb)))
        return {"n_contigs": n.value, "total_bases": tb.value, "packed_bytes": pb.value}

    def fetch_bases(self, contig_index, start, length):
        buf = ctypes.create_string_buffer(length + 1)
        _lib.check(self._h, lib.calitas_fetch_bases(self._h, contig_index, start, length, buf))
        return buf.raw[:length].decode()

    def window_table(self, window_size, step, min_length, chrom_index=-1):
        out, n = ctypes.POINTER(ctypes.c_int32)(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_window_table(self._h, window_size, step, min_length, chrom_index, ctypes.byref(out), ctypes.byref(n)))
        rows = [(out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i in range(n.value)]
        lib.calitas_free(out)
        return rows

    def search_raw(self, guides, params):
        """calitas_search; returns (ctypes array pointer, count) -- caller must free with lib.calitas_free."""
        n = len(guides)
        self._keep = [g.to_c() for g in guides]
        arr = (GuideT * n)(*self._keep)
        out, cnt = ctypes.POINTER(AlnT)(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_search(self._h, n, arr, ctypes.byref(params), ctypes.byref(out), ctypes.byref(cnt)))
        return out, cnt.value

    def tile_census(self):
        """calitas_reference_tiles: {"tiles", "dead", "masked", "tile_bases"} of the resident reference."""
        v = [ctypes.c_uint64() for _ in range(4)]
        _lib.check(self._h, lib.calitas_reference_tiles(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("tiles", "dead", "masked", "tile_bases"), [x.value for x in v]))

    def search_variants_raw(self, guide, guide_id, params, vcf_path, version=None, time_stamp=None, chrom=None):
        """calitas_search_variants without bringing the text into Python: (n_bytes, n_rows, n_variant_windows)."""
        g = guide.to_c()
        tsv, nbytes, rows, nwin = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_search_variants(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params), str(vcf_path).encode(),
                                                        chrom.encode() if chrom is not None else None, None,
                                                        version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                        ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows), ctypes.byref(nwin)))
        t_free = time.perf_counter()
        lib.calitas_free(tsv)
        self.last_free_ms = (time.perf_counter() - t_free) * 1e3    # (a hits.txt of tens of gigabytes: handing the block back is not free)
        return nbytes.value, rows.value, nwin.value

    def vcf_identifier(self, vcf_path):
        """calitas_vcf_identifier: "name:md5" of a VCF (ReferenceHit.scala:175-183), computed by the library."""
        out = ctypes.c_void_p()
        _lib.check(self._h, lib.calitas_vcf_identifier(self._h, str(vcf_path).encode(), ctypes.byref(out)))
        try:
            return ctypes.string_at(out).decode()
        finally:
            lib.calitas_free(out)

    def vcf_records(self, vcf_path, chrom=None):
        """calitas_vcf_records: the records of a VCF as the variant search reads them, a list of
        (chrom, pos, end, id, ref, [alts], [afs as the floats the search keeps])."""
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_vcf_records(self._h, str(vcf_path).encode(), chrom.encode() if chrom is not None else None, ctypes.byref(out), ctypes.byref(n)))
        try:
            text = ctypes.string_at(out).decode()
        finally:
            lib.calitas_free(out)
        recs = []
        for line in text.split("\n")[:-1]:
            c, pos, end, vid, ref, alts, afs = line.split("\t")
            recs.append((c, int(pos), int(end), vid, ref, alts.split(","), [float(x) for x in afs.split(",")] if afs else []))
        assert len(recs) == n.value
        return recs

    def search_variants_into(self, guide, guide_id, params, vcf_path, address, capacity, version=None, time_stamp=None, chrom=None):
        """calitas_search_variants_into: the text goes to `capacity` bytes at `address` (memory of the caller, page-locked with pin_host: every
        contig's rows then cross the bus straight to their place).  Returns (n_bytes, n_rows, n_variant_windows)."""
        g = guide.to_c()
        nbytes, rows, nwin = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_search_variants_into(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params), str(vcf_path).encode(),
                                                             chrom.encode() if chrom is not None else None, None,
                                                             version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                             ctypes.c_void_p(address), capacity, ctypes.byref(nbytes), ctypes.byref(rows), ctypes.byref(nwin)))
        return nbytes.value, rows.value, nwin.value

    def scan_candidates(self, guides, params, columnwise=False):
        """calitas_scan_candidates: the candidate filter alone (columnwise: the same set from round 1's kernel, a test hook).  Returns a sorted list of (contig_index, contig_offset, pass, guide):
        one entry per end column whose seamless glocal bottom-row score reaches minGuideScore (pass 0 = target as is, the column
        is the alignment's last base; pass 1 = reverse-complemented target, the column is its first base in contig coordinates).
        Columns in the padding behind a contig are dropped."""
        import bisect
        n = len(guides)
        keep = [g.to_c() for g in guides]
        arr = (GuideT * n)(*keep)
        out, cnt = ctypes.POINTER(ctypes.c_uint32)(), ctypes.c_uint64()
        fn = lib.calitas_scan_candidates_columnwise if columnwise else lib.calitas_scan_candidates
        _lib.check(self._h, fn(self._h, n, arr, ctypes.byref(params), ctypes.byref(out), ctypes.byref(cnt)))
        try:
            words = out[:2 * cnt.value]
        finally:
            lib.calitas_free(out)
        nc = self.reference_info()["n_contigs"]
        bases, lens = [], []
        for i in range(nc):
            g, nm, ln = ctypes.c_uint64(), ctypes.c_char_p(), ctypes.c_uint64()
            _lib.check(self._h, lib.calitas_contig_packed_base(self._h, i, ctypes.byref(g)))
            _lib.check(self._h, lib.calitas_contig_name(self._h, i, ctypes.byref(nm), ctypes.byref(ln)))
            bases.append(g.value); lens.append(ln.value)
        res = []
        for k in range(cnt.value):
            gword, info = words[2 * k], words[2 * k + 1]
            c = bisect.bisect_right(bases, gword * 16) - 1
            for b in range(16):
                if (info >> b) & 1:
                    off = gword * 16 + b - bases[c]
                    if 0 <= off < lens[c]:
                        res.append((c, off, (info >> 16) & 1, (info >> 17) & 0x7F))
        res.sort()
        return res

    def search(self, guides, params):
        """Per-window accepted alignments of every guide, in the reference's order (list of Alignment)."""
        out, n = self.search_raw(guides, params)
        try:
            return [Alignment(out[i]) for i in range(n)]
        finally:
            lib.calitas_free(out)

    def search_hits(self, guide, guide_id, params, version=None, time_stamp=None, decode=True):
        """calitas_search_hits: one guide against the resident reference, finished hits.txt text back (tsv_text, n_rows);
        decode=False skips the copy into a Python str and returns (n_bytes, n_rows); decode="bytes" returns the raw bytes."""
        g = guide.to_c()
        tsv, nbytes, rows = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_search_hits(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params),
                                                    version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                    ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows)))
        if decode == "bytes":
            text = ctypes.string_at(tsv, nbytes.value)
        else:
            text = ctypes.string_at(tsv, nbytes.value).decode() if decode else nbytes.value
        lib.calitas_free(tsv)
        return text, rows.value

    def search_hits_into(self, guide, guide_id, params, address, capacity, version=None, time_stamp=None):
        """calitas_search_hits_into: the text goes to `capacity` bytes at `address` (memory of the caller, ideally pinned with
        pin_host).  Returns (n_bytes, n_rows)."""
        g = guide.to_c()
        nbytes, rows = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_search_hits_into(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params),
                                                         version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                         ctypes.c_void_p(address), capacity, ctypes.byref(nbytes), ctypes.byref(rows)))
        return nbytes.value, rows.value

    @staticmethod
    def alloc_host(nbytes):
        """calitas_alloc_host: the address of a page-locked block of the runtime's own (free it with free_host) -- the destination the
        *_into calls like best."""
        p = lib.calitas_alloc_host(nbytes)
        if not p:
            raise MemoryError("calitas_alloc_host(%d)" % nbytes)
        return p

    @staticmethod
    def free_host(address):
        lib.calitas_free(ctypes.c_void_p(address))

    def pin_host(self, address, nbytes):
        _lib.check(self._h, lib.calitas_pin_host(self._h, ctypes.c_void_p(address), nbytes))

    def unpin_host(self, address):
        _lib.check(self._h, lib.calitas_unpin_host(self._h, ctypes.c_void_p(address)))

    def search_hits_stream(self, guide, guide_id, params, write, version=None, time_stamp=None):
        """calitas_search_hits_stream: `write(memoryview)` receives consecutive pieces of hits.txt (one piece when the search fits
        one call; header, then per-contig pieces when it does not fit the device).  A piece is the library's own buffer, valid only
        during the call -- file.write() takes it as it is, bytes(piece) keeps a copy.  Returns (n_bytes, n_rows)."""
        g = guide.to_c()
        nbytes, rows = ctypes.c_uint64(), ctypes.c_uint64()
        failure = []

        def sink(piece, n, _user):
            try:
                write(memoryview((ctypes.c_char * n).from_address(piece)).cast("B"))
                return 0
            except Exception as e:          # an exception must not cross the C frames
                failure.append(e)
                return 1
        cb = _lib.TextSink(sink)
        rc = lib.calitas_search_hits_stream(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params),
                                            version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                            cb, None, ctypes.byref(nbytes), ctypes.byref(rows))
        if failure:
            raise failure[0]
        _lib.check(self._h, rc)
        return nbytes.value, rows.value

    @contextlib.contextmanager
    def search_hits_view(self, guide, guide_id, params, version=None, time_stamp=None):
        """calitas_search_hits without a copy: yields (memoryview over the library's text buffer, n_rows); the buffer is
        released when the block ends."""
        g = guide.to_c()
        tsv, nbytes, rows = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_search_hits(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params),
                                                    version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                    ctypes.byref(tsv), ctypes.byref(nbytes), ctypes.byref(rows)))
        try:
            yield memoryview((ctypes.c_char * nbytes.value).from_address(tsv.value)).cast("B"), rows.value
        finally:
            lib.calitas_free(tsv)

    def search_hits_batch(self, guides, guide_ids, params, version=None, time_stamp=None, decode=True):
        """calitas_search_hits_batch: a list of (tsv_text or n_bytes, n_rows), one per guide, pipelined on the device."""
        n = len(guides)
        keep = [g.to_c() for g in guides]
        garr = (GuideT * n)(*keep)
        ids = (ctypes.c_char_p * n)(*[i.encode() for i in guide_ids])
        tsv = (ctypes.c_void_p * n)()
        nbytes, rows = (ctypes.c_uint64 * n)(), (ctypes.c_uint64 * n)()
        _lib.check(self._h, lib.calitas_search_hits_batch(self._h, n, garr, ids, ctypes.byref(params),
                                                          version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                          tsv, nbytes, rows))
        out = []
        for i in range(n):
            if decode == "digest":      # a checksum of the text instead of the text (a full-size batch is many gigabytes)
                import zlib
                what = (zlib.crc32((ctypes.c_char * nbytes[i]).from_address(tsv[i])), nbytes[i])
            else:
                what = ctypes.string_at(tsv[i], nbytes[i]).decode() if decode else nbytes[i]
            out.append((what, rows[i]))
            lib.calitas_free(tsv[i])
        return out

    def timing(self):
        t = TimingT()
        _lib.check(self._h, lib.calitas_get_timing(self._h, ctypes.byref(t)))
        return {f: getattr(t, f) for f, _ in TimingT._fields_}

    def hits_tsv_raw(self, guide, guide_id, params, alns_ptr, n, version=None, time_stamp=None, decode=True):
        """calitas_hits_tsv on a C array of alignments. decode=False returns (None, n_rows) without copying the text
        into a Python str (the rows are still built)."""
        g = guide.to_c()
        tsv, rows = ctypes.c_void_p(), ctypes.c_uint64()
        _lib.check(self._h, lib.calitas_hits_tsv(self._h, ctypes.byref(g), guide_id.encode(), ctypes.byref(params), alns_ptr, n,
                                                 version.encode() if version else None, time_stamp.encode() if time_stamp else None,
                                                 ctypes.byref(tsv), ctypes.byref(rows)))
        text = ctypes.string_at(tsv).decode() if decode else None
        lib.calitas_free(tsv)
        return text, rows.value

    def hits_tsv(self, guide, guide_id, params, alignments, version=None, time_stamp=None):
        n = len(alignments)
        arr = (AlnT * max(1, n))(*[a.to_c() for a in alignments])
        return self.hits_tsv_raw(guide, guide_id, params, arr, n, version, time_stamp)

    def padded_strings(self, guide, aln):
        g, a = guide.to_c(), aln.to_c()
        bufs = [ctypes.create_string_buffer(_lib.MAX_OPS + 1) for _ in range(3)]
        _lib.check(self._h, lib.calitas_padded_strings(self._h, ctypes.byref(g), ctypes.byref(a), *bufs))
        return tuple(b.value.decode() for b in bufs)


def window_filter(alignments, max_total_diffs, max_overlap):
    """SequentialGuideAligner.scala:315-320 on one window's alignments; returns the survivors in output order."""
    n = len(alignments)
    arr = (AlnT * max(1, n))(*[a.to_c() for a in alignments])
    order = (ctypes.c_int32 * max(1, n))()
    kept = ctypes.c_int32()
    rc = lib.calitas_window_filter(arr, n, max_total_diffs, max_overlap, order, ctypes.byref(kept))
    if rc != _lib.OK:
        raise CalitasError(rc, "calitas_window_filter")
    return [alignments[order[i]] for i in range(kept.value)]


class SearchReference:
    """Mirror of the SearchReference tool (SearchReference.scala:451-649), reference-only branch.

    new SearchReference(guide=..., guideId=..., ref=..., output=...).execute() in the reference becomes
    SearchReference(guide=..., guide_id=..., ref=..., output=...).execute() here; `threads` is accepted and ignored
    (the GPU replaces the thread pool).  A Context may be passed to reuse a resident reference."""

    def __init__(self, guide, guide_id, ref=None, output=None, auxiliary_pams=(), threads=8, window_size=1000,
                 max_guide_diffs=Defaults.MaxGuideDiffs, max_pam_mismatches=Defaults.MaxPamMismatches,
                 max_gaps_between_guide_and_pam=Defaults.MaxGapsBetweenGuideAndPam, max_total_diffs=None,
                 max_overlap=Defaults.MaxOverlap, guide_mismatch_net_cost=Defaults.MismatchNetCost,
                 pam_mismatch_net_cost=Defaults.PamMismatchNetCost, genome_gap_net_cost=Defaults.GenomeGapNetCost,
                 guide_gap_net_cost=Defaults.GuideGapNetCost, chrom=None, variants=None,
                 max_variants=Defaults.MaxVariantsInCluster, context=None, device=0, eqx_by_score=0, two_stage=False):
        self.variants = variants
        self.two_stage = two_stage
        self.guide_str, self.guide_id, self.ref, self.output = guide, guide_id, ref, output
        self.query = Guide(guide, auxiliary_pams)  # SearchReference.scala:511: fail early on an invalid guide
        self.chrom = chrom
        self._kw = dict(window_size=window_size, max_guide_diffs=max_guide_diffs, max_pam_mismatches=max_pam_mismatches,
                        max_gaps_between_guide_and_pam=max_gaps_between_guide_and_pam, max_total_diffs=max_total_diffs,
                        max_overlap=max_overlap, guide_mismatch_net_cost=guide_mismatch_net_cost,
                        pam_mismatch_net_cost=pam_mismatch_net_cost, genome_gap_net_cost=genome_gap_net_cost,
                        guide_gap_net_cost=guide_gap_net_cost, max_variants=max_variants, eqx_by_score=eqx_by_score)
        self.context = context
        self.device = device
        self.timing = None
        self.wall_ms = None

    def run(self, version=None, time_stamp=None):
        """Returns (tsv_text, n_rows)."""
        ctx = self.context
        own = ctx is None
        if own:
            ctx = Context(self.device)
            ctx.set_reference_fasta(self.ref)
        try:
            chrom_index = -1
            if self.chrom is not None:
                if self.chrom not in ctx.contig_names:
                    raise ValueError("Unknown chromosome: %s" % self.chrom)
                chrom_index = ctx.contig_names.index(self.chrom)
            if self.variants is not None:   # SearchReference.scala:570-630
                from . import variants as V
                return V.search_variants(self, ctx, self.variants, chrom_index, version, time_stamp)
            params = make_params(chrom_index=chrom_index, **self._kw)
            t0 = time.perf_counter()
            if self.two_stage:   # calitas_search, then calitas_hits_tsv on the host copy of the alignments
                out, n = ctx.search_raw([self.query], params)
                try:
                    self.timing = ctx.timing()
                    text, rows = ctx.hits_tsv_raw(self.query, self.guide_id, params, out, n, version, time_stamp)
                finally:
                    lib.calitas_free(out)
            else:
                text, rows = ctx.search_hits(self.query, self.guide_id, params, version, time_stamp)
                self.timing = ctx.timing()
            self.wall_ms = (time.perf_counter() - t0) * 1e3
            return text, rows
        finally:
            if own:
                ctx.close()

    def execute(self):
        text, _ = self.run()
        if self.output is None:
            import sys
            sys.stdout.write(text)
        else:
            with open(self.output, "w") as f:
                f.write(text)


def read_hits(path_or_text):
    """Metric.read[ReferenceHit]: list of dicts keyed by column name."""
    text = path_or_text
    if "\n" not in path_or_text:
        with open(path_or_text) as f:
            text = f.read()
    lines = text.splitlines()
    header = lines[0].split("\t")
    return [dict(zip(header, ln.split("\t"))) for ln in lines[1:]]
