"""ctypes binding of libcalitas_hip.so (C ABI: include/calitas_hip.h).

The library is built in-tree by `make -C calitas_amd/csrc` (see __graft_entry__.build).  There is no Python or CPU
fallback: if the shared library is missing, importing this module raises; if no GPU is present, creating a device
context raises CalitasError(ENODEV).
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CALITAS_LIB_PATH: another build of the same library -- tools/sanitize.sh points it at the host-side sanitizer builds
# (make SAN=address|thread).  Still a native library: there is no Python or CPU implementation to fall back to.
LIB_PATH = os.environ.get("CALITAS_LIB_PATH") or os.path.join(HERE, "libcalitas_hip.so")

MAX_OPS = 128
OK, EINVAL, ENODEV, EHIP, EIO, ESTATE, ENOMEM = 0, 1, 2, 3, 4, 5, 6


class CalitasError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("calitas error %d: %s" % (code, message))
        self.code = code


class GuideT(ctypes.Structure):
    _fields_ = [("protospacer", ctypes.c_char_p), ("n_pams", ctypes.c_int32), ("pams", ctypes.POINTER(ctypes.c_char_p)),
                ("pam_is_5prime", ctypes.c_int32), ("cli_length", ctypes.c_int32)]


class ParamsT(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "window_size", "max_guide_diffs", "max_pam_mismatches", "max_gaps_between_guide_and_pam", "max_total_diffs", "max_overlap",
        "guide_mismatch_net_cost", "pam_mismatch_net_cost", "genome_gap_net_cost", "guide_gap_net_cost", "chrom_index",
        "eqx_by_score", "max_variants", "first_window", "n_windows")]


class AlnT(ctypes.Structure):
    _fields_ = [("guide_index", ctypes.c_int32), ("contig_index", ctypes.c_int32), ("window_start", ctypes.c_int32),
                ("start_offset", ctypes.c_int32), ("end_offset", ctypes.c_int32), ("guide_start_offset", ctypes.c_int32),
                ("guide_end_offset", ctypes.c_int32), ("score", ctypes.c_int32), ("strand", ctypes.c_int8),
                ("pam_index", ctypes.c_int8), ("n_ops", ctypes.c_int16), ("ops", ctypes.c_uint8 * MAX_OPS)]


class ExtHitT(ctypes.Structure):
    _fields_ = [("contig_index", ctypes.c_int32), ("coordinate_start", ctypes.c_int32), ("end", ctypes.c_int32), ("score", ctypes.c_int32),
                ("strand", ctypes.c_int8), ("variant_description", ctypes.c_char_p), ("row", ctypes.c_char_p)]


class TimingT(ctypes.Structure):
    _fields_ = [("scan_kernel_ms", ctypes.c_double), ("align_kernel_ms", ctypes.c_double), ("gpu_total_ms", ctypes.c_double),
                ("host_post_ms", ctypes.c_double), ("bases_scanned", ctypes.c_uint64), ("packed_bytes", ctypes.c_uint64),
                ("scan_records", ctypes.c_uint64), ("candidate_columns", ctypes.c_uint64), ("raw_alignments", ctypes.c_uint64),
                ("accepted_alignments", ctypes.c_uint64), ("retries", ctypes.c_uint32), ("lanes", ctypes.c_uint32),
                ("hits_kernel_ms", ctypes.c_double), ("hits_copy_ms", ctypes.c_double), ("hit_rows", ctypes.c_uint64),
                ("hits_bytes", ctypes.c_uint64), ("contig_passes", ctypes.c_uint32), ("binned_lanes", ctypes.c_uint32),
                ("owned_general_lanes", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


# every symbol include/calitas_hip.h declares
SYMBOLS = ["calitas_create", "calitas_destroy", "calitas_last_error", "calitas_free", "calitas_set_reference",
           "calitas_set_reference_fasta", "calitas_save_index", "calitas_load_index", "calitas_reference_info", "calitas_contig_name", "calitas_genome_build", "calitas_fetch_bases", "calitas_expand_rows",
           "calitas_window_table", "calitas_search", "calitas_search_hits", "calitas_search_hits_stream", "calitas_search_hits_into", "calitas_pin_host", "calitas_unpin_host", "calitas_alloc_host", "calitas_release_parked", "calitas_search_hits_batch", "calitas_get_timing", "calitas_scan_candidates", "calitas_scan_candidates_columnwise", "calitas_contig_packed_base", "calitas_reference_tiles", "calitas_window_filter", "calitas_hits_tsv", "calitas_hits_tsv_ext", "calitas_search_variants", "calitas_search_variants_into", "calitas_vcf_identifier", "calitas_vcf_records",
           "calitas_padded_strings", "calitas_align_windows", "calitas_padded_strings_target", "calitas_version", "calitas_switches", "calitas_reap_wait"]

if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: build it with `make -C calitas_amd/csrc` (hipcc, gfx950). "
                      "calitas_amd has no fallback implementation." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)
lib.calitas_last_error.restype = ctypes.c_char_p
lib.calitas_last_error.argtypes = [ctypes.c_void_p]
lib.calitas_version.restype = ctypes.c_char_p
lib.calitas_switches.restype = ctypes.c_char_p
if hasattr(lib, "calitas_reap_wait"):
    lib.calitas_reap_wait.restype = None
    lib.calitas_reap_wait.argtypes = []
lib.calitas_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
lib.calitas_destroy.argtypes = [ctypes.c_void_p]
lib.calitas_destroy.restype = None
lib.calitas_free.argtypes = [ctypes.c_void_p]
lib.calitas_free.restype = None
lib.calitas_set_reference.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint64),
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p]
lib.calitas_set_reference_fasta.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
lib.calitas_save_index.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
lib.calitas_load_index.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
lib.calitas_reference_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_contig_name.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_fetch_bases.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p]
if hasattr(lib, "calitas_expand_rows"):   # (an older build behind CALITAS_LIB_PATH, for A/B runs, does not have it)
    lib.calitas_expand_rows.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p,
                                        ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_window_table.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_search.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(GuideT), ctypes.POINTER(ParamsT),
                               ctypes.POINTER(ctypes.POINTER(AlnT)), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_scan_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(GuideT), ctypes.POINTER(ParamsT),
                                        ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32)), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_scan_candidates_columnwise.argtypes = lib.calitas_scan_candidates.argtypes
lib.calitas_reference_tiles.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_uint64)] * 4
lib.calitas_contig_packed_base.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_get_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(TimingT)]
lib.calitas_window_filter.argtypes = [ctypes.POINTER(AlnT), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
lib.calitas_hits_tsv.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT),
                                 ctypes.POINTER(AlnT), ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p,
                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_search_hits.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT), ctypes.c_char_p,
                                    ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_search_hits_batch.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(GuideT), ctypes.POINTER(ctypes.c_char_p),
                                          ctypes.POINTER(ParamsT), ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_search_hits_into.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT), ctypes.c_char_p,
                                         ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_pin_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
lib.calitas_unpin_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.calitas_genome_build.restype = ctypes.c_char_p
lib.calitas_genome_build.argtypes = [ctypes.c_void_p]
TextSink = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
lib.calitas_search_hits_stream.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT), ctypes.c_char_p,
                                           ctypes.c_char_p, TextSink, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_search_variants.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT), ctypes.c_char_p,
                                        ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
if hasattr(lib, "calitas_search_variants_into"):   # (an older build behind CALITAS_LIB_PATH, for A/B runs, does not have it)
    lib.calitas_search_variants_into.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT), ctypes.c_char_p,
                                                 ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64,
                                                 ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
if hasattr(lib, "calitas_alloc_host"):
    lib.calitas_alloc_host.argtypes = [ctypes.c_uint64]
    lib.calitas_alloc_host.restype = ctypes.c_void_p
if hasattr(lib, "calitas_vcf_records"):
    lib.calitas_vcf_identifier.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.calitas_vcf_records.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_hits_tsv_ext.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.c_char_p, ctypes.POINTER(ParamsT), ctypes.POINTER(AlnT),
                                     ctypes.c_uint64, ctypes.POINTER(ExtHitT), ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p,
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
lib.calitas_padded_strings.argtypes = [ctypes.c_void_p, ctypes.POINTER(GuideT), ctypes.POINTER(AlnT), ctypes.c_char_p,
                                       ctypes.c_char_p, ctypes.c_char_p]


lib.calitas_align_windows.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(GuideT), ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ParamsT),
                                      ctypes.POINTER(ctypes.POINTER(AlnT)), ctypes.POINTER(ctypes.c_uint64),
                                      ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
lib.calitas_padded_strings_target.argtypes = [ctypes.POINTER(GuideT), ctypes.POINTER(AlnT), ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int32,
                                              ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]


def check(ctx, rc):
    if rc != OK:
        msg = lib.calitas_last_error(ctx)
        raise CalitasError(rc, msg.decode() if msg else "?")
