/* calitas_hip.h -- C ABI of the MI355X-native CALITAS SearchReference hot path (libcalitas_hip.so).
 *
 * The reference (editasmedicine/calitas, Scala/JVM) has no FFI.  The narrowest seam with stable meaning is
 *   SequentialGuideAligner.align(guide, target, targetName, targetOffset, maxGuideDiffs, maxGapsBetweenGuideAndPam,
 *                                maxPamDiffs, maxTotalDiffs, maxOverlap): Seq[GuideAlignment]
 *   (calitas/src/main/scala/com/editasmedicine/aligner/SequentialGuideAligner.scala:228-236), called once per
 *   reference window from SearchReference.execute (SearchReference.scala:537-561).
 * One call per 1000-bp window is too fine for a GPU, so this ABI lifts the same contract to
 * "all windows of a resident reference x a batch of guides": calitas_search returns, window by window and in the
 * reference's order, exactly the GuideAlignments those align() calls return.  Everything below that seam
 * (the fgbio glocal DP, PAM extension, per-window overlap filter) runs in hand-written HIP kernels for gfx950;
 * everything above it (removeOverlaps, ReferenceHit rows, hits.txt) is host code with the reference's semantics.
 *
 * Conventions: plain C types only; inputs are borrowed for the duration of the call; outputs are allocated by the
 * library and released with calitas_free; every function returning int returns 0 on success and a non-zero
 * CALITAS_E* code on failure, with a message available from calitas_last_error.  Results are never truncated:
 * a device buffer overflow is detected and the search is re-run with larger buffers.
 * Threading: calls on one context are serialised by the caller; one context drives one GPU.
 */
#ifndef CALITAS_HIP_H
#define CALITAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CALITAS_OK 0
#define CALITAS_EINVAL 1   /* bad argument / unsupported configuration */
#define CALITAS_ENODEV 2   /* no usable HIP device, or context created host-only */
#define CALITAS_EHIP 3     /* HIP runtime error */
#define CALITAS_EIO 4      /* file could not be read / written */
#define CALITAS_ESTATE 5   /* call made in the wrong state (e.g. search before set_reference) */
#define CALITAS_ENOMEM 6   /* a device allocation failed (calitas_search_hits answers it with one pass per contig before giving up) */

#define CALITAS_MAX_PROTOSPACER 32   /* rows of the bit-vector scan kernel */
#define CALITAS_MAX_PAMS 8
#define CALITAS_MAX_PAM_LEN 16
#define CALITAS_MAX_OPS 128          /* padded alignment columns per record */
#define CALITAS_MAX_GUIDES 64        /* guides per calitas_search batch */

typedef struct calitas_ctx calitas_ctx;

/* Guide = SequentialGuideAligner.Guide (SequentialGuideAligner.scala:32-52): protospacer in upper case, zero or more
 * PAMs in lower case, all on the same side. */
typedef struct {
  const char* protospacer;
  int32_t n_pams;
  const char* const* pams;
  int32_t pam_is_5prime;
  int32_t cli_length;        /* length of the `-i` string (protospacer + primary PAM), SearchReference.scala:528,536 */
} calitas_guide_t;

/* Flags of SearchReference (SearchReference.scala:452-470) that reach the aligner. */
typedef struct {
  int32_t window_size;                     /* -w  default 1000 */
  int32_t max_guide_diffs;                 /* -d  default 5 */
  int32_t max_pam_mismatches;              /* -p  default 1 */
  int32_t max_gaps_between_guide_and_pam;  /* -g  default 3 */
  int32_t max_total_diffs;                 /* -D  <0 => d+g+p (SearchReference.scala:493) */
  int32_t max_overlap;                     /* -O  default 10 */
  int32_t guide_mismatch_net_cost;         /* -m  default -120 */
  int32_t pam_mismatch_net_cost;           /* -M  default -260 */
  int32_t genome_gap_net_cost;             /* -b  default -122 */
  int32_t guide_gap_net_cost;              /* -B  default -121 */
  int32_t chrom_index;                     /* -c  contig index, <0 => all */
  int32_t eqx_by_score;                    /* bit flags for the two readings of fgbio 2.0.0 that the reference's own tests do not
                                            * pin; 0 = the defaults.
                                            * bit 0 (1): '=' / 'X' by pairing score > 0 instead of IUPAC compatibility (U2; differs for a target N)
                                            * bit 1 (2): one alignment per bottom-row matrix cell (Diag, Left, Up) that reaches minScore
                                            *            instead of one per end column from the best of the three (U1) */
  int32_t max_variants;                    /* -V  only echoed into aligner_other_parameters */
  /* A window range [first_window, first_window + n_windows) of windowIterator's sequence for this (window size, step) over the whole
   * reference, contigs in order (the index calitas_window_table lists them in when min_length is 0); n_windows = 0 means all of
   * them.  The piece of a job one process of a multi-GPU partition runs (calitas_amd/shard.py window_partition):
   *  - calitas_search aligns exactly those windows: the alignments of consecutive ranges, concatenated, are those of the whole call;
   *  - calitas_search_hits / calitas_search_hits_into return the rows the range OWNS: the hits whose coordinate_start lies at or
   *    behind the start of window first_window and before the start of window first_window + n_windows.  coordinate_start is the
   *    first sort key of hits.txt, so the texts of consecutive ranges (minus their header lines) concatenate to the text of the
   *    whole call, wherever the cuts fall; removeOverlaps is exact across a cut because every process also aligns the windows
   *    around its stretch that can decide its hits.
   *  - calitas_search_hits_batch does the same for every guide of the batch (round 4: BASELINE config 4 on several GPUs is every
   *    process running all guides on its stretch), guides pipelined through the device stages as in the whole-genome batch.
   * The stream call refuses a range. */
  int32_t first_window;
  int32_t n_windows;
} calitas_params_t;

/* One GuideAlignment (GuideAlignment.scala:72-88).  Coordinates are 0-based half-open on the contig.  ops holds
 * one byte per padded column in the orientation of the guide as given on the command line:
 * '=' match, 'X' mismatch, 'I' base only in the guide (gap in genome), 'D' base only in the genome.
 * The padded strings are a pure function of (query, ops, strand, start_offset) and the reference bases. */
typedef struct {
  int32_t guide_index;
  int32_t contig_index;
  int32_t window_start;        /* 0-based offset of the (N-trimmed) window the alignment was found in */
  int32_t start_offset, end_offset;
  int32_t guide_start_offset, guide_end_offset;
  int32_t score;
  int8_t strand;               /* '+' or '-' */
  int8_t pam_index;            /* index into the guide's pams, -1 for a PAM-less guide */
  int16_t n_ops;
  uint8_t ops[CALITAS_MAX_OPS];
} calitas_aln_t;

/* Timing and volume of the last calitas_search on this context; kernel times are HIP-event measurements on the
 * stream the kernels were launched on.  Exception, for ranges whose tail ran on the per-bin kernels (binned_lanes): no event sits
 * between those kernels (each would hold the next kernel back by ~5 us), so align_kernel_ms is the difference of two stamps of the
 * device's wall clock taken by the kernels themselves, gpu_total_ms = the scan + the stamps' span up to the start of the row kernel,
 * and hits_kernel_ms = (end of the scan .. end of the row kernel, by events) - align_kernel_ms, which includes the kernel boundaries
 * of the chain.  scan_kernel_ms is always a pair of events riding on the scan's dispatch. */
typedef struct {
  double scan_kernel_ms;       /* bit-vector scan over the packed reference (both strands, all guides of the batch) */
  double align_kernel_ms;      /* banded glocal DP + traceback + PAM extension on the scan's candidates */
  double gpu_total_ms;         /* first launch to last copy-back */
  double host_post_ms;         /* per-window filter on the host */
  uint64_t bases_scanned;      /* reference bases covered by the scan (one count per base, not per strand) */
  uint64_t packed_bytes;       /* 2-bit bytes the scan reads from HBM per pass */
  uint64_t scan_records;       /* 16-base groups with at least one candidate end column */
  uint64_t candidate_columns;  /* candidate end columns (guide x strand x position) handed to the aligner kernel */
  uint64_t raw_alignments;     /* alignments surviving PAM extension, before the per-window filter */
  uint64_t accepted_alignments;
  uint32_t retries;            /* re-runs caused by device buffer overflow */
  uint32_t lanes;              /* contig ranges the last calitas_search_hits call pipelined (1 = one pass; 0 after calitas_search) */
  /* calitas_search_hits only */
  double hits_kernel_ms;       /* removeOverlaps + sorts + row text on the device */
  double hits_copy_ms;         /* the text's copy-back */
  uint64_t hit_rows;
  uint64_t hits_bytes;
  uint32_t contig_passes;        /* calitas_search_hits in one-pass-per-contig mode: passes run (0 otherwise) */
  uint32_t binned_lanes;         /* calitas_search_hits*: contig ranges / passes / guides whose tail (per-window filter ... rows) ran on the
                                  * per-bin kernels (binned.hpp); the others ran on the general kernels (same text) */
  uint32_t owned_general_lanes;  /* a window range (first_window / n_windows): pieces whose bins were crowded and whose owned rows the general
                                  * kernels decided from the same alignments (round 4; before, such a range searched its contigs whole) */
  uint32_t reserved;
} calitas_timing_t;

/* Context ------------------------------------------------------------------------------------------------------- */

/* device_id >= 0: bind to that HIP device (fails with CALITAS_ENODEV when there is none -- there is no CPU fallback).
 * device_id == -1: host-only context for reference packing, window tables and hits formatting; calitas_search fails. */
int calitas_create(int device_id, calitas_ctx** out);
void calitas_destroy(calitas_ctx* ctx);
const char* calitas_last_error(const calitas_ctx* ctx);   /* ctx may be NULL: last error of calitas_create */
void calitas_free(void* p);

/* Reference ----------------------------------------------------------------------------------------------------- */

/* Replaces ReferenceSequenceIterator + windowIterator's byte[] contigs (SearchReference.scala:39-49): ASCII bases of
 * every contig, in sequence-dictionary order.  Packs to 2 bit/base + an exception-run table (N runs, IUPAC codes) and
 * uploads to HBM.  genome_build = first AS tag of the .dict or "unknown" (ReferenceHit.scala:208).
 * bases[i] == NULL with lengths[i] > 0: contig i is ABSENT -- its name and length count (windowIterator's sequence, the coordinates
 * and the sort order are those of the whole dictionary) but its bases are not held here: what a process of a multi-GPU job does
 * with the contigs its window range (calitas_params_t.first_window / n_windows) does not touch.  A search that would need an absent
 * contig fails with CALITAS_EINVAL. */
int calitas_set_reference(calitas_ctx* ctx, int32_t n_contigs, const char* const* names, const uint64_t* lengths,
                          const uint8_t* const* bases, const char* genome_build);
/* Convenience: read FASTA (+ .dict next to it when present) and call calitas_set_reference. */
int calitas_set_reference_fasta(calitas_ctx* ctx, const char* fasta_path);
/* Persistent packed index: the 2-bit codes, exception mask, run table and tile table of the resident reference, so a later
 * run can skip FASTA parsing and packing (replaces the per-run ReferenceSequenceFile scan of SearchReference.scala:34-49). */
int calitas_save_index(const calitas_ctx* ctx, const char* path);
int calitas_load_index(calitas_ctx* ctx, const char* path);
int calitas_reference_info(const calitas_ctx* ctx, int32_t* n_contigs, uint64_t* total_bases, uint64_t* packed_bytes);
int calitas_contig_name(const calitas_ctx* ctx, int32_t contig_index, const char** name, uint64_t* length);
/* genome_build column: first AS tag of the sequence dictionary, or "unknown" (ReferenceHit.scala:208). */
const char* calitas_genome_build(const calitas_ctx* ctx);
/* Upper-cased bases [start, start+len) of a contig re-derived from the packed form (what fetchBases sees after
 * toUpperCase, ReferenceHit.scala:261-266); out must hold len bytes. */
int calitas_fetch_bases(const calitas_ctx* ctx, int32_t contig_index, uint64_t start, uint32_t len, char* out);

/* Every environment switch the library reads (calitas_amd/csrc/tuning.hpp): "NAME <values> what it does" per line; the product has one
 * behaviour, the switches are for measurements, for tests that force a fallback path and for a caller that shares the card.  The string
 * lives as long as the library. */
const char* calitas_switches(void);

/* calitas_free of a text of gigabytes, and the tables calitas_search_variants builds on the way (millions of small heap blocks), are
 * handed back on a thread of the library's own: the calls return at once (CALITAS_FREE_NOW=1: before they return).  This waits until
 * that thread has nothing left to do -- for a caller who measures, or who wants the memory back before going on.  No reference
 * counterpart (the JVM's collector plays this part there). */
void calitas_reap_wait(void);

/* Test hook without a reference counterpart: the host half of the compact rows calitas_search_hits_batch moves over PCIe (round 4).  A
 * row of hits.txt is head | chromosome \t middle | tail with head (guide_id, unpadded_guide_sequence, genome_build) and tail (aligner ..
 * time_stamp, ReferenceHit.scala:99-132) the same for every row of a call; the device writes `chromosome \t middle \n` per row and
 * the library puts head and tail back on its worker pool.  compact[0..n): `rows` such rows; out: room for n + rows * (strlen(head) +
 * strlen(tail) - 1) bytes (out_capacity is checked).  *written receives the bytes written.  CALITAS_EINVAL when the text does not hold
 * exactly `rows` newline-terminated rows or out is too small.  ctx may be a host-only context (its worker pool is used). */
int calitas_expand_rows(calitas_ctx* ctx, const char* compact, uint64_t n, uint64_t rows, const char* head, const char* tail, char* out,
                        uint64_t out_capacity, uint64_t* written);

/* Window table of windowIterator (SearchReference.scala:39-71) after the length filter (SearchReference.scala:536):
 * rows of (contig_index, start0, end0) with N-trimmed 0-based half-open bounds.  Caller frees *out. */
int calitas_window_table(const calitas_ctx* ctx, int32_t window_size, int32_t step, int32_t min_length, int32_t chrom_index,
                         int32_t** out, uint64_t* n_windows);

/* Search -------------------------------------------------------------------------------------------------------- */

/* SearchReference.execute's alignment phase (SearchReference.scala:527-561) for n_guides guides in one pass over the
 * resident reference.  *out receives, guide by guide and window by window in windowIterator order, the alignments
 * SequentialGuideAligner.align returns for each window (after its per-window overlap filter). */
int calitas_search(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                   calitas_aln_t** out, uint64_t* n_out);
int calitas_get_timing(const calitas_ctx* ctx, calitas_timing_t* out);

/* The candidate filter of calitas_search on its own (the stage that decides which end columns the aligner kernel looks at):
 * every (16-base word of the packed reference, strand, guide) that holds at least one end column whose glocal bottom-row score
 * reaches minGuideScore for a seamless scan of the contig, i.e. fgbio's enumeration rule (SequentialGuideAligner.scala:261-299)
 * before windowing.  Out: n records of two uint32 each, sorted: [0] packed position / 16, [1] bits 0-15 column mask,
 * bit 16 strand pass (0 = target as is, 1 = reverse-complemented), bits 17-23 guide.  Exposed so that the two scan kernels can
 * be held against each other and against a plain dynamic-programming count in the tests; calitas_free releases *records. */
int calitas_scan_candidates(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                            uint32_t** records, uint64_t* n_records);
/* Test hook only: the same record set from round 1's column-wise scan kernel (one DP column per 32-bit word, the text walked base by
 * base).  No search entry point uses that kernel; it is kept as an independent second implementation of the filter for the tests. */
int calitas_scan_candidates_columnwise(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                                       uint32_t** records, uint64_t* n_records);
/* Scan tiles of the resident reference: how many there are (contig tiles, padding excluded), how many of them the scan skips because
 * they hold nothing but upper-case N (no window can contain any of their bases, SearchReference.scala:58-59), how many carry
 * exception bases (N-run edges, IUPAC codes, contig ends), and the bases per tile. */
int calitas_reference_tiles(const calitas_ctx* ctx, uint64_t* n_tiles, uint64_t* n_dead, uint64_t* n_masked, uint64_t* tile_bases);
/* Packed position (the unit of calitas_scan_candidates) of base 0 of contig i. */
int calitas_contig_packed_base(const calitas_ctx* ctx, int32_t i, uint64_t* gbase);

/* SearchReference.execute for one guide on the reference genome, end to end (SearchReference.scala:527-564 and 641-648):
 * calitas_search followed by removeOverlaps, ReferenceHit.sort and the 34-column rows, with the alignments never leaving
 * the device -- the per-window filter, removeOverlaps, both sorts and the row text are produced by kernels and only the
 * finished hits.txt text (header + rows, NUL-terminated, *tsv_bytes without the NUL) is copied back.  Byte-identical to
 * calitas_search + calitas_hits_tsv; when a device stage cannot represent a search the host
 * implementation of that stage finishes the call, and when the buffers of a pass over the whole reference do not fit the
 * device (very permissive searches) the call runs one pass per contig instead (CALITAS_ENOMEM only if a single contig does
 * not fit).  aligner_version / time_stamp as in calitas_hits_tsv. */
int calitas_search_hits(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                        const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows);

/* calitas_search_hits for a batch of guides (BASELINE config 4: 96 guides against one reference) -- the loop a caller would
 * write around SearchReference, with the guides flowing through the stages as a pipeline: while guide g's alignments are
 * filtered, de-duplicated, turned into rows and copied back, guide g+1 is already being scanned.  tsv[i] / tsv_bytes[i] /
 * n_rows[i] receive what calitas_search_hits returns for guides[i] (byte-identical); each text is freed with calitas_free.
 * All guides must have the same length (one window tiling).  guide_ids may be NULL. */
int calitas_search_hits_batch(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const char* const* guide_ids,
                              const calitas_params_t* params, const char* aligner_version, const char* time_stamp, char** tsv,
                              uint64_t* tsv_bytes, uint64_t* n_rows);

/* SequentialGuideAligner.align on explicit (guide, target) pairs -- the per-task call of PairwiseAlignSequences
 * (PairwiseAlignSequences.scala:64 -> alignBest, SequentialGuideAligner.scala:333-345) and AlignToReference
 * (AlignToReference.scala:114-135 -> alignToRef / alignToRefBest, SequentialGuideAligner.scala:359-418).  Task t aligns
 * guides[t] to targets[t] (target_lengths[t] bytes, used as given: no N trimming, case-insensitive scoring) with
 * targetOffset = target_offsets[t] (NULL = 0).  params->max_guide_diffs < 0 selects alignBest's limits per task
 * (d = protospacer length, p = longest PAM, D = d + g + p, maxOverlap = 0); otherwise the limits in params apply to every
 * task.  *out receives, task by task, what align() returns (forward-strand list then reverse-strand list after the
 * overlap filter); contig_index / guide_index of each record hold the task index; counts[t] = alignments of task t. */
int calitas_align_windows(calitas_ctx* ctx, int32_t n_tasks, const calitas_guide_t* guides, const uint8_t* const* targets,
                          const uint32_t* target_lengths, const int32_t* target_offsets, const calitas_params_t* params,
                          calitas_aln_t** out, uint64_t* n_out, uint32_t** counts);
/* Padded strings of a record returned by calitas_align_windows, from the caller's own target bytes (case preserved). */
int calitas_padded_strings_target(const calitas_guide_t* guide, const calitas_aln_t* aln, const uint8_t* target, uint32_t target_length,
                                  int32_t target_offset, char* padded_guide, char* padded_alignment, char* padded_target);

/* Host-side stages, usable on a host-only context ------------------------------------------------------------------ */

/* The per-window greedy filter of SequentialGuideAligner.align (SequentialGuideAligner.scala:315-320) on the
 * alignments of ONE window given in enumeration order (forward-strand list then reverse-strand list): stable sort by
 * (score desc, gap bases asc), keep if edits <= max_total_diffs and no kept same-strand alignment overlaps by more
 * than max_overlap.  keep[i] is set to 1 for survivors; order[] receives the indices of survivors in output order. */
int calitas_window_filter(const calitas_aln_t* alns, int32_t n, int32_t max_total_diffs, int32_t max_overlap,
                          int32_t* order, int32_t* n_kept);

/* removeOverlaps + ReferenceHit.sort + the 34-column rows (SearchReference.scala:641-648, ReferenceHit.scala:99-287)
 * for the alignments of one guide.  Returns the full hits.txt text (header + rows) in *tsv. */
int calitas_hits_tsv(const calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                     const calitas_aln_t* alns, uint64_t n_alns, const char* aligner_version, const char* time_stamp,
                     char** tsv, uint64_t* n_rows);

/* A hit row built by the caller -- the variant branch of SearchReference.execute (SearchReference.scala:570-630) builds its
 * ReferenceHits from variant windows on the host.  calitas_hits_tsv_ext lets such hits take part in removeOverlaps (grouped by
 * chromosome, strand and variant_description, SearchReference.scala:656) and in the final sort next to the reference hits;
 * `row` (34 tab-separated columns, no newline) is emitted verbatim. */
typedef struct {
  int32_t contig_index;
  int32_t coordinate_start;     /* ReferenceHit.coordinate_start */
  int32_t end;                  /* ReferenceHit.end = coordinate_start + cigar.lengthOnTarget - 1 (ReferenceHit.scala:135-138) */
  int32_t score;
  int8_t strand;
  const char* variant_description;   /* NULL or "" = none */
  const char* row;
} calitas_ext_hit_t;
int calitas_hits_tsv_ext(const calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                         const calitas_aln_t* alns, uint64_t n_alns, const calitas_ext_hit_t* ext, uint64_t n_ext,
                         const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* n_rows);

/* calitas_search_hits with the text delivered into a buffer of the caller (dst_capacity bytes, NUL included) instead of a block of
 * the library: what a multi-process job uses to have every process's piece of hits.txt land in one shared mapping without a second
 * copy (bench.py --gpus N).  The buffer should be page-locked -- calitas_pin_host does that -- or the copy from the device falls back
 * to the runtime's staged path; a block from calitas_alloc_host is what the copy engines like best.  A search that does not fit the
 * device in one pass runs one pass per contig, as calitas_search_hits does, with every contig's rows crossing the bus straight to
 * their place in the buffer.  CALITAS_EINVAL when the buffer is too small. */
int calitas_search_hits_into(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                             const char* aligner_version, const char* time_stamp, char* dst, uint64_t dst_capacity, uint64_t* tsv_bytes,
                             uint64_t* n_rows);
/* Page-locks / releases host memory the caller owns (hipHostRegister): destinations of calitas_search_hits_into. */
int calitas_pin_host(calitas_ctx* ctx, void* p, uint64_t bytes);
int calitas_unpin_host(calitas_ctx* ctx, void* p);
/* calitas_free parks the blocks it is given -- up to 24 GB of texts and arrays, and one pageable block of a gigabyte or more (the text
 * of a whole-genome search with variants: writing 22 GB into pages nobody has touched, and handing them back, is most of what such a
 * call costs on the host) -- for the next call of the kind to take as they are.  calitas_release_parked() returns all of it to the
 * system; CALITAS_FREE_NOW=1 parks nothing big. */
void calitas_release_parked(void);

/* A page-locked block of the runtime's own (hipHostMalloc), handed back with calitas_free: the destination the *_into calls like best --
 * the GPU's copy engines write into such a block directly (a range that calitas_pin_host merely locked is served by the runtime's
 * copy kernels instead).  NULL when the block cannot be had.  Meant to be reused from call to call: page-locking is not cheap. */
void* calitas_alloc_host(uint64_t bytes);

/* calitas_search_hits with the text handed to a callback instead of returned in one block (Metric.writer's output stream,
 * SearchReference.scala:646-648): `sink` receives consecutive pieces of hits.txt -- the whole text in one piece when the search fits
 * one call, the header and then every contig's rows (in portions of at most 1 GB) when it runs one pass per contig; a piece is
 * valid only during the call.  A non-zero return of the sink aborts the search with CALITAS_EIO.  For searches whose hits.txt has
 * tens of gigabytes this avoids holding it in memory. */
typedef int (*calitas_text_sink_t)(const char* piece, uint64_t bytes, void* user);
int calitas_search_hits_stream(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                               const char* aligner_version, const char* time_stamp, calitas_text_sink_t sink, void* user,
                               uint64_t* tsv_bytes, uint64_t* n_rows);

/* SearchReference.execute with --variants (SearchReference.scala:570-648), end to end: the reference hits of calitas_search plus
 * the hits of every variant window -- variantWindowIterator / nextChunk / reChunk / alleleCombos / buildVariantWindow
 * (SearchReference.scala:217-399) on the host, the windows aligned on the GPU in batches (the calitas_align_windows path), coordinates
 * lifted back with refOffsetAtBaseOffset (SearchReference.scala:133-156), window-local flanks (SearchReference.scala:598-613), the
 * variant columns of ReferenceHit.Builder.build (ReferenceHit.scala:211-233) -- merged by removeOverlaps / ReferenceHit.sort.
 * vcf_path: plain or gzip VCF, records in reference order (CHROM POS ID REF ALT FILTER INFO with AF / END); chrom: NULL or the
 * --chrom filter (params->chrom_index must name the same contig); vcf_id: the "name:md5" string of ReferenceHit.scala:175-183,
 * or NULL to have it computed from the file; params->max_variants = --max-variants.  *n_windows (optional) receives the number of variant windows. */
int calitas_search_variants(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                            const char* vcf_path, const char* chrom, const char* vcf_id, const char* aligner_version,
                            const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows, uint64_t* n_windows);
/* calitas_search_variants with the text delivered into a buffer of the caller (dst_capacity bytes, NUL included), as
 * calitas_search_hits_into does for a search without variants.  A hits.txt with variants at whole-genome size is tens of gigabytes
 * (21.8 GB for BASELINE config 5's shape): into a block of the library's it crosses the bus into a bounce buffer and is copied from there
 * into pages nobody has touched yet; into a page-locked buffer (calitas_pin_host, once, reused from call to call) every contig's rows
 * cross the bus straight to their place.  CALITAS_EINVAL when the buffer is too small (the message says how far the text got). */
int calitas_search_variants_into(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                                 const char* vcf_path, const char* chrom, const char* vcf_id, const char* aligner_version,
                                 const char* time_stamp, char* dst, uint64_t dst_capacity, uint64_t* tsv_bytes, uint64_t* n_rows,
                                 uint64_t* n_windows);

/* What calitas_search_variants knows about a VCF, on its own.
 * calitas_vcf_identifier: the "name:md5" string of ReferenceHit.scala:175-183 (*id: a block for calitas_free) -- a caller that
 * searches many guides against one VCF computes it once and passes it as vcf_id (0.14 s per 127 MB otherwise, per call).
 * calitas_vcf_records: the records as the search reads them -- plain or gzip, parsed in waves on the context's worker pool, the --chrom
 * filter applied -- one line each: CHROM, POS, end (INFO END, else POS + len(REF) - 1: fgbio Variant.end), ID ("" for "."), REF, the ALT
 * alleles and the AF values (as %.9g of the float the search keeps) separated by commas, tab-separated.  Works on a host-only context
 * (calitas_create(-1)); there so that the reader can be held against an independent one (tests/test_variants_host.py). */
int calitas_vcf_identifier(calitas_ctx* ctx, const char* vcf_path, char** id);
int calitas_vcf_records(calitas_ctx* ctx, const char* vcf_path, const char* chrom, char** text, uint64_t* n_records);

/* Padded strings of one alignment (Alignment.paddedString as used at SequentialGuideAligner.scala:511, plus the
 * reverse-complement handling of 5' PAM guides): each buffer must hold CALITAS_MAX_OPS+1 bytes. */
int calitas_padded_strings(const calitas_ctx* ctx, const calitas_guide_t* guide, const calitas_aln_t* aln, char* padded_guide,
                           char* padded_alignment, char* padded_target);

const char* calitas_version(void);

#ifdef __cplusplus
}
#endif
#endif
