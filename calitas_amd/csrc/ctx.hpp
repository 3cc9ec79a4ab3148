// ctx.hpp -- private: the context object behind the C ABI and helpers shared by api.cpp / align_windows.cpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include <sched.h>

#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/calitas_hip.h"
#include "common.hpp"
#include "kernels.hpp"
#include "parallel.hpp"
#include "post.hpp"
#include "refpack.hpp"
#include "select.hpp"
#include "hits.hpp"
#include "binned.hpp"
#include "dma.hpp"
#include "mailbox.hpp"

using namespace calitas;

struct calitas_ctx {
  int device = -1;
  std::string err;
  PackedRef ref;
  bool has_ref = false;
  // device state
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};            // [0..3] scan start / scan end / align+trace end / filter end, [4..5] row kernels, [6..7] text copy
  uint32_t* d_codes = nullptr;
  uint2* d_planes = nullptr;        // codes as bit-planes per 32 bases (scan_rows.hip)
  uint32_t* d_mask = nullptr;
  Run* d_runs = nullptr;
  ContigInfo* d_contigs = nullptr;
  TileInfo* d_tiles = nullptr;
  uint32_t* d_tile_list = nullptr;
  uint64_t* d_win_base = nullptr;   // window table for (win_W, win_step)
  int2* d_win = nullptr;
  uint64_t win_cap = 0;
  int win_W = 0, win_step = 0;
  GuideDev* d_guides = nullptr;
  GuideDev* h_guides = nullptr;     // pinned: the source of the upload (an async copy from pageable memory waits for the stream to drain)
  ScanRecord* d_recs = nullptr;
  RawAln* d_raw = nullptr;
  uint32_t* d_counters = nullptr;   // [0] scan records, [1] raw alignments, [2] anomalies, [3] passing candidates (items), [4] candidates
  uint8_t* d_slab = nullptr;        // strips handed from align_kernel to trace_kernel
  uint64_t slab_cap = 0;            // bytes
  uint64_t* d_items = nullptr;      // passing candidates (align_kernel -> trace_kernel)
  uint32_t item_cap = 0;
  uint32_t* h_counters = nullptr;   // pinned
  Mailbox mbox;                     // how the counters reach h_counters between two stages of a search (mailbox.hpp)
  uint32_t rec_cap = 0, raw_cap = 0;
  RawAln* h_raw = nullptr;          // pinned staging for the copy-back
  uint32_t h_raw_cap = 0;
  calitas_timing_t timing{};
  SelectWork* select = nullptr;     // GPU per-window filter scratch
  HitsWork* hits = nullptr;         // GPU removeOverlaps / sort / rows scratch
  uint64_t ref_serial = 0, hits_names_serial = ~0ull;
  // binned tail (binned.hpp): the owner keeps the bins' geometry, every lane its own scratch
  uint32_t* d_bin_base = nullptr;   // per contig: index of its first bin, for bin_shift (owner)
  uint32_t* d_bin_contig = nullptr; // per bin: its contig
  std::vector<uint32_t> bin_base;   // the same on the host
  int bin_shift = 0;                // 0 = not built
  BinnedWork* binned = nullptr;     // lane
  double align_ms_by_stamps = -1;   // lane: >= 0: align_kernel + trace_kernel of the current search ran without an event behind them (binned.hpp, BIN_BOX_STAMPS)
  int rows_ev0 = 4;                 // lane: ev[rows_ev0] .. ev[5] bracket the row stage of the last call
  bool binned_late_check = false;   // lane: the text being copied comes from the binned rows kernel (its late flags are checked after the copy's wait)
  // the last search on this context the binned tail declined (crowded bins, a long repeat): protospacer length, PAMs, minGuideScore.
  // A search at least as permissive goes to the general kernels directly.
  int bin_decl_L = 0, bin_decl_pams = -1, bin_decl_min_score = 0;
  uint64_t bin_decl_guide = 0;      // ... of which guide (a hash of its row sets and PAM masks): a crowded bin is where THIS guide meets a repeat, and a
                                    // batch of 96 guides must not lose the bins for all because one of them did
  HitsWork* hits_alt = nullptr;     // second row-stage scratch of the per-contig passes: contig c+1's rows are built while contig c's text is copied
  uint64_t hits_alt_names_serial = ~0ull;
  // chunked calitas_search_hits: the parent owns the lanes and the stream all scans are queued on
  calitas_ctx* parent = nullptr;    // set in a lane: the context whose reference and window table it uses
  std::vector<calitas_ctx*> lanes;
  // calitas_align_windows: the temporary packed reference of a call's tasks on the device, kept between calls (freeing device memory
  // waits for the whole device, and the variant branch aligns its windows beside the reference passes of the same call)
  struct AlignScratch { void* p[8] = {}; size_t cap[8] = {}; } aw;
  calitas_ctx* side2 = nullptr;     // ... and a second one: two batches of variant windows are aligned side by side
  calitas_ctx* side = nullptr;      // a child context of its own (stream, buffers) for work that runs beside a search of this context: the
                                    // variant windows' alignment while the reference passes of the same call are under way (calitas_side_context)
  struct LaneThreads* lane_threads = nullptr;   // parent: the host threads that drive lanes 1.. (search.cpp)
  hipStream_t scan_stream = nullptr;
  hipStream_t scan_more[3] = {nullptr, nullptr, nullptr};   // a batch's scans take turns on scan_stream and these (CALITAS_BATCH_SCAN_STREAMS): the next scan fills the CUs the one before it leaves as it drains
  hipStream_t copy_stream = nullptr;  // parent: the text copies of all lanes
  hipEvent_t scan_done = nullptr;   // lane: recorded on the parent's scan stream after this lane's scan
  hipEvent_t t_scan0 = nullptr, t_scan1 = nullptr;   // the two events that bracket the last scan kernel (ev[0] / ev[1], or scan_done events)
  hipEvent_t rows_ready = nullptr;  // lane: recorded on its stream after its row kernels
  hipEvent_t inputs_ready = nullptr;  // lane: its scan's inputs (guide constants, cleared counters) are in place (queued on the lane's own stream)
  uint64_t last_text_bytes = 0;
  // the last search that had to run one pass per contig: protospacer length, number of PAMs, minGuideScore (a search at least as
  // permissive goes there directly instead of finding out again)
  int seq_L = 0, seq_pams = -1, seq_min_score = 0;
  double seq_recs_per_tile = 0;     // scan records per live tile of that search (estimate_scan_records): sizes the per-contig passes
  // ... and the last one that fit one pass: a search at most as permissive needs no estimate
  int fit_L = 0, fit_pams = -1, fit_min_score = 0;
  std::mutex host_mu;               // host stages of concurrent lanes take turns on the worker pool
  DmaCopier dma;                    // parent: SDMA copies of the finished text (dma.hpp)
  bool dma_tried = false;
  WorkerPool* pool = nullptr;
  ~calitas_ctx() { delete pool; }
};


int calitas_fail(calitas_ctx* ctx, int code, const std::string& msg);
void* calitas_out_alloc(size_t size);
void* calitas_out_shrink(void* p, size_t size);  // gives back what a block has far too much of (api.cpp)
void* calitas_out_take_big(size_t min_bytes);   // the parked pageable block of >= 1 GB, if there is one with that much room (api.cpp)
void* calitas_out_alloc_pinned(size_t size);   // page-locked: the destination of the text copy-back
void* calitas_out_grow(void* p, size_t keep, size_t size);   // pageable block grown in place (realloc); p may be NULL
// search.cpp
int calitas_search_impl(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                        calitas_aln_t** out, uint64_t* n_out);
int calitas_search_hits_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                             const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows);
int calitas_search_hits_into_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                  const char* aligner_version, const char* time_stamp, char* dst, uint64_t dst_capacity, uint64_t* tsv_bytes,
                                  uint64_t* n_rows);
int calitas_search_hits_stream_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                    const char* aligner_version, const char* time_stamp, calitas_text_sink_t sink, void* user,
                                    uint64_t* tsv_bytes, uint64_t* n_rows);
// hits of the caller's own (one HitsExt per contig, hits.hpp) brought into every contig's device row stage; *declined (with CALITAS_ESTATE):
// a stage left the device path and nothing is returned -- the caller merges on the host
int calitas_search_hits_ext_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                 const char* aligner_version, const char* time_stamp, const HitsExtSource& source, char** tsv,
                                 uint64_t* tsv_bytes, uint64_t* n_rows, bool* declined, char* user_dst = nullptr, uint64_t user_cap = 0);
int calitas_side_context(calitas_ctx* ctx, calitas_ctx** side, int which = 0);
// Work nobody waits for -- handing memory back -- on a thread of the library's own, in the order it was given (joined when the library
// is unloaded).  A call that built millions of small objects, or a caller that frees a text of tens of gigabytes, returns at once.
void calitas_reap_later(std::function<void()> job);
int calitas_search_hits_batch_impl(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const char* const* guide_ids,
                                   const calitas_params_t* params, const char* aligner_version, const char* time_stamp, char** tsv,
                                   uint64_t* tsv_bytes, uint64_t* n_rows);
int calitas_scan_candidates_impl(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                                 uint32_t** records, uint64_t* n_records, bool columnwise = false);
void calitas_destroy_lanes(calitas_ctx* ctx);
void calitas_default_version_and_stamp(const char* aligner_version, const char* time_stamp, std::string& version, std::string& stamp);
// Per-guide device constants for limits (d, p) and costs; returns an error text or "".
std::string build_guide_dev(const GuideHost& gh, const calitas_params_t& p, const Scores& sc, int max_guide_diffs, int max_pam_mismatches,
                            GuideDev& gd);
int ensure_buffers(calitas_ctx* ctx, uint32_t rec_cap, uint32_t raw_cap, uint64_t slab_per_rec, uint32_t item_cap);

// Host waits on the critical path poll instead of blocking: a call has four of them per lane and a blocking wait adds tens of
// microseconds of wake-up latency each.
// (After a few thousand polls the thread yields between polls, so an oversubscribed host is not starved by waiting lanes.)
// (Spinning for ~50 us, then yielding the core, then sleeping between polls: mailbox.hpp, Backoff.)
template <typename Q>
inline hipError_t calitas_poll(Q query) {
  hipError_t e;
  Backoff wait;
  while ((e = query()) == hipErrorNotReady) wait.pause();
  return e;
}
inline hipError_t calitas_spin_sync(hipStream_t s) { return calitas_poll([s] { return hipStreamQuery(s); }); }
inline hipError_t calitas_spin_sync(hipEvent_t ev) { return calitas_poll([ev] { return hipEventQuery(ev); }); }

#define HIP_TRY(ctx, call)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) { if (e_ == hipErrorOutOfMemory) (void)hipGetLastError();                    \
      return calitas_fail(ctx, e_ == hipErrorOutOfMemory ? CALITAS_ENOMEM : CALITAS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
  } while (0)
