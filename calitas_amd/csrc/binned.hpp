// binned.hpp -- the tail of SearchReference.execute for one guide, fused per reference bin (binned.hip): per-window filter
// (SequentialGuideAligner.scala:315-320), GuideAlignment coordinates, ReferenceHit.sort, removeOverlaps (SearchReference.scala:653-675)
// and the hits.txt rows (ReferenceHit.scala:210-254) in TWO launches behind trace_kernel, with no host round trip between them.
//
// The general kernels (select.hip, hits.hip) group the raw alignments by window with a counting sort over the whole window table and
// order the accepted ones globally: nine + seven + five dependent launches and three host round trips per contig range, ~0.3 ms of
// latency that does not shrink with the input.  Here trace_kernel drops every alignment into the *bin* (a fixed
// power-of-two stretch of one contig, >= 2 windows) its window starts in, and one wave per bin does everything else for the hits whose
// coordinate_start lies in its bin, from its own bin and the edges of the two neighbouring ones -- see binned.hip for why that is exact
// and when a bin declines (the call then finishes on the general kernels, from the same raw alignments).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "common.hpp"
#include "hits.hpp"
#include "kernels.hpp"
#include "mailbox.hpp"

namespace calitas {

struct BinnedWork;   // per-lane device scratch

constexpr uint32_t BIN_CAP = 64;            // raw alignments listed per bin (a bin with more is "crowded": the call declines)
constexpr uint32_t BIN_ROWS = 32;           // rows (kept hits) per bin
constexpr unsigned long long BIN_HOST_TEXT = 128u << 10;   // a text up to this size is written into page-locked host memory by the rows kernel itself

constexpr uint32_t BIN_FLAG_CROWDED = 1;    // a bin, its context, its accepted alignments or its rows exceed what one wave holds
constexpr uint32_t BIN_FLAG_HALO = 2;       // removeOverlaps: a hit of the bin hangs on a cluster that starts left of the known context
constexpr uint32_t BIN_FLAG_ROW = 4;        // a row has more padded columns (or a longer span) than the row builder lays out
constexpr uint32_t BIN_FLAG_RANGE = 8;      // a coordinate outside the key's range
constexpr uint32_t BIN_FLAG_TEXT = 16;      // the text buffer is too small: only the rows kernel has to run again
constexpr uint32_t BIN_FLAG_INTERNAL = 32;  // a row's length differs between the two kernels (a bug, never a property of the input)

// Mailbox words of the post of the rows kernel (after the sequence word).
constexpr int BIN_BOX_COUNTERS = 1;         // 1..8: the eight counters of the lane (scan records, raw alignments, anomalies, ...)
constexpr int BIN_BOX_ROWS = 9, BIN_BOX_BYTES = 10 /* and 11 */, BIN_BOX_FLAGS = 12, BIN_BOX_LATE = 13 /* flags raised while rows are written */,
              BIN_BOX_ACCEPTED = 14, BIN_BOX_COMPLEX = 15 /* statistics: bins that took a whole wave */,
              BIN_BOX_STAMPS = 16 /* .. 21: the device's wall clock (two words each) at the start of align_kernel, of bin_hits_small_kernel and of
                                     the rows kernel -- the lane's kernel times without an event on any dispatch between them */;

// Bases per bin for a window size: the smallest power of two that leaves room for two windows of context on the left and the longest
// hit on the right, at least 8 kb; 0 = this window size is not handled.
int binned_shift(int window_size);

struct BinnedGeometry {      // where the bins of a lane's range lie
  const uint32_t* d_bin_base;   // per contig (n_contigs + 1)
  const uint32_t* d_bin_contig; // per bin: its contig
  int n_contigs;
  uint32_t bin_first, n_bins;   // the range's bins
  uint32_t bin_shift;
};

// Scratch for n_bins bins; the bin counters, chunk sums and flags are cleared on `stream`.  Call at the start of a search, ahead of
// the kernels (nothing of it has to sit between the end of the scan and align_kernel).
hipError_t binned_prepare(BinnedWork** work, uint32_t n_bins, hipStream_t stream);
// The same without the clear: *clear / *clear_bytes (a multiple of 16) is what the caller has to zero ahead of trace_kernel -- the
// lane's setup kernel does it together with the other small inputs of a call (kernels.hpp, LaneSetupArgs).
hipError_t binned_prepare_host(BinnedWork** work, uint32_t n_bins, void** clear, size_t* clear_bytes);
// Where trace_kernel drops the alignments.
void binned_fill_align_args(const BinnedWork* work, const BinnedGeometry& geo, AlignArgs& aa);

struct BinnedParams {
  int window_size, step, max_total_diffs, max_overlap, max_ops;
  // The rows this call owns: hits whose (contig << 32 | coordinate_start) lies in [own_lo, own_hi) -- 0 / ~0 for a whole reference.
  // A window range of calitas_search_hits (calitas_params_t::first_window / n_windows) becomes such a stretch: from the start of its
  // first window to the start of the window behind its last.  The caller's bin range covers the stretch plus one bin on each side
  // (the context of the first and the last owned bin), and the aligner ran on every window those bins' contexts reach.
  unsigned long long own_lo, own_hi;
};

// Queues the two kernels behind trace_kernel.  hits: the lane's HitsWork after hits_prepare / hits_set_names (constant row pieces, contig
// names, the text buffer).  d_counters: the lane's eight counters (posted with the result).  The rows kernel posts rows, bytes and
// flags to `post` when it STARTS (they are final then); the text is complete when the stream is.  ev_*: optional timing events that
// ride on the dispatches (each costs the kernel behind it ~5 us of its start: pass nullptr and take the times from the posted stamps).
hipError_t binned_run(BinnedWork* work, HitsWork** hits, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                      const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters, hipStream_t stream,
                      Mailbox* post, hipEvent_t ev_hits_done, hipEvent_t ev_rows_start, hipEvent_t ev_rows_done, bool with_rows = true);
// The rows kernel by itself, behind a binned_run(..., with_rows = false): host_dst / host_dst_cap (may be null / 0) is page-locked memory
// of the caller the device can address -- the place the text finally goes.  A text of up to host_dst_cap bytes is written there by the
// kernel (46-49 GB/s for rows of ~520 bytes, tools/host_write_bench.hip: the copy engine's rate, without the copy's start-up and the
// wait between the kernel and it); a longer one goes to the device buffer as usual.  The posted byte count tells which.
hipError_t binned_rows(BinnedWork* work, HitsWork** hits, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                       const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters, hipStream_t stream,
                       Mailbox* post, hipEvent_t ev_rows_done, char* host_dst, unsigned long long host_dst_cap);
// After BIN_FLAG_TEXT: the text buffer grown to `bytes`, the rows kernel once more.
hipError_t binned_rerun_rows(BinnedWork* work, HitsWork** hits, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                             const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters,
                             uint64_t bytes, hipStream_t stream, Mailbox* post, hipEvent_t ev_rows_done);
// Where the text is: in the lane's page-locked host buffer when the posted byte count is <= binned_host_cap() (complete when the
// stream is; no copy), else in the device buffer of `hits`.
const char* binned_text(const HitsWork* hits);
const char* binned_host_text(const BinnedWork* work);
unsigned long long binned_host_cap(const BinnedWork* work);
// Milliseconds between two posted stamps (BIN_BOX_STAMPS + 2 * from / to) of the lane's last post.
double binned_stamp_ms(const BinnedWork* work, const Mailbox& box, int from, int to);
void binned_destroy(BinnedWork* work);

}  // namespace calitas
