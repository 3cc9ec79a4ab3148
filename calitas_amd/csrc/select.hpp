// select.hpp -- GPU implementation of the per-window filter stage (see select.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "common.hpp"
#include "mailbox.hpp"

namespace calitas {

struct SelectWork;   // device scratch, grown on demand and reused across searches

// bit of (*d_counts)[1]: the per-window counters were not zero when the stage started (an internal error, never a property of the
// input): the caller fails the search instead of falling back to the host filter
constexpr uint32_t SELECT_FLAG_INTERNAL = 0x80000000u;
// bit of (*d_counts)[1]: the one-workgroup version of the stage (small inputs) met a window it does not filter: call select_run again
// with general = true
constexpr uint32_t SELECT_FLAG_RETRY = 0x40000000u;

// True when the global window index (guide x windows) fits the counters and an end column fits the enumeration key.
bool select_supported(uint64_t windows_per_guide, int window_size, int n_guides);

// Groups, filters and compacts d_raw[0..n_raw).  [window_lo, window_lo + windows_per_guide) = the entries of the device window
// table the alignments can fall into (all of it, or the contig range of a lane).  On return
// (stream-ordered) *d_final holds the accepted alignments in (guide, contig, window, retval) order and (*d_counts)[0] their
// number; (*d_counts)[1] != 0 means a window exceeded the kernels' limits and the result must not be used.  Call select_done
// once the stream has been synchronised without error (it certifies that the per-window counters are back to zero).
// post: the last kernel also publishes (*d_counts)[0..3) to that mailbox -- wait for it with mailbox_wait.
hipError_t select_run(SelectWork** work, const RawAln* d_raw, uint32_t n_raw, const GuideDev* d_guides, const uint64_t* d_win_base,
                      const int2* d_win, uint64_t window_lo, uint64_t windows_per_guide, int n_guides, int max_total_diffs, int max_overlap,
                      hipStream_t stream, const RawAln** d_final, const uint32_t** d_counts, Mailbox* post = nullptr, bool general = false);
// The same stage queued right behind trace_kernel, before the host knows the call's counters (d_counters: the eight words of
// AlignArgs::rec_count .. ; the capacities are what the host would check them against): the one-workgroup kernel reads the number of
// alignments there and posts the eight counters (mailbox words 1..8) together with its counts (words 9..11).  SELECT_FLAG_RETRY in
// word 10: it did nothing -- an overflow, an anomaly, or more alignments than it takes -- and the caller goes on as if it had not run.
hipError_t select_run_speculative(SelectWork** work, const RawAln* d_raw, const uint32_t* d_counters, uint32_t rec_cap, uint32_t raw_cap,
                                  uint32_t item_cap, const GuideDev* d_guides, const uint64_t* d_win_base, const int2* d_win, uint64_t window_lo,
                                  uint64_t windows_per_guide, int max_total_diffs, int max_overlap, hipStream_t stream,
                                  const RawAln** d_final, Mailbox* post);
void select_done(SelectWork* work);
void select_destroy(SelectWork* work);

}  // namespace calitas
