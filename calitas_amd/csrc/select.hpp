// select.hpp -- GPU implementation of the per-window filter stage (see select.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "common.hpp"

namespace calitas {

struct SelectWork;   // device scratch, grown on demand and reused across searches

// True when the key layout of the sort can represent this search (contigs < 2^18, windows per contig < 2^22, window < 8192).
bool select_supported(uint64_t n_contigs, uint64_t max_windows_per_contig, int window_size, int n_guides);

// Sorts, filters and compacts d_raw[0..n_raw).  On return (stream-ordered) *d_final holds the accepted alignments in
// (guide, contig, window, retval) order and (*d_counts)[0] their number; (*d_counts)[1] != 0 means a window exceeded the
// kernel's group limit and the result must not be used.
hipError_t select_run(SelectWork** work, const RawAln* d_raw, uint32_t n_raw, const GuideDev* d_guides, const uint64_t* d_win_base,
                      const int2* d_win, int max_total_diffs, int max_overlap, hipStream_t stream, const RawAln** d_final,
                      const uint32_t** d_counts);
void select_destroy(SelectWork* work);

}  // namespace calitas
