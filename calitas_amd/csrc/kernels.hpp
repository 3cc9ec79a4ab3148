// kernels.hpp -- launch interface of kernels.hip (device pointers only).
#pragma once
#include "mailbox.hpp"
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "common.hpp"

namespace calitas {

struct ScanArgs {
  const uint32_t* codes;
  const uint2* planes;     // the same 2-bit codes as two bit-planes per 32 bases (.x = low bits, .y = high bits): what scan_rows_kernel streams
  const uint32_t* mask;
  const TileInfo* tiles;
  const GuideDev* guides;
  ScanRecord* recs;
  uint32_t* rec_count;
  uint32_t rec_capacity;
  int32_t n_guides;
  int32_t chrom_index;
  uint32_t tile_offset;    // first tile of this launch (a chunked search scans one contig range per launch)
  uint32_t tile_stride;    // 1; > 1 = every tile_stride-th tile only (the record-count estimate of a search that may not fit the device)
};

struct AlignArgs {
  const uint32_t* codes;
  const uint32_t* mask;
  const Run* runs;
  int64_t n_runs;
  const ContigInfo* contigs;
  const TileInfo* tiles;
  const uint64_t* win_base;  // per contig: index of its window 0 in win[] (n_contigs + 1 entries)
  const int2* win;           // N-trimmed window bounds for the current (window size, step)
  const GuideDev* guides;
  const ScanRecord* recs;
  const uint32_t* rec_count;
  RawAln* out;
  uint32_t* out_count;
  uint32_t* anomalies;
  uint32_t* trace_done;     // trace_kernel: workgroups finished (zero at launch); the last one posts the counters to the mailbox
  uint8_t* slab;            // one slab per aligner job (SlabHeader): rec_capacity x slots_per_rec of them
  uint32_t* job_count;      // jobs numbered by expand_kernel (zero at launch)
  uint32_t* cand_count;     // statistics only
  uint64_t* items;          // passing candidates, (slab index << 4 | candidate slot): align_kernel appends, trace_kernel consumes
  uint32_t* item_count;
  uint32_t item_capacity;
  uint32_t slab_bytes;      // size of one slab
  uint32_t slots_per_rec;   // windows a 16-base record can fall into
  uint32_t rec_capacity;
  uint32_t out_capacity;
  uint64_t gw_lo, gw_hi;   // global window indices [gw_lo, gw_hi) this call covers (calitas_params_t::first_window / n_windows)
  uint32_t tile_words;     // code words per scan tile
  // Reference bins (binned.hip): trace_kernel also drops every alignment into the bin its window starts in, so that the stages behind
  // it can work bin by bin without a global grouping step.  bin_idx == nullptr: off.
  uint32_t* bin_idx;       // (n_bins of this lane's range) x bin_cap indices into out[]
  uint32_t* bin_count;     // alignments per bin, zero at launch; may exceed bin_cap (the surplus is not stored: the bin is "crowded")
  const uint32_t* bin_base;  // per contig: index of its first bin
  uint32_t bin_first;      // first bin of this lane's range (bins / bin_count are indexed relative to it)
  uint32_t bin_n;          // bins of the range (an alignment whose window starts outside is not listed: cannot happen, checked)
  uint32_t bin_shift;      // log2 of the bases per bin
  uint32_t bin_cap;
  int32_t low_prio;                 // 1: expand / align / trace at the scan's wave priority (a tail that runs beside the next range's scan and ends no call)
  int32_t pack16;                   // 1: every guide has the same protospacer length (<= 20) and the cells fit sixteen bits: align_pk_kernel
  int32_t max_guide_len;            // longest protospacer among the guides (0 = unknown): align_kernel packs three jobs per wave up to 20
  unsigned long long* stamps;       // (binned tail) stamps[0] = the device's wall clock when align_kernel starts; may be null
  SearchDev sp;
};

// The small inputs of one lane's search in ONE launch: the guide's constants, the lane's eight counters cleared, the row stage's three
// counts cleared, the constant row strings, the bins' scratch cleared.  As separate stream commands (two uploads, three or four fills)
// they took ~4.5 us each on the device and as long again on the host: 28 of the 140 us of an E. coli-sized call.  Everything travels
// in the kernel's argument block (one guide, row strings up to LANE_SETUP_BLOB bytes); callers fall back to the commands otherwise.
constexpr uint32_t LANE_SETUP_BLOB = 1536;
struct LaneSetupArgs {
  alignas(16) GuideDev guide;      // (16-byte aligned: the kernel reads its argument block in 16-byte pieces -- it lives in host memory)
  GuideDev* d_guides;              // (both null: the row stage's part only)
  uint32_t* d_counters;            // 8 words
  uint64_t* d_row_counts;          // 3 x 64 bit, or null
  char* d_blob;                    // or null
  uint4* clear;                    // or null; clear_bytes is a multiple of 16
  uint32_t clear_bytes;
  uint32_t blob_bytes;             // d_blob has room for this rounded up to 16
  alignas(16) uint8_t blob[LANE_SETUP_BLOB];
};
hipError_t launch_lane_setup(const LaneSetupArgs& a, hipStream_t stream);

hipError_t launch_scan(const ScanArgs& a, int chunk, uint32_t n_tiles, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// scan_rows.hip: the row-wise scan (default) and the one-off conversion codes[] -> planes[] at upload time
hipError_t launch_scan_rows(const ScanArgs& a, int chunk, int warm_words, uint32_t n_tiles, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
hipError_t launch_planes(const uint32_t* codes, uint2* planes, uint64_t n32, hipStream_t stream);
// expand_kernel (scan records -> aligner jobs, their slabs' heads written) + align_kernel
hipError_t launch_align(const AlignArgs& a, uint32_t n_blocks, hipStream_t stream);
// post: the kernel's last workgroup publishes the eight counters at a.rec_count to that mailbox (mailbox.hpp) -- wait for it with mailbox_wait
hipError_t launch_trace(const AlignArgs& a, uint32_t n_blocks, hipStream_t stream, hipEvent_t stop = nullptr, Mailbox* post = nullptr);
hipError_t launch_window_table(const Run* runs, int64_t n_runs, const ContigInfo* contigs, const uint64_t* win_base, int n_contigs,
                               uint64_t n_windows, int W, int step, int2* out, hipStream_t stream);
hipError_t launch_dpp_selftest(int* out, hipStream_t stream);

}  // namespace calitas
