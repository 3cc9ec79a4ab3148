// hits.hpp -- device implementation of SearchReference's tail for the reference-genome branch: removeOverlaps
// (SearchReference.scala:653-675), ReferenceHit.sort (ReferenceHit.scala:284) and the 34-column rows (ReferenceHit.scala:210-254).
// See hits.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <functional>
#include <string>
#include <vector>

#include "common.hpp"
#include "post.hpp"

namespace calitas {

struct HitsWork;   // device scratch, grown on demand and reused

struct HitsRef {   // the resident reference
  const uint32_t* codes;
  const uint32_t* mask;
  const Run* runs;
  int64_t n_runs;
  const ContigInfo* contigs;
  int n_contigs;
};

struct HitsResult {
  uint32_t flags;        // != 0: the device path declined (see HITS_FLAG_*); nothing else is valid
  uint32_t n_rows;
  uint64_t text_bytes;   // bytes of row text at d_text
  const char* d_text;
  // HitsExtRows::fill_on_host: per entry of the HitsExt, the offset of its row's hole in the text, ~0 for an entry that was not kept.
  // Page-locked host memory of the work's; filled by a copy queued on the stream behind the kernels (valid once the stream has passed
  // it -- the caller waits for the text anyway) and until the next hits_run on this work.
  const uint64_t* ext_place;
};

constexpr uint32_t HITS_FLAG_SCORE_RANGE = 1;   // a score fell outside the sort key's range
constexpr uint32_t HITS_FLAG_CLUSTER = 2;       // an overlap cluster is longer than one lane should walk

constexpr uint32_t HITS_FLAG_ROW = 4;           // a row has more padded columns than max_ops allows
constexpr uint32_t HITS_FLAG_INTERNAL = 16;     // a row's length differs between len_kernel and rows_kernel (a bug, never a property of the input)
constexpr uint32_t HITS_FLAG_HALO = 8;          // HitsOwn: an owned hit hangs on a removeOverlaps cluster that starts where not every hit is known

// Whether the sort keys can represent this search at all.
bool hits_supported(uint64_t n_contigs, int max_overlap, int score_lo, int score_hi);

// Contig names for the chromosome column; call again after the reference changes.
hipError_t hits_set_names(HitsWork** work, const std::vector<std::string>& names, hipStream_t stream);

// The call's constant row pieces and cleared counters, queued on `stream`: call it at the start of a search_hits call, ahead of the
// search kernels; hits_run does it itself otherwise.
hipError_t hits_prepare(HitsWork** work, const RowStrings& strings, hipStream_t stream);
// The same without the two stream commands: the caller clears d_counts[0..3) and brings blob[0..blob_bytes) to d_blob itself -- the
// lane's setup kernel does it together with the other small inputs of a call (kernels.hpp, LaneSetupArgs).  `blob` stays valid until
// the next hits_prepare* on this work.
struct HitsSetup { const char* blob; uint32_t blob_bytes; char* d_blob; uint64_t* d_counts; };
hipError_t hits_prepare_host(HitsWork** work, const RowStrings& strings, HitsSetup* out);

// Hits the caller built itself and wants placed among (and, for the reference's removeOverlaps group, walked with) the device's: the
// hits of variant windows (SearchReference.scala:570-630).  They arrive after every hit of the reference windows (SR:622), so among
// equal ReferenceHit.sort keys they follow the device's hits, in the order given here.  An entry without HITS_EXT_PLACED belongs to the
// group "chromosome : strand : no variant_description" (SR:656) and takes part in that group's walk like any reference hit; an entry
// with it was kept by a walk of its own group on the host and is only given its place in the order.  Every entry brings its finished
// row (text + newline): the rows kernel copies the rows of the entries that survive to their place in the text.  Host memory; contig
// is the one contig all of them (and all of d_final's alignments) lie on.
constexpr uint32_t HITS_EXT_MINUS = 1, HITS_EXT_PLACED = 2;
struct HitsExtKey { int32_t coordinate_start, end, score; uint32_t flags; };
struct HitsExtRows {                   // what rows_for hands back: the fields of the same names below
  const uint64_t* row_off = nullptr;
  const char* rows = nullptr;
  uint32_t n_seg = 0;
  const char* const* seg = nullptr;
  const uint64_t* seg_off = nullptr;
  // The rows stay on the host: only row_off is looked at, the rows kernel leaves a hole of the row's length for every kept entry and
  // HitsResult::ext_place says where; the caller writes the rows into the text once it is on the host.  (Round 5: the rows of a variant
  // search's entries carry the VCF's MD5, which takes 0.14 s per 127 MB to compute -- with the rows on the host the first contigs' texts
  // cross the bus meanwhile; and 0.5 GB of rows per call go neither up to the device nor through its rows kernel.)
  bool fill_on_host = false;
};
struct HitsExt {
  int32_t contig = 0;
  uint32_t n = 0;
  const HitsExtKey* keys = nullptr;
  const uint64_t* row_off = nullptr;   // n + 1 offsets into rows
  const char* rows = nullptr;
  // ... or the rows in n_seg consecutive pieces (rows null): piece s holds bytes [seg_off[s], seg_off[s + 1]) of the rows' text -- the
  // caller's workers each wrote a block of rows into a buffer of their own, and joining 1.1 GB of them only to upload them is a copy
  uint32_t n_seg = 0;
  const char* const* seg = nullptr;
  const uint64_t* seg_off = nullptr;   // n_seg + 1
  uint32_t* kept = nullptr;            // out (optional): how many of the entries were kept
  // Rows on demand (round 5): with rows_for set, row_off / rows / seg* above are not looked at.  Nine in ten entries of a variant search
  // repeat a reference hit and lose against it in the device's walk, so the caller is asked for rows only when the walks have decided:
  // hits_run brings one byte per entry to the host (kept[e] != 0: entry e is in the text), calls rows_for(kept, &rows) and takes the n + 1
  // offsets (an entry that was not kept has length 0) and the text from `rows` -- memory of the caller's that stays valid until
  // hits_run returns.  != 0: the caller gave up, hits_run returns hipErrorUnknown.  One more host round trip per call of hits_run.
  std::function<int(const uint8_t* kept, HitsExtRows* rows)> rows_for;
  // HitsExtRows::fill_on_host: called by whoever brought the text to the host (search.cpp, the per-contig passes) with
  // HitsResult::ext_place and the text's address: the rows of the kept entries go to text + place[e].  stays: the text is where it will
  // be when the call returns (a buffer of the caller's), so the callee may do the work later, on a thread of its own that it joins
  // before its call ends (place[] is only valid during fill); otherwise the rows must be in the text when fill returns.
  std::function<int(const uint64_t* place, char* text, bool stays)> fill;
};

// The alignments given to hits_run are those of a window range of a larger job (a process's stretch of a multi-GPU partition plus the
// context around it), and only the rows the stretch OWNS are wanted: hits whose key (contig << 32 | coordinate_start) lies in [lo, hi).
// removeOverlaps is exact for them as long as the cluster an owned hit belongs to starts at a hit from `safe` on -- from there every hit
// that could overlap is among the alignments (left of it, windows outside the range may have held more); a cluster that starts earlier
// and reaches an owned hit raises HITS_FLAG_HALO (a chain of overlapping hits longer than the context: the caller searches whole contigs).
struct HitsOwn { unsigned long long lo = 0, hi = ~0ull, safe = 0; };

// Where a per-contig pass gets its HitsExt from: get(contig, &ext) is called right before the contig's row stage and may block until the
// caller has finished the contig's entries (*ext = nullptr: none).  A return value != 0 means the caller gave up: the search stops.
// compact_rows: the entries' rows are compact (post.hpp, compact_row_strings_keep_build: no guide_id / protospacer, "\n" for a tail), and
// so may the device's own be -- the per-contig texts then cross PCIe compact and are expanded on the host
struct HitsExtSource { std::function<int(int contig, const HitsExt** ext)> get; bool compact_rows = false; };

// d_final[0..n): accepted alignments of ONE guide in calitas_search order (device memory).  Stream-ordered except for one
// synchronisation to learn the text size.  max_ops bounds the padded columns of any alignment of this search (it sizes the
// per-row LDS slots).  window_reach: the number of window steps after which two windows share no base (the final order then comes
// from a count among neighbouring windows), or 0 for the general sort (crowded windows).  On success the rows are at res->d_text in
// final order.
hipError_t hits_run(HitsWork** work, const HitsRef& ref, const RawAln* d_final, uint32_t n, const GuideDev* d_guides,
                    const uint64_t* d_win_base, const int2* d_win, const RowStrings& strings, int max_overlap, int score_hi,
                    int max_ops, uint32_t window_reach, hipStream_t stream, HitsResult* res, const HitsExt* ext = nullptr,
                    const HitsOwn* own = nullptr);
// Flags raised while the rows of the last hits_run were being written (HITS_FLAG_INTERNAL); valid once the stream the rows kernel ran
// on is done.
uint32_t hits_late(const HitsWork* work);
void hits_destroy(HitsWork* work);

}  // namespace calitas
