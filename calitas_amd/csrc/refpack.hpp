// refpack.hpp -- host-side packed reference and the window arithmetic shared by host and device.
#pragma once
#include "common.hpp"

namespace calitas {

struct PackedRef {
  int chunk = 0;                      // bases per scan lane
  uint64_t tile = 0;                  // chunk * LANES_PER_TILE
  uint64_t total_packed = 0;          // bases in packed space (multiple of tile; first and last tile are padding)
  uint64_t total_bases = 0;           // real bases
  std::vector<std::string> names;
  std::vector<ContigInfo> contigs;
  std::vector<uint32_t> codes;        // total_packed / 16
  std::vector<uint32_t> mask;         // total_packed / 32
  std::vector<Run> runs;              // sorted by start
  std::vector<TileInfo> tiles;        // total_packed / tile
  std::vector<uint32_t> masked_tiles; // indices of tiles with flag 1 (scanned by the exception-aware kernel variant)
  std::string genome_build = "unknown";
  // absent[i] != 0: contig i has its name and length (the window table, the bins and every coordinate count it) but no bases here -- a
  // process of a multi-GPU job holds the contigs its window range touches and nothing else; empty = all contigs resident
  std::vector<uint8_t> absent;
  bool is_absent(size_t i) const { return i < absent.size() && absent[i] != 0; }

  // Upper-cased base at a packed position (what the reference sees after StringUtil.toUpperCase, SearchReference.scala:67).
  char base_upper(uint64_t gpos) const;
  // Original byte class for windowing: returns the run containing gpos or nullptr.
  const Run* run_at(uint64_t gpos) const;
};

// Packs ASCII contigs.  `threads` <= 0 picks the hardware concurrency.
class WorkerPool;
// (pool: the targets are packed in blocks by its workers -- the variant branch packs 65 536 windows per batch)
void pack_targets_dense(PackedRef& out, int n, const uint64_t* lengths, const uint8_t* const* bases, WorkerPool* pool = nullptr);
void pack_reference(PackedRef& out, int n_contigs, const char* const* names, const uint64_t* lengths,
                    const uint8_t* const* bases, const char* genome_build, int threads);

// Index of the last run with start <= gpos, or -1.
CAL_HD inline int64_t run_floor(const Run* runs, int64_t n_runs, uint64_t gpos) {
  int64_t lo = 0, hi = n_runs;  // first run with start > gpos
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (runs[mid].start <= gpos) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

// Window k of a contig per SearchReference.windowIterator (SearchReference.scala:52-68): start = k*step (only while
// start < len-1), end = min(len, start+W), then leading/trailing UPPER-CASE 'N' bytes are trimmed.  Returns false when
// the contig has no window k.  a/b are 0-based half-open contig offsets; b - a may be <= 0 (the reference then emits
// its 1-byte `empty` sentinel, which the length filter at SearchReference.scala:536 always drops for real guides).
CAL_HD inline bool window_bounds(const Run* runs, int64_t n_runs, uint64_t gbase, uint64_t len, int W, int step, uint64_t k,
                                 int64_t& a, int64_t& b) {
  uint64_t s = k * (uint64_t)step;
  if (len < 2 || s >= len - 1) return false;
  uint64_t e = s + (uint64_t)W;
  if (e > len) e = len;
  a = (int64_t)s; b = (int64_t)e;
  int64_t r = run_floor(runs, n_runs, gbase + s);
  if (r >= 0 && runs[r].ch == 'N' && gbase + s < runs[r].start + runs[r].len) {
    uint64_t ne = runs[r].start + runs[r].len - gbase;
    a = (int64_t)(ne < e ? ne : e);
  }
  if (a < b) {
    r = run_floor(runs, n_runs, gbase + (uint64_t)b - 1);
    if (r >= 0 && runs[r].ch == 'N' && gbase + (uint64_t)b - 1 < runs[r].start + runs[r].len) {
      int64_t ns = (int64_t)(runs[r].start - gbase);
      b = ns > a ? ns : a;
    }
  }
  return true;
}

// Number of window starts on a contig: |Range(0, len-1, step)|.
CAL_HD inline uint64_t window_count(uint64_t len, int step) {
  if (len < 2) return 0;
  return (len - 2) / (uint64_t)step + 1;
}

}  // namespace calitas
