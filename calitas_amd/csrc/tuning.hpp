// tuning.hpp -- every environment switch of the library, in one place.
//
// The product has one behaviour; these switches exist for measurements (A/B runs, tools/ab_env.py, tools/sweep_env.py), for tests that
// force a fallback path, and for a caller that shares the card.  They are read when a call starts a stage (tests change them between
// calls of one process), always through TUNE_GET / TUNE_ON / TUNE_SET, which refuse a name that is not in the table below AT COMPILE TIME
// (a static_assert on the literal) -- so the table is complete, `calitas_switches()` (include/calitas_hip.h) prints it, and nothing in
// the library stops the process over a switch.  Switches that make the library return WRONG bytes for the sake of a timing experiment
// exist only in builds made with `make EXPERIMENTS=1` (-DCALITAS_EXPERIMENTS); the shipped library does not know their names.
#pragma once
#include <cstdlib>
#include <cstring>

namespace calitas {
namespace tune {

struct Switch { const char* name; const char* values; const char* what; };

// kind: D = diagnostics, F = forces a fallback / alternative path that returns the same bytes (tests), T = tuning (measured defaults,
// DESIGN.md says where), R = resources
constexpr Switch kSwitches[] = {
  {"CALITAS_TRACE", "1 | 2", "D: one line per call / stage on stderr; 2: the host-side time line of calitas_search_hits (microsecond marks)"},
  {"CALITAS_TWIN_STATS", "1", "D: variant branch: how many description-less hits of variant windows repeat a reference hit (host merge only)"},
  {"CALITAS_THREADS", "n", "R: worker pool size (default: the CPUs of the process, at most 16)"},
  {"CALITAS_DEVICE_BUDGET_MB", "n", "R: device memory the library may plan with (reference + search scratch); a search beyond it runs one pass per contig or is refused"},
  {"CALITAS_HOST_FILTER", "1", "F: per-window filter (SGA:315-320) on the host instead of select.hip"},
  {"CALITAS_HOST_HITS", "1", "F: removeOverlaps / sort / rows on the host (post.cpp) instead of hits.hip / binned.hip"},
  {"CALITAS_VARIANTS_HOST", "1", "F: variant branch: merge alignment records on the host (round 3) instead of bringing the variant windows' hits into the device's row stage"},
  {"CALITAS_FAIL_ALIGN_BATCH", "k", "F: variant branch: the k-th batch of variant windows (0-based) fails in the aligner stage (tests of the stages' error path)"},
  {"CALITAS_VARIANTS_COMPACT", "1", "T: variant branch: the per-contig texts of the reference passes cross PCIe as compact rows (default off: the branch is bound by its host threads)"},
  {"CALITAS_VARIANTS_ROWS", "all | device", "F: calitas_search_variants makes the row of every hit of a variant window up front instead of the kept ones' on demand / sends the kept rows to the device instead of writing them into the text on the host"},
  {"CALITAS_SEQUENTIAL", "1", "F: calitas_search_hits as one pass per contig whatever the size"},
  {"CALITAS_SDMA", "0", "F: text copies with hipMemcpyAsync instead of the SDMA engine (dma.cpp)"},
  {"CALITAS_BINNED", "1 | 0 | last | from1", "T/F: the per-bin tail for every range / none / the last range only (default: calls of one or two ranges, the last range of three, every window range)"},
  {"CALITAS_BATCH_BINNED", "1", "T: guide batches on large references keep the per-bin tail (default: general kernels from 2 Gb on)"},
  {"CALITAS_OWN_GENERAL_OFF", "1", "F: a window range with a crowded bin searches its contigs whole (round 3) instead of finishing on the general kernels with HitsOwn"},
  {"CALITAS_BINNED_COMPLEX", "1", "F: every bin through the wave-per-bin kernel"},
  {"CALITAS_BINNED_TEXT_KB", "n", "F: first guess of the per-bin text buffer (forces the regrow path)"},
  {"CALITAS_BINNED_HOST_TEXT", "0 | 1", "F/T: short texts written into page-locked host memory by the rows kernel (default 1)"},
  {"CALITAS_BINNED_HOST_TEXT_KB", "n", "T: ... up to this size (default 128)"},
#ifdef CALITAS_EXPERIMENTS
  {"CALITAS_BINNED_SKIP", "1 | 2 | 3", "D: timing experiments only (the text is wrong): skip the wave-per-bin kernel / the rows"},
  {"CALITAS_BATCH_TEXT", "skip | copy", "D: timing experiments only (the texts are wrong): a batch's rows stay on the device / cross the bus but are not expanded"},
#endif
  {"CALITAS_TEXT_IN_PLACE_OFF", "1", "F: the last range's text takes the copy instead of being written to its final place by the rows kernel"},
  {"CALITAS_CHUNKS", "k | a:b:c", "T: contig ranges of a chunked calitas_search_hits (default 5.8:2.9:1.3 from 2 Gb, 5:3 from 600 Mb, 3:2 from 256 Mb)"},
  {"CALITAS_CHUNK", "64..512", "T: bases per scan lane chunk (set_reference; default by genome size)"},
  {"CALITAS_INPUTS_FIRST", "0 | 1 | 2", "T: where the ranges' small inputs are queued (default 2)"},
  {"CALITAS_LANE_SETUP", "0", "F: separate stream commands instead of the one-launch lane setup"},
  {"CALITAS_LANE_PRIO", "low | low0 | low01", "T: lanes' streams at low priority (experiment)"},
  {"CALITAS_ALIGN_LPJ", "32", "F: two jobs of 32 lanes per aligner wave even for guides of up to 20 rows"},
  {"CALITAS_TAIL_PRIO_NARROW", "0", "T: expand / align / trace of the ranges whose tail runs beside the next range's scan at the scan's wave priority (experiment)"},
  {"CALITAS_ALIGN_PACK", "0", "F: one job per lane group in the aligner (align_kernel) where two would fit (align_pk_kernel: cells in sixteen bits)"},
  {"CALITAS_ALIGN_BLOCKS", "n", "T: align_kernel grid, units of four one-wave workgroups (default 512)"},
  {"CALITAS_ALIGN_BLOCKS_NARROW", "n", "T: ... for the ranges whose tail runs beside the next scan"},
  {"CALITAS_TRACE_BLOCKS", "n", "T: trace_kernel grid (default 2048)"},
  {"CALITAS_TRACE_BLOCKS_NARROW", "n", "T: ... for the ranges whose tail runs beside the next scan"},
  {"CALITAS_FREE_NOW", "1", "F/T: calitas_free of a block of gigabytes hands its pages back before it returns (default: on the library's own thread)"},
  {"CALITAS_BATCH_LANES", "1..8", "T: guides in flight in calitas_search_hits_batch (default 5)"},
  {"CALITAS_BATCH_SCAN_STREAMS", "1..4", "T: a batch's scans on one stream, one after the other, or taking turns on several"},
  {"CALITAS_COMPACT_ROWS", "0", "F/T: full rows over PCIe instead of compact rows + host expansion (batches, the leading ranges of a chunked call)"},
  {"CALITAS_COMPACT_LANES", "n", "T: how many leading ranges of a chunked call move compact rows (default: all of three or more, all but the last of two)"},
  {"CALITAS_COMPACT_PIECE_KB", "n", "T: compact text copied and expanded in pieces of this size (default: one piece)"},
  {"CALITAS_EXPAND_THREADS", "n", "T: workers that expand compact rows (default: the whole pool)"},
};

constexpr bool same(const char* a, const char* b) {
  while (*a && *a == *b) { ++a; ++b; }
  return *a == *b;
}
constexpr bool known(const char* name) {
  for (const Switch& s : kSwitches) if (same(s.name, name)) return true;
  return false;
}

}  // namespace tune
}  // namespace calitas

// getenv for a switch of the table.  NAME is a string literal; one that is not in the table does not compile.
#define TUNE_GET(NAME) ([]() -> const char* { static_assert(::calitas::tune::known(NAME), "switch missing from tuning.hpp's table"); return std::getenv(NAME); }())
#define TUNE_ON(NAME) ([]() -> bool { const char* e_ = TUNE_GET(NAME); return e_ && std::atoi(e_) != 0; }())
#define TUNE_SET(NAME) (TUNE_GET(NAME) != nullptr)
