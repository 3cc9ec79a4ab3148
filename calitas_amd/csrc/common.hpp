// common.hpp -- data layout shared by the host code and the HIP kernels of libcalitas_hip.
//
// Packed reference ("packed space"):
//   * every contig starts at a multiple of TILE bases (TILE = 256 lanes x chunk bases, chosen at set_reference time) and
//     is followed by at least one chunk of padding, so a scan tile never spans two contigs and a tile's halo never
//     sees another contig's real bases;
//   * codes[]: 2 bits per base, 16 bases per uint32, base i of a word at bits [2i, 2i+1]; A=0 C=1 G=2 T/U=3;
//   * mask[]:  1 bit per base, 32 bases per uint32; 1 = "exception" (non-ACGT byte or padding). For an exception base
//     the 2-bit code says what kind: 0 = can never match (N, n, padding, unknown bytes; GuideAlignmentScorer forces a
//     mismatch for N/n, SequentialGuideAligner.scala:144), 1 = IUPAC ambiguity code other than N (the scan treats it
//     as a wildcard, the aligner kernel looks the exact code up in runs[]);
//   * runs[]: maximal runs of identical non-ACGT bytes, sorted by packed position, original byte kept so the
//     windowing code can tell upper-case 'N' (trimmed, SearchReference.scala:58-59) from everything else;
//   * tile_info[]: per scan tile the contig it belongs to and a flag (0 = plain, 1 = has exception bases in the tile
//     or its halo, 2 = nothing but upper-case N / padding in tile and halo => no window can contain any of its bases).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define CAL_HD __host__ __device__
#else
#define CAL_HD
#endif

namespace calitas {

constexpr int LANES_PER_TILE = 256;
constexpr int MAX_L = 32;            // protospacer rows of the scan bit-vector
constexpr int MAX_PAMS = 8;
constexpr int MAX_PAM_LEN = 16;
constexpr int MAX_GUIDES = 64;
constexpr int STRIP_MAX_COLS = 96;   // columns of one aligner-kernel strip (16 candidate columns + span + 1; span <= 2 * MAX_L)
constexpr int RAW_MAX_OPS = 80;      // guide-part ops of one raw alignment (L + max extra genome bases)

struct Run {             // exception run in packed space
  uint64_t start;        // packed position of the first base
  uint32_t len;
  uint8_t ch;            // original byte ('N', 'n', 'R', ...); 0 = padding
  uint8_t pad[3];
};

struct ContigInfo {
  uint64_t gbase;        // packed position of base 0
  uint64_t len;
};

struct TileInfo {
  uint32_t contig;
  uint32_t flag;         // see header comment
};

// IUPAC set of a byte as a 4-bit mask A=1 C=2 G=4 T=8; 0 for anything that is not an IUPAC letter.
CAL_HD inline int iupac_mask(unsigned char b) {
  switch (b & 0xDF) {  // ASCII upper-case fold (letters only; other bytes fall to default)
    case 'A': return 1;  case 'C': return 2;  case 'G': return 4;  case 'T': return 8;  case 'U': return 8;
    case 'M': return 3;  case 'R': return 5;  case 'W': return 9;  case 'S': return 6;  case 'Y': return 10;  case 'K': return 12;
    case 'V': return 7;  case 'H': return 11; case 'D': return 13; case 'B': return 14; case 'N': return 15;
    default: return 0;
  }
}

// Target-side "tmask" used by the aligner kernel: bits 0-3 IUPAC set, bit 4 = pairing is forced to a mismatch
// (target N/n, padding, unknown byte).  For N the set bits stay 15 so that '='/'X' by compatibility
// still sees N as compatible.
CAL_HD inline int target_mask(unsigned char b) {
  int m = iupac_mask(b);
  if ((b & 0xDF) == 'N') return 15 | 16;
  if (m == 0) return 16;
  return m;
}

// Scoring derived from the four net costs (SequentialGuideAligner.scala:192-208,213).
struct Scores {
  int match, mismatch, pam_match, pam_mismatch, query_gap, target_gap, worst_guide_diff;
};
inline int iabs(int x) { return x < 0 ? -x : x; }
inline Scores derive_scores(int mmNet, int pamNet, int genomeGapNet, int guideGapNet) {
  Scores s;
  s.match = iabs(mmNet) / 2;
  s.mismatch = -(iabs(mmNet) - s.match);
  s.query_gap = -iabs(guideGapNet);
  s.target_gap = -iabs(genomeGapNet) + s.match;
  s.pam_match = iabs(pamNet) / 2;
  s.pam_mismatch = -(iabs(pamNet) - s.pam_match);
  int w = -iabs(mmNet);
  if (-iabs(genomeGapNet) < w) w = -iabs(genomeGapNet);
  if (-iabs(guideGapNet) < w) w = -iabs(guideGapNet);
  s.worst_guide_diff = w;
  return s;
}

// Per-guide constants uploaded to the device for one search.
struct GuideDev {
  uint32_t peq_a[8];      // scan, left-to-right pass: Eq vector per (exception<<2 | code), guide rows top-aligned in 32 bits
  uint32_t peq_b[8];      // scan, right-to-left pass (target is read complemented)
  uint64_t row_sets[2];   // scan (row-wise kernel): the same IUPAC sets, 4 bits per protospacer row, rows 0-15 / 16-31
  uint8_t qmask[MAX_L];   // IUPAC set of each row of the aligner-space query (guideFw, or guideRc for a 5' PAM)
  uint8_t pam_mask[MAX_PAMS][MAX_PAM_LEN];  // IUPAC sets of the aligner-space PAMs (pamFw or pamRc)
  uint8_t pam_len[MAX_PAMS];
  int32_t L;
  int32_t n_pams;         // 0 for a PAM-less guide
  int32_t scan_max_edits; // E: a bottom-row score >= min_guide_score implies <= E edits
  int32_t min_guide_score;
  int32_t span;           // L + max genome-only bases of any alignment scoring >= min_guide_score
  int32_t cli_length;
  int32_t max_guide_diffs;      // -d for this guide
  int32_t max_pam_mismatches;   // -p
  int32_t max_diffs_filtering;  // d + g + p (SequentialGuideAligner.scala:249)
  int32_t pam5;                 // 1 for a 5' PAM guide (the reverse-complemented pass then yields the forward-strand list)
};

struct SearchDev {        // scalar parameters of one search
  int32_t window_size, step, n_guides;
  int32_t max_guide_diffs, max_pam_mismatches, max_gaps, max_diffs_filtering;
  int32_t match, mismatch, pam_match, pam_mismatch, query_gap, target_gap;
  int32_t eqx_by_score;   // '=' / 'X' by pairing score instead of compatibility
  int32_t per_matrix;     // one alignment per bottom-row matrix cell >= minScore instead of per end column
  int32_t chrom_index;
};

// One 16-base group with candidate end columns, written by the scan kernel.
struct ScanRecord {
  uint32_t gword;         // packed position / 16
  uint32_t info;          // bits 0-15 candidate mask (bit k = base k of the word), bit 16 direction (0 = A, 1 = B), bits 17-23 guide
};

// One alignment after PAM extension (output of extendAndFilterRight), in aligner space.
struct RawAln {
  uint32_t contig;
  uint32_t window_k;      // index k of the window (start = k * step) on its contig
  int32_t score;
  uint16_t t_start;       // 1-based first target column, strand space
  uint16_t t_end_guide;   // 1-based last target column of the guide part
  uint8_t dir;            // 0 = A (target as is), 1 = B (reverse-complemented target)
  uint8_t guide;
  int8_t pam;             // PAM index or -1
  uint8_t offset;         // genome bases skipped between guide and PAM
  uint8_t n_ops;          // guide-part ops
  uint8_t pad;            // per-matrix enumeration: which matrix the traceback started in (0 Diag, 1 Left, 2 Up); else 0
  uint16_t pam_x;         // bit i set = PAM position i is 'X'
  uint8_t ops[RAW_MAX_OPS / 4];  // 2 bits per op in traceback (reverse) order: 0 '=', 1 'X', 2 'I', 3 'D'
};

// Counts over the packed ops of a RawAln (2 bits per op, traceback order, unused slots zero), computed on whole words so that
// no kernel needs a per-op loop with a runtime index into a private array.
struct OpCounts {
  int non_eq;    // ops != '='           (X, I, D)
  int gaps;      // ops >= 'I'           (I, D)
  int not_ins;   // ops != 'I' among the first n_ops (=, X, D: the ops that consume a target base)
  int lead_d;    // run of D at the end of the traceback order (= start of the aligner order)
  int trail_d;   // run of D at the start of the traceback order (= end of the aligner order)
};
CAL_HD inline int popc32(uint32_t x) { return __builtin_popcount(x); }
static_assert(RAW_MAX_OPS / 16 == 5, "OpsWords holds five words");
struct OpsWords {       // the five ops words of a RawAln held in registers (selected by comparison, never indexed)
  uint32_t a, b, c, d, e;
  CAL_HD uint32_t word(int k) const { return k == 0 ? a : k == 1 ? b : k == 2 ? c : k == 3 ? d : e; }
  CAL_HD int op(int i) const { return (int)((word(i >> 4) >> ((i & 15) * 2)) & 3u); }
};
CAL_HD inline OpsWords load_ops_words(const uint8_t* ops) {   // RawAln::ops is 4-byte aligned
  const uint32_t* w = reinterpret_cast<const uint32_t*>(ops);
  return OpsWords{w[0], w[1], w[2], w[3], w[4]};
}
CAL_HD inline OpCounts count_ops(const OpsWords& w, int n_ops) {
  OpCounts c{0, 0, 0, 0, 0};
  int ins = 0;
  const uint32_t ws[5] = {w.a, w.b, w.c, w.d, w.e};           // constant indices only: stays in registers
  for (int k = 0; k < 5; k++) {
    const uint32_t lo = ws[k] & 0x55555555u, hi = (ws[k] >> 1) & 0x55555555u;
    c.non_eq += popc32(lo | hi);
    c.gaps += popc32(hi);
    ins += popc32(hi & ~lo);
  }
  c.not_ins = n_ops - ins;
  for (int i = 0; i < n_ops && w.op(i) == 3; i++) c.trail_d++;
  for (int i = n_ops - 1; i >= 0 && w.op(i) == 3; i--) c.lead_d++;
  return c;
}

// One aligner job = one scan record x one window that holds some of its candidate columns; one slab per job.  The slab is the job's
// whole life: expand_kernel writes its head (the strip's geometry, the guide's row sets, the target masks tb[ntb]) with one lane
// group per job, align_kernel reads exactly that -- one 16-byte load per lane, issued a job ahead --, fills the strip and adds the
// trace bytes tr[L][stride] and which candidate columns passed, trace_kernel walks it.  Slabs have a fixed size per search and the
// job's number as their address, so the hand-overs need no atomics beyond the one that numbers the jobs.
struct SlabHeader {
  uint32_t pass_mask;     // (align_kernel) bit x: the x-th candidate column of this job reached min_guide_score; 0 from expand_kernel
  uint32_t contig;
  uint32_t window_k;
  int32_t n;              // window length
  int32_t c0;             // strip boundary column (strip columns are c0+1 .. c0+ncols)
  uint16_t ncols, ntb;
  uint8_t dir, guide, true_border, L;
  uint16_t stride, pad;
  uint16_t j[16];         // (align_kernel) strand-space end column (1-based) of candidate x, for the candidates that passed
  // what align_kernel needs besides the geometry above, so that a job is ONE read of its slab's head:
  uint8_t qmask[MAX_L];   // IUPAC set of each query row (GuideDev::qmask of the job's guide)
  int32_t min_score;      // GuideDev::min_guide_score
  uint32_t sel;           // candidate columns of the record's 16-base word that lie inside this window (bit b = base b)
  int32_t jbase;          // strand-space column of bit 0: j = jbase + b (dir 0) or jbase - b (dir 1)
  int32_t reserved[5];
};
static_assert(sizeof(SlabHeader) == 128, "slab header layout");

}  // namespace calitas

// Wave priority of the kernels that follow a scan (align ... rows).  They share SIMDs with the scan of the next contig range, whose
// waves issue a vector instruction every cycle they can; at the default priority a small kernel's waves got one issue slot in
// five to nine and took 40-70 us instead of 5-10 (rocprofv3 timeline).  The scan stays at priority 0.
#if defined(__HIPCC__)
#define CALITAS_TAIL_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

#ifdef CALITAS_ALLOC_DEBUG
#include "dbg_alloc.hpp"   // make DBG_ALLOC=1: allocation registry dumped on abort
#endif
