// kernels.hip -- the two hand-written gfx950 kernels of the SearchReference hot path.
//
//  scan_kernel   exact filter for "bottom-row glocal score >= minGuideScore" (the enumeration rule of
//                fgbio Aligner.align(query, target, minScore), SequentialGuideAligner.scala:261,278,295,299).
//                With the reference's linear gap costs a bottom-row score >= minGuideScore implies at most E edits
//                (SearchReference.scala:432-441), so the filter is Myers' bit-vector edit distance: one 32-bit
//                column vector per lane, the protospacer rows top-aligned so the row-L delta falls out of the
//                shift as a carry.  Every lane owns CHUNK consecutive bases of a 256-lane tile that the workgroup
//                streams from HBM into LDS with coalesced 16-byte loads; it runs the tile once left-to-right
//                (target as is) and once right-to-left (reverse-complemented target) with 32 warm-up columns.
//                Integer VALU work; no MFMA.  Emits one record per 16-base word that holds a candidate column.
//  align_kernel  the glocal DP itself on the strips the scan flagged: three score matrices (Diag/Left/Up) with
//                fgbio's tie rules, antidiagonal wavefront with one lane per query row, neighbours exchanged with
//                DPP wave shifts, trace matrix staged in LDS, then traceback, '='/'X' ops, and the PAM extension
//                of extendAndFilterRight (SequentialGuideAligner.scala:433-492), one lane per candidate column.
//                Only columns [j - span - 1, j] are filled: by the locality argument in DESIGN.md this reproduces
//                the value and the trace of every cell on the optimal path of a candidate (L, j) bit for bit.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include <chrono>
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "tuning.hpp"
#include "refpack.hpp"
#include "kernels.hpp"
#include "mailbox.hpp"

namespace calitas {

// ------------------------------------------------------------------------------------------------------------------
// scan_kernel
// ------------------------------------------------------------------------------------------------------------------

// One column of Myers' bit-vector algorithm (search variant: free start in the text, so no carry into row 1).
// Guide rows occupy the top L bits; the padding bits below keep Pv=1, Mv=0 as long as eq has them clear.
// The horizontal deltas of row L sit in bit 31, so the two shifts are written as x+x and their carry-outs update
// the running bottom-row score (v_add_co / v_addc on gfx950).
__device__ __forceinline__ void myers_step(uint32_t eq, uint32_t& pv, uint32_t& mv, int& score, int& smin) {
  // Hand-scheduled: 13 VALU instructions per column (hipcc's own selection of the same expression needs 17).
  // v_min_i32 issues at ~0.6x the rate of v_or_b32 on gfx950 (tools/valu_bench2.hip), hence the sign-bit accumulator.
  //   t  = (eq & pv) + pv
  //   mh = pv & ((t ^ pv) | eq)            bitop3(pv, t, eq)  = 0xb0
  //   xh = (t ^ pv) | eq                   bitop3(t, pv, eq)  = 0xbe
  //   ph = mv | ~(xh | pv)                 bitop3(mv, xh, pv) = 0xf1
  //   xv = eq | mv
  //   ph <<= 1, score += carry;  mh <<= 1, score -= carry
  //   pv = mh | ~(xv | ph)                 bitop3(mh, xv, ph) = 0xf1
  //   mv = ph & xv
  //   sacc |= score          (the caller keeps score biased by -(E+1): the sign bit of sacc = "some column <= E")
  uint32_t t, xh, ph, mh, xv;
  asm("v_and_b32 %5, %9, %0\n\t"
      "v_add_u32 %5, %5, %0\n\t"
      "v_bitop3_b32 %7, %0, %5, %9 bitop3:0xb0\n\t"
      "v_bitop3_b32 %6, %5, %0, %9 bitop3:0xbe\n\t"
      "v_or_b32 %8, %9, %1\n\t"
      "v_bitop3_b32 %4, %1, %6, %0 bitop3:0xf1\n\t"
      "v_add_co_u32 %7, vcc, %7, %7\n\t"
      "v_subb_co_u32 %2, vcc, %2, 0, vcc\n\t"
      "v_add_co_u32 %4, vcc, %4, %4\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_bitop3_b32 %0, %7, %8, %4 bitop3:0xf1\n\t"
      "v_and_b32 %1, %4, %8\n\t"
      "v_or_b32 %3, %3, %2"
      : "+v"(pv), "+v"(mv), "+v"(score), "+v"(smin), "=&v"(ph), "=&v"(t), "=&v"(xh), "=&v"(mh), "=&v"(xv)
      : "v"(eq)
      : "vcc");
}

// Two independent columns (pass A and pass B) interleaved instruction by instruction so that dependent VALU
// instructions of one recurrence are never adjacent; pass B carries through an SGPR pair instead of VCC.
__device__ __forceinline__ void myers_step2(uint32_t eqa, uint32_t& pva, uint32_t& mva, int& sca, int& mina,
                                            uint32_t eqb, uint32_t& pvb, uint32_t& mvb, int& scb, int& minb) {
  uint32_t ta, xha, pha, mha, xva, tb, xhb, phb, mhb, xvb;
  unsigned long long cb;
  asm("v_and_b32 %9, %19, %0\n\t"
      "v_and_b32 %14, %20, %4\n\t"
      "v_add_u32 %9, %9, %0\n\t"
      "v_add_u32 %14, %14, %4\n\t"
      "v_bitop3_b32 %11, %0, %9, %19 bitop3:0xb0\n\t"
      "v_bitop3_b32 %16, %4, %14, %20 bitop3:0xb0\n\t"
      "v_bitop3_b32 %10, %9, %0, %19 bitop3:0xbe\n\t"
      "v_bitop3_b32 %15, %14, %4, %20 bitop3:0xbe\n\t"
      "v_or_b32 %12, %19, %1\n\t"
      "v_or_b32 %17, %20, %5\n\t"
      "v_bitop3_b32 %8, %1, %10, %0 bitop3:0xf1\n\t"
      "v_bitop3_b32 %13, %5, %15, %4 bitop3:0xf1\n\t"
      "v_add_co_u32 %11, vcc, %11, %11\n\t"
      "v_add_co_u32 %16, %18, %16, %16\n\t"
      "v_subb_co_u32 %2, vcc, %2, 0, vcc\n\t"
      "v_subb_co_u32 %6, %18, %6, 0, %18\n\t"
      "v_add_co_u32 %8, vcc, %8, %8\n\t"
      "v_add_co_u32 %13, %18, %13, %13\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32 %6, %18, 0, %6, %18\n\t"
      "v_bitop3_b32 %0, %11, %12, %8 bitop3:0xf1\n\t"
      "v_bitop3_b32 %4, %16, %17, %13 bitop3:0xf1\n\t"
      "v_and_b32 %1, %8, %12\n\t"
      "v_and_b32 %5, %13, %17\n\t"
      "v_or_b32 %3, %3, %2\n\t"
      "v_or_b32 %7, %7, %6"
      : "+v"(pva), "+v"(mva), "+v"(sca), "+v"(mina), "+v"(pvb), "+v"(mvb), "+v"(scb), "+v"(minb),
        "=&v"(pha), "=&v"(ta), "=&v"(xha), "=&v"(mha), "=&v"(xva),
        "=&v"(phb), "=&v"(tb), "=&v"(xhb), "=&v"(mhb), "=&v"(xvb), "=&s"(cb)
      : "v"(eqa), "v"(eqb)
      : "vcc");
}

// Pair index of bases 2j and 2j+1 of a code word (+ their exception bits for masked tiles):
// bits 0-1 code of base 2j, bits 2-3 code of base 2j+1, bit 4 / bit 5 their exception bits.
template <bool MASKED>
__device__ __forceinline__ uint32_t pair_index(uint32_t word, uint32_t mbits, int j) {
  uint32_t idx = (word >> (4 * j)) & 15u;
  if (MASKED) idx |= ((mbits >> (2 * j)) & 3u) << 4;
  return idx;
}

// Replays one 16-base word with a per-column threshold test (taken only when the word's minimum score is <= E).
template <bool MASKED, bool FORWARD>
__device__ __noinline__ uint32_t replay_word(const uint2* tab, uint32_t word, uint32_t mbits, uint32_t pv, uint32_t mv, int score) {
  uint32_t hm = 0;   // score is biased by -(E+1): negative = candidate column
  for (int s = 0; s < 16; s++) {
    const int k = FORWARD ? s : 15 - s;
    const uint2 e = tab[pair_index<MASKED>(word, mbits, k >> 1)];
    int unused = 0;
    myers_step((k & 1) ? e.y : e.x, pv, mv, score, unused);
    hm |= (uint32_t)(score < 0) << k;
  }
  return hm;
}

// Records are staged in LDS and flushed once per tile: a returning atomic on ONE global word completes at only
// ~90 per microsecond chip-wide (MI355X_MICROARCH.md, row "dequeue"), which a per-record append would approach.
constexpr int SCAN_STAGE = 192;

__device__ __forceinline__ void stage_record(const ScanArgs& a, ScanRecord* s_recs, uint32_t* s_nrec, uint32_t gword, uint32_t info) {
  ScanRecord r;
  r.gword = gword; r.info = info;
  const uint32_t slot = atomicAdd(s_nrec, 1u);          // LDS atomic
  if (slot < (uint32_t)SCAN_STAGE) { s_recs[slot] = r; return; }
  const uint32_t g = atomicAdd(a.rec_count, 1u);        // stage full (dense tile): append directly
  if (g < a.rec_capacity) a.recs[g] = r;
}

template <int CHUNK, bool MASKED>
__device__ __forceinline__ void scan_tile(const ScanArgs& a, uint32_t tile, uint32_t* s_codes, uint2* s_tab, ScanRecord* s_recs,
                                          uint32_t* s_nrec) {
  constexpr int WPC = CHUNK / 16;           // code words per lane chunk
  constexpr int MPC = CHUNK / 32;           // mask words per lane chunk
  constexpr int CSTR = WPC + 1;             // padded stride: lane l reads word l*CSTR + k -> conflict-free banks
  constexpr int NV = LANES_PER_TILE + 2;    // virtual chunks: left halo, 256 lanes, right halo
  constexpr int TAB = MASKED ? 64 : 16;     // entries of one pair table
  const int tid = threadIdx.x;

  // ---- stream the tile (+ one halo chunk each side) into LDS: 16-byte coalesced loads, padded scatter ----
  const uint64_t w0 = (uint64_t)tile * (LANES_PER_TILE * WPC);  // first code word of the tile
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.codes + (w0 - WPC));
    constexpr int NQ = NV * WPC / 4;
    for (int q = tid; q < NQ; q += LANES_PER_TILE) {
      const uint4 v = src[q];
      const int i = q * 4;
      const int vc = i / WPC, k = i % WPC;  // WPC is a multiple of 4, so the four words stay in one chunk
      uint32_t* d = &s_codes[vc * CSTR + k];
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  const uint32_t* cw = &s_codes[(tid + 1) * CSTR];
  // exception bits are rare (N-run edges, contig ends, IUPAC codes): the few tiles that have them read the 1-bit
  // mask straight from global memory, one word per 32 bases, instead of spending LDS on it
  const uint32_t* gm = a.mask + (w0 / 2 + (uint64_t)tid * MPC);
  const uint32_t gword0 = (uint32_t)(w0 + (uint64_t)tid * WPC);

  for (int gi = 0; gi < a.n_guides; gi++) {
    const GuideDev& g = a.guides[gi];
    __syncthreads();                        // previous guide's table no longer in use / tile data visible
    for (int i = tid; i < 2 * TAB; i += LANES_PER_TILE) {
      const uint32_t* peq = (i >= TAB) ? g.peq_b : g.peq_a;
      const int idx = i & (TAB - 1);
      const int lo = (idx & 3) | ((idx >> 4) & 1) << 2, hi = ((idx >> 2) & 3) | ((idx >> 5) & 1) << 2;
      s_tab[i] = make_uint2(peq[lo], peq[hi]);
    }
    __syncthreads();
    const int L = g.L, E = g.scan_max_edits;
    const int warm = (L + E + 15) >> 4;     // warm-up words (host guarantees warm <= WPC)
    const uint2* tabA = &s_tab[0];
    const uint2* tabB = &s_tab[TAB];

    // Pass A runs left to right over the chunk (target as is); pass B right to left (target complemented = left to
    // right over the reverse complement).  The two recurrences are independent, so they share one loop for ILP.
    // Scores are kept biased by -(E+1): "score <= E" is the sign bit.
    uint32_t pvA = 0xFFFFFFFFu, mvA = 0u, pvB = 0xFFFFFFFFu, mvB = 0u;
    int scA = L - E - 1, scB = L - E - 1;
    const int n_it = WPC + warm;
    for (int it = 0; it < n_it; it++) {
      const int wa = it - warm;             // < 0: tail of the left neighbour's chunk (cw[wa - 1] skips the pad word)
      const int wb = WPC - 1 + warm - it;   // >= WPC: head of the right neighbour's chunk (cw[wb + 1])
      const uint32_t wordA = (wa >= 0) ? cw[wa] : cw[wa - 1];
      const uint32_t wordB = (wb < WPC) ? cw[wb] : cw[wb + 1];
      uint32_t mA = 0, mB = 0;
      if (MASKED) {
        mA = (gm[wa >> 1] >> ((wa & 1) * 16)) & 0xFFFFu;
        mB = (gm[wb >> 1] >> ((wb & 1) * 16)) & 0xFFFFu;
      }
      const uint32_t pvA0 = pvA, mvA0 = mvA, pvB0 = pvB, mvB0 = mvB;
      const int scA0 = scA, scB0 = scB;
      int accA = 0, accB = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint2 ea = tabA[pair_index<MASKED>(wordA, mA, j)];
        const uint2 eb = tabB[pair_index<MASKED>(wordB, mB, 7 - j)];
        myers_step2(ea.x, pvA, mvA, scA, accA, eb.y, pvB, mvB, scB, accB);
        myers_step2(ea.y, pvA, mvA, scA, accA, eb.x, pvB, mvB, scB, accB);
      }
      if (wa >= 0 && accA < 0) {
        const uint32_t hm = replay_word<MASKED, true>(tabA, wordA, mA, pvA0, mvA0, scA0);
        stage_record(a, s_recs, s_nrec, gword0 + (uint32_t)wa, hm | ((uint32_t)gi << 17));
      }
      if (wb < WPC && accB < 0) {
        const uint32_t hm = replay_word<MASKED, false>(tabB, wordB, mB, pvB0, mvB0, scB0);
        stage_record(a, s_recs, s_nrec, gword0 + (uint32_t)wb, hm | (1u << 16) | ((uint32_t)gi << 17));
      }
    }
  }
}

// One workgroup per tile of the packed space.  Dead tiles (nothing but upper-case N / padding, which every window
// trims away) exit at once; tiles with exception bases take the MASKED instantiation (block-uniform branch).
template <int CHUNK>
__global__ __launch_bounds__(LANES_PER_TILE) void scan_kernel(ScanArgs a) {
  __shared__ uint32_t s_codes[(LANES_PER_TILE + 2) * (CHUNK / 16 + 1)];
  __shared__ uint2 s_tab[2 * 64];           // [direction][pair index] -> (Eq of base 2j, Eq of base 2j+1)
  __shared__ ScanRecord s_recs[SCAN_STAGE];
  __shared__ uint32_t s_nrec, s_base;
  const uint32_t tile = blockIdx.x * a.tile_stride + a.tile_offset;
  const TileInfo ti = a.tiles[tile];
  if (ti.flag == 2u || ti.contig == 0xFFFFFFFFu) return;
  if (a.chrom_index >= 0 && ti.contig != (uint32_t)a.chrom_index) return;
  if (threadIdx.x == 0) s_nrec = 0;         // made visible by the first barrier inside scan_tile
  if (ti.flag != 0u) scan_tile<CHUNK, true>(a, tile, s_codes, s_tab, s_recs, &s_nrec);
  else scan_tile<CHUNK, false>(a, tile, s_codes, s_tab, s_recs, &s_nrec);
  // ---- flush the tile's records with one global atomic ----
  __syncthreads();
  const uint32_t n = min(s_nrec, (uint32_t)SCAN_STAGE);
  if (n == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(a.rec_count, n);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += LANES_PER_TILE) {
    const uint32_t g = s_base + i;
    if (g < a.rec_capacity) a.recs[g] = s_recs[i];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// align_kernel
// ------------------------------------------------------------------------------------------------------------------

constexpr int TR_UP = 0, TR_LEFT = 1, TR_DIAG = 2;   // ordered so that max() of (score*4 + code) breaks ties Diag > Left > Up
constexpr int NEG = -(1 << 20);                      // "minus infinity" that survives a few hundred additions
// A job (one scan record: its windows one after the other) takes LPJ lanes of a wave, lane r of them = query row r + 1.  LPJ = 32: two
// jobs per wave, guides up to 32 rows.  LPJ = 21: THREE jobs per wave (lanes 0-20, 21-41, 42-62; lane 63 idles) for guides of up to
// 20 rows -- every search with the usual 20-nt protospacer: 60 of 64 lanes hold a row instead of 40, a third fewer wave instructions
// for the same cells.  (Lane LPJ - 1 of a job is never a row then: it holds "row 0" for the job above it, see the fill.)
// One wave per workgroup: ~11 KB of LDS, which fits on a CU next to four scan workgroups (37 KB each of 160 KB) -- a 256-thread
// workgroup (40 KB) had to wait until the scan of the next range let go of a CU.
// per wave: flush threshold + the most one record iteration can add (jobs x 8 windows x 16 candidates; x 3 when every matrix
// of a cell is an alignment of its own)
constexpr int STAGE_FLUSH = 16;
template <bool PM, int LPJ> constexpr int ITEM_STAGE = STAGE_FLUSH + (PM ? 3 : 1) * (64 / LPJ) * 16;
constexpr int TB_LEN = STRIP_MAX_COLS + 48;          // strip columns + gap + PAM look-ahead
constexpr int TRACE_STAGE = 384;                     // RawAln records staged in LDS per trace_kernel workgroup
constexpr int TR_STRIDE = 100;                       // bytes per trace row (>= STRIP_MAX_COLS + 4, word aligned; lane r writes byte 99r + t)

__device__ __forceinline__ int shift_up_lane(int v) {
  // value of lane-1 (DPP wave shift right by one); lane 0 keeps its own value, which callers ignore
  return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ int comp_mask4(int m) {  // IUPAC set of the complementary base(s)
  return ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3);
}

// ------------------------------------------------------------------------------------------------------------------
// expand_kernel: scan records -> aligner jobs.
//
// Round 5.  align_kernel used to start every record with a chain of five dependent global loads (record -> tile -> contig -> window
// table base -> window bounds) and then fetched the strip's bases column by column: its waves stood in s_waitcnt half their time
// (profiles/r05_pmc_tail_before.txt: SQ_WAIT_ANY 51 % of SQ_WAVE_CYCLES at one or two waves per SIMD) with registers and LDS held
// all the while.  Here that chain runs ONCE per record with a lane per record -- 64 independent chains per wave, thousands of waves --,
// every (record, window) that holds candidate columns gets a number (one atomic per wave and round), and sixteen lanes per job write
// the head of the job's slab: geometry, the guide's row sets, and the strip's target masks decoded from the packed reference (two code
// words and two mask words per sixteen columns; run lookups for exception bases).  align_kernel then needs one 16-byte load per lane
// per job, from an address that depends on nothing but the job's number.
// ------------------------------------------------------------------------------------------------------------------
struct JobSeed {           // phase A -> phase B, through LDS: the job's header words and where its strip starts in the packed reference
  uint32_t slab;           // the job's slab: record x slots_per_rec + window slot
  uint32_t pad;
  uint32_t contig, window_k, n, c0;
  uint32_t cols;           // ncols | ntb << 16
  uint32_t what;           // dir | guide << 8 | true_border << 16 | L << 24
  uint32_t sel;            // the candidate columns inside the window (bits of the record's 16-base word)
  int32_t jbase;
  uint64_t gpos0;          // packed position of strip column c0 + 1 (tb[0]); the columns go up from there (dir 0) or down (dir 1)
};

__device__ __forceinline__ int tmask_at(const Run* runs, int64_t n_runs, uint64_t gpos, uint32_t code, uint32_t exc, int dir) {
  int m;
  if (!exc) {
    m = 1 << code;
  } else {
    int64_t r = run_floor(runs, n_runs, gpos);
    uint8_t ch = 0;
    if (r >= 0 && gpos < runs[r].start + runs[r].len) ch = runs[r].ch;
    m = target_mask(ch);
  }
  if (dir) m = (m & 16) | comp_mask4(m & 15);
  return m;
}

// Sixteen target masks (one 16-byte piece of a strip's tb[]): the bases at packed positions plo .. plo + nv - 1, in column order --
// ascending for the forward strand, descending and complemented for the reverse strand (d2).  cwl / cwh: the code words of plo and of
// plo + nv - 1 (the same word when the piece does not straddle), mwl / mwh likewise for the exception mask.
__device__ __forceinline__ uint4 decode_piece(const Run* runs, int64_t n_runs, uint32_t cwl, uint32_t cwh, uint32_t mwl, uint32_t mwh, uint64_t plo,
                                              int nv, int d2) {
  const uint64_t phi = plo + (uint64_t)(nv - 1);
  const uint64_t cw = ((uint64_t)cwh << 32) | (uint64_t)cwl, mw = ((uint64_t)mwh << 32) | (uint64_t)mwl;
  const uint32_t c32 = (plo >> 4) == (phi >> 4) ? (uint32_t)((uint32_t)cw >> ((plo & 15) * 2)) : (uint32_t)(cw >> ((plo & 15) * 2));
  const uint32_t m16 = ((plo >> 5) == (phi >> 5) ? (uint32_t)((uint32_t)mw >> (plo & 31)) : (uint32_t)(mw >> (plo & 31))) & ((1u << nv) - 1u);
  uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int ia = d2 ? nv - 1 - i : i;                          // ascending index of column i of the piece (negative beyond it: masked below)
    const uint32_t code = (c32 >> ((ia & 15) * 2)) & 3u;
    uint32_t tm = 1u << (d2 ? 3u - code : code);                 // plain base: its set, complemented for the reverse strand
    if (i >= nv) tm = 0u;
    if (i < 4) w0 |= tm << (i * 8); else if (i < 8) w1 |= tm << ((i - 4) * 8); else if (i < 12) w2 |= tm << ((i - 8) * 8); else w3 |= tm << ((i - 12) * 8);
  }
  // exception bases (N runs, IUPAC codes, padding: rare): their masks come from the run table
  for (uint32_t m = m16; m != 0u; m &= m - 1u) {
    const int ia = __ffs(m) - 1;
    const int tm = tmask_at(runs, n_runs, plo + (uint64_t)ia, 0u, 1u, d2);
    const int i = d2 ? nv - 1 - ia : ia;
    const uint32_t clr = ~(0xFFu << ((i & 3) * 8)), put = (uint32_t)tm << ((i & 3) * 8);
    if ((i >> 2) == 0) w0 = (w0 & clr) | put; else if ((i >> 2) == 1) w1 = (w1 & clr) | put; else if ((i >> 2) == 2) w2 = (w2 & clr) | put; else w3 = (w3 & clr) | put;
  }
  return make_uint4(w0, w1, w2, w3);
}

__global__ __launch_bounds__(256) void expand_kernel(AlignArgs a) {
  if (!a.low_prio) CALITAS_TAIL_PRIO();
  if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[0] = (unsigned long long)wall_clock64();   // (binned.hpp, BIN_BOX_STAMPS)
  __shared__ JobSeed s_seed[4][64];
  __shared__ int s_gint[MAX_GUIDES][4];                           // L, span, min_guide_score, cli_length
  __shared__ __attribute__((aligned(16))) uint8_t s_qmask[MAX_GUIDES][MAX_L];
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  for (int i = (int)threadIdx.x; i < a.sp.n_guides; i += (int)blockDim.x) {
    s_gint[i][0] = a.guides[i].L; s_gint[i][1] = a.guides[i].span; s_gint[i][2] = a.guides[i].min_guide_score; s_gint[i][3] = a.guides[i].cli_length;
  }
  for (int i = (int)threadIdx.x; i < a.sp.n_guides * (MAX_L / 4); i += (int)blockDim.x)
    reinterpret_cast<uint32_t*>(s_qmask[i / (MAX_L / 4)])[i % (MAX_L / 4)] = reinterpret_cast<const uint32_t*>(a.guides[i / (MAX_L / 4)].qmask)[i % (MAX_L / 4)];
  __syncthreads();
  uint32_t n_recs = *a.rec_count;
  if (n_recs > a.rec_capacity) n_recs = a.rec_capacity;
  const SearchDev& sp = a.sp;
  const int W = sp.window_size, step = sp.step;
  __shared__ uint32_t s_cnt[4], s_over;
  for (uint32_t blk = blockIdx.x * 256u; blk < n_recs; blk += gridDim.x * 256u) {   // (workgroup-uniform: there are barriers inside)
    // ---- phase A: a lane per record ----
    const uint32_t ri = blk + threadIdx.x;
    ScanRecord rec{0u, 0u};
    if (ri < n_recs) rec = a.recs[ri];
    const uint32_t cmask = rec.info & 0xFFFFu;
    const int dir = (int)((rec.info >> 16) & 1u), gi = (int)((rec.info >> 17) & 0x7Fu);
    uint32_t contig = 0;
    uint64_t gbase = 0, win_lo = 0, win_cnt = 0;
    int L = 0, span = 0, g_cli = 0;
    int64_t p0 = 0, khi = -1, klo = 0;
    if (cmask != 0u && gi < sp.n_guides) {
      L = s_gint[gi][0]; span = s_gint[gi][1]; g_cli = s_gint[gi][3];
      contig = a.tiles[rec.gword / a.tile_words].contig;
      gbase = a.contigs[contig].gbase;
      const uint64_t clen = a.contigs[contig].len;
      win_lo = a.win_base[contig]; win_cnt = a.win_base[contig + 1] - win_lo;
      p0 = (int64_t)((uint64_t)rec.gword * 16 - gbase);          // contig offset of bit 0
      const int64_t plo = p0 + (__ffs(cmask) - 1), phi = p0 + (31 - __clz(cmask));
      klo = (plo - W + 1 + step - 1) / step;                      // ceil((plo - W + 1) / step) for a positive numerator
      if (plo - W + 1 <= 0) klo = 0;
      khi = phi / step;
      if ((uint64_t)plo >= clen) khi = -1;                        // only padding columns: they belong to no window
    }
    // Rounds: round s looks at window klo + s of every record -- a record's candidate columns fall into at most slots_per_rec windows
    // (two with the usual tiling), so this is a loop with a trip count the whole grid shares, and the ballot below is taken in
    // straight-line code.  (A first version let every lane walk to its next window with candidate columns in a `while` with `break`s
    // and took the ballot behind it: the compiler kept the lanes that left the loop empty-handed apart from the others, their ballot
    // came out empty, they left -- and the jobs that phase B hands to THEIR lanes were never written: stale slab heads, at random.)
    for (int round = 0; round < (int)a.slots_per_rec; round++) {
      bool have = false;
      JobSeed seed{};
      const int64_t k = klo + round;
      int2 wab = make_int2(0, 0);
      const bool in_range = k <= khi && (uint64_t)k < win_cnt &&                 // (no such window on this contig: Range(0, len-1, step), SR:52)
                            win_lo + (uint64_t)k >= a.gw_lo && win_lo + (uint64_t)k < a.gw_hi;   // (outside this call's window range)
      if (in_range) wab = a.win[win_lo + (uint64_t)k];            // N-trimmed bounds, precomputed by window_table_kernel
      {
        const int64_t wa = wab.x, wb = wab.y;
        const int n = (int)(wb - wa);
        // candidate columns of this word that fall inside the window
        uint32_t sel = 0;
        for (int b = 0; b < 16; b++) if ((cmask >> b) & 1u) { const int64_t p = p0 + b; if (p >= wa && p < wb) sel |= 1u << b; }
        if (in_range && n >= g_cli && sel != 0u) {                // (n < g_cli: SearchReference.scala:536)
          const int sfirst = __ffs(sel) - 1, slast = 31 - __clz(sel);
          int jmin, jmax, jb;                                     // strand-space columns (1-based)
          if (dir == 0) { jmin = (int)(p0 + sfirst - wa) + 1; jmax = (int)(p0 + slast - wa) + 1; jb = (int)(p0 - wa) + 1; }
          else          { jmin = (int)(wb - (p0 + slast));    jmax = (int)(wb - (p0 + sfirst)); jb = (int)(wb - p0); }
          int c0 = jmin - span - 1;
          if (c0 < 0) c0 = 0;
          const int ncols = jmax - c0;                            // <= 16 + span + 1 <= STRIP_MAX_COLS (host-checked)
          int look = jmax + sp.max_gaps + MAX_PAM_LEN;            // PAM look-ahead, clipped to the window
          if (look > n) look = n;
          const int ntb = look - c0;                              // tb[x] = column c0 + 1 + x
          seed.slab = ri;                                         // (round 0; a later round's jobs get their slabs below)
          seed.contig = contig; seed.window_k = (uint32_t)k; seed.n = (uint32_t)n; seed.c0 = (uint32_t)c0;
          seed.cols = (uint32_t)ncols | ((uint32_t)ntb << 16);
          seed.what = (uint32_t)dir | ((uint32_t)gi << 8) | ((c0 == 0 ? 1u : 0u) << 16) | ((uint32_t)L << 24);
          seed.sel = sel; seed.jbase = jb;
          seed.gpos0 = gbase + (uint64_t)(dir ? wb - c0 - 1 : wa + c0);
          have = true;
        }
      }
      const unsigned long long bal = __ballot(have);
      const bool any_wide = __ballot(have && (seed.cols >> 16) > 64u) != 0ull;   // (both ballots in straight-line code, see above)

      // Round 0's job lives in slab `record`; a record without one says so there (ncols = 0: align_kernel looks).  The jobs of later
      // rounds -- a second window that holds the same columns: 3 % of the records -- are numbered behind the records' slabs
      // (rec_capacity + k) with ONE atomic per workgroup and round.  (Numbering all jobs with an atomic per wave and round was 3 800
      // returning atomics on one word per hg38-sized pass, 42 of the kernel's 60 us -- DESIGN.md 4.7; a fixed slab per later round
      // instead left align_kernel 120 000 empty slabs to look into, a memory round trip each: +30 us there.)
      if (round == 0 && ri < n_recs && !have) reinterpret_cast<uint32_t*>(a.slab + (uint64_t)ri * a.slab_bytes)[5] = 0u;
      const uint32_t nj = (uint32_t)__popcll(bal), rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (round != 0) {                                           // (workgroup-uniform)
        if (lane == 0) s_cnt[wave] = nj;
        __syncthreads();
        if (threadIdx.x == 0) {
          const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
          s_over = tot ? atomicAdd(a.job_count, tot) : 0u;
        }
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += s_cnt[w];
        seed.slab = a.rec_capacity + s_over + before + rank;     // (below rec_capacity x slots_per_rec: at most slots_per_rec - 1 later windows per record)
        __syncthreads();                                          // (s_cnt / s_over are rewritten in the next round)
      }
      if (bal == 0ull) continue;                                  // (wave-uniform)
      if (have) s_seed[wave][rank] = seed;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- phase B: four lanes per job write the head of its slab: lane p of the four decodes the target masks of columns 16 p ..
      //      16 p + 15 (a strip has 45-61 columns at d = 5) and writes its share of the header.  Sixteen jobs per pass, the passes unrolled
      //      with ALL their loads issued before the first is used: a wave's round is one trip to memory, not one per pass.  (Eight
      //      lanes per job and a loop of dependent passes took 40-60 us per launch, most of it waiting.)  Strips wider than 64
      //      columns (other limits, explicit targets) get their remaining pieces in a loop behind. ----
      {
        const int p4 = lane & 3;
        constexpr int PASSES = 4;
        uint32_t cwl[PASSES], cwh[PASSES], mwl[PASSES], mwh[PASSES];
#pragma unroll
        for (int it = 0; it < PASSES; it++) {
          const uint32_t q = (uint32_t)(it * 16 + (lane >> 2));
          cwl[it] = cwh[it] = mwl[it] = mwh[it] = 0u;
          if (q < nj) {
            const JobSeed& sd = s_seed[wave][q];
            const int ntb = (int)(sd.cols >> 16), d2 = (int)(sd.what & 1u), x0 = p4 * 16;
            if (x0 < ntb) {
              const int nv = min(16, ntb - x0);
              const uint64_t plo = d2 ? sd.gpos0 - (uint64_t)(x0 + nv - 1) : sd.gpos0 + (uint64_t)x0, phi = plo + (uint64_t)(nv - 1);
              cwl[it] = a.codes[plo >> 4]; cwh[it] = a.codes[phi >> 4]; mwl[it] = a.mask[plo >> 5]; mwh[it] = a.mask[phi >> 5];
            }
          }
        }
#pragma unroll
        for (int it = 0; it < PASSES; it++) {
          const uint32_t q = (uint32_t)(it * 16 + (lane >> 2));
          if (q < nj) {
            const JobSeed sd = s_seed[wave][q];
            const int ntb = (int)(sd.cols >> 16), ncols = (int)(sd.cols & 0xFFFFu), d2 = (int)(sd.what & 1u), g2 = (int)((sd.what >> 8) & 0xFFu), x0 = p4 * 16;
            uint8_t* slab = a.slab + (uint64_t)sd.slab * a.slab_bytes;
            if (x0 < ntb) {
              const int nv = min(16, ntb - x0);
              const uint64_t plo = d2 ? sd.gpos0 - (uint64_t)(x0 + nv - 1) : sd.gpos0 + (uint64_t)x0;
              *reinterpret_cast<uint4*>(slab + sizeof(SlabHeader) + x0) = decode_piece(a.runs, a.n_runs, cwl[it], cwh[it], mwl[it], mwh[it], plo, nv, d2);
            }
            uint4* head = reinterpret_cast<uint4*>(slab);
            if (p4 == 0) { head[0] = make_uint4(0u, sd.contig, sd.window_k, sd.n); head[6] = make_uint4((uint32_t)s_gint[g2][2], sd.sel, (uint32_t)sd.jbase, 0u); }
            else if (p4 == 1) head[1] = make_uint4(sd.c0, sd.cols, sd.what, (uint32_t)((ncols + 4) & ~3));
            else if (p4 == 2) head[4] = reinterpret_cast<const uint4*>(s_qmask[g2])[0];
            else head[5] = reinterpret_cast<const uint4*>(s_qmask[g2])[1];
          }
        }
        if (any_wide) {                                           // (wave-uniform) pieces 4 .. 8 of the wide strips: eight jobs per pass, a lane per piece
          for (uint32_t q0 = 0; q0 < nj; q0 += 8u) {
            const uint32_t q = q0 + (uint32_t)(lane >> 3);
            const int x0 = (4 + (lane & 7)) * 16;
            if (q < nj) {
              const JobSeed sd = s_seed[wave][q];
              const int ntb = (int)(sd.cols >> 16), d2 = (int)(sd.what & 1u);
              if (x0 < ntb) {
                const int nv = min(16, ntb - x0);
                const uint64_t plo = d2 ? sd.gpos0 - (uint64_t)(x0 + nv - 1) : sd.gpos0 + (uint64_t)x0, phi = plo + (uint64_t)(nv - 1);
                *reinterpret_cast<uint4*>(a.slab + (uint64_t)sd.slab * a.slab_bytes + sizeof(SlabHeader) + x0) =
                    decode_piece(a.runs, a.n_runs, a.codes[plo >> 4], a.codes[phi >> 4], a.mask[plo >> 5], a.mask[phi >> 5], plo, nv, d2);
              }
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// PM: the per-matrix reading of fgbio's enumeration.  A template parameter, not a run-time branch: its LDS
// (s_fin3, the larger item stage) would cost the default reading occupancy.
// A job's inputs are the head of its slab as expand_kernel wrote it: 128 bytes of header + TB_LEN bytes of target masks = 17 x 16
// bytes, lane x of the job's lanes loading piece x -- for the NEXT job while this one is being filled, so the load's latency hides
// behind the fill and the wave never waits on a chain of dependent loads.
constexpr int JOB_HEAD16 = (int)(sizeof(SlabHeader) + TB_LEN) / 16;
static_assert((sizeof(SlabHeader) + TB_LEN) % 16 == 0 && JOB_HEAD16 < 20, "one 16-byte piece of a job's head per lane of the job, and a lane for the second slot's header");

template <bool PM, int LPJ>
__global__ __launch_bounds__(64) void align_kernel(AlignArgs a) {
  if (!a.low_prio) CALITAS_TAIL_PRIO();
  static_assert(LPJ == 32 || LPJ == 21, "two or three jobs per wave");
  constexpr int JOBS = 64 / LPJ;            // jobs per wave (= per workgroup)
  constexpr int ROWS = LPJ == 32 ? MAX_L : LPJ - 1;   // rows a job can have
  constexpr int STAGE = ITEM_STAGE<PM, LPJ>;
  // trace rows are 100 bytes apart: lane r writes byte 99r + t at step t, which spreads the lanes of a job over the banks
  __shared__ __attribute__((aligned(16))) uint8_t s_tr[JOBS][ROWS][TR_STRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t s_hd[JOBS][sizeof(SlabHeader)];   // the job's header as it came
  __shared__ __attribute__((aligned(16))) uint8_t s_tbm[JOBS][TB_LEN];              // the bases a column matches (none for an N)
  __shared__ int s_fin[JOBS][STRIP_MAX_COLS + 1];
  __shared__ int s_fin3[PM ? JOBS : 1][3][PM ? STRIP_MAX_COLS + 1 : 1];   // per-matrix enumeration only: Diag / Left / Up of the bottom row
  // passing candidates are staged per wave and appended to a.items with one global atomic per flush: trace_kernel then
  // runs one lane per *passing* candidate instead of one per candidate slot (4 % of the slots pass at d = 5)
  __shared__ uint64_t s_items[1][STAGE];
  __shared__ uint32_t s_nitems[1];
  const int job = LPJ == 32 ? (int)(threadIdx.x >> 5) : (int)(threadIdx.x >= 21) + (int)(threadIdx.x >= 42) + (int)(threadIdx.x >= 63);
  const int r = (int)threadIdx.x - job * LPJ;   // lane within the job = query row r+1
  const int wave = 0, wlane = threadIdx.x & 63;
  if (wlane == 0) s_nitems[wave] = 0;
  __syncthreads();
  // all lanes of the wave that are still in the job loop call this together
  auto flush_items = [&](uint32_t threshold) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t n = s_nitems[wave];
    if (n >= threshold && n != 0) {
      const unsigned long long act = __ballot(1);
      const int leader = __ffsll((long long)act) - 1;
      uint32_t base = 0;
      if (wlane == leader) base = atomicAdd(a.item_count, n);
      base = __shfl(base, leader);
      const int rank = __popcll(act & ((1ull << wlane) - 1ull)), nact = __popcll(act);
      for (uint32_t i = (uint32_t)rank; i < n; i += (uint32_t)nact)
        if (base + i < a.item_capacity) a.items[base + i] = s_items[wave][i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (wlane == leader) s_nitems[wave] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };
  uint8_t (*tr)[TR_STRIDE] = s_tr[job < JOBS ? job : 0];
  uint8_t* tbm = s_tbm[job < JOBS ? job : 0];
  const uint32_t* hd32 = reinterpret_cast<const uint32_t*>(s_hd[job < JOBS ? job : 0]);
  int* fin = s_fin[job < JOBS ? job : 0];
  int (*fin3)[PM ? STRIP_MAX_COLS + 1 : 1] = s_fin3[PM ? (job < JOBS ? job : 0) : 0];

  uint32_t n_recs = *a.rec_count;
  if (n_recs > a.rec_capacity) n_recs = a.rec_capacity;
  const SearchDev& sp = a.sp;
  // the jobs: slab `record` for every scan record (its first window with candidate columns), then the slabs behind them
  // (rec_capacity + k: the second windows, numbered by expand_kernel)
  uint64_t n_over = *a.job_count;
  { const uint64_t room = (uint64_t)a.rec_capacity * (a.slots_per_rec - 1u); if (n_over > room) n_over = room; }
  const uint64_t n_virtual = (uint64_t)n_recs + n_over;
  auto slab_of = [&](uint64_t v) { return v < n_recs ? v : (uint64_t)a.rec_capacity + (v - n_recs); };

  const uint32_t total_jobs = gridDim.x * JOBS;
  // (lane 63 of a three-job wave belongs to no job)
  uint64_t vi = job < JOBS ? (uint64_t)blockIdx.x * JOBS + (uint64_t)job : n_virtual;
  uint4 pf = make_uint4(0u, 0u, 0u, 0u);                       // piece r of the head of the job's slab, loaded a job ahead
  if (vi < n_virtual && r < JOB_HEAD16) pf = reinterpret_cast<const uint4*>(a.slab + slab_of(vi) * a.slab_bytes)[r];
  for (; vi < n_virtual; vi += total_jobs) {
    {
    flush_items(STAGE_FLUSH);                // a job adds at most 16 candidates (x 3 per-matrix) per job of the wave
    // ---- stage the job's head in LDS: header as it is, target masks as "the bases this column matches" ----
    if (r < (int)(sizeof(SlabHeader) / 16)) {
      reinterpret_cast<uint4*>(s_hd[job])[r] = pf;
    } else if (r < JOB_HEAD16) {
      // tb byte: bits 0-3 IUPAC set, bit 4 forced mismatch (N) -> tbm byte: the set, or nothing when forced
      auto conv = [](uint32_t t) { const uint32_t f = (t >> 4) & 0x01010101u; return t & 0x0F0F0F0Fu & ~(f * 0xFFu); };
      reinterpret_cast<uint4*>(tbm)[r - (int)(sizeof(SlabHeader) / 16)] = make_uint4(conv(pf.x), conv(pf.y), conv(pf.z), conv(pf.w));
    }
    if (vi + total_jobs < n_virtual && r < JOB_HEAD16) pf = reinterpret_cast<const uint4*>(a.slab + slab_of(vi + total_jobs) * a.slab_bytes)[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if ((hd32[5] & 0xFFFFu) == 0u) continue;                  // (a record whose columns lie in no window of this call: no job in its slab)
    const uint64_t ji = slab_of(vi);                          // the job's slab
    {
      const int c0 = (int)hd32[4];
      const int ncols = (int)(hd32[5] & 0xFFFFu), ntb = (int)(hd32[5] >> 16);
      const int dir = (int)(hd32[6] & 0xFFu), L = (int)(hd32[6] >> 24);
      const bool true_border = ((hd32[6] >> 16) & 0xFFu) != 0u;
      const int g_min_score = (int)hd32[24];
      const uint32_t sel = hd32[25];
      const int jb = (int)hd32[26];
      const int qm = (r < L) ? (int)reinterpret_cast<const uint8_t*>(hd32 + 16)[r] : 0;
      (void)ntb;

      // ---- fill: antidiagonal wavefront, lane r = row r+1 ----
      // A cell of a matrix is kept as score * 4 + the matrix's code (TR_DIAG 2 > TR_LEFT 1 > TR_UP 0): the max of two cells breaks
      // ties the way fgbio does (Diag over Left over Up) and the code bits of the winner say which one it was -- one max where
      // there was a max, a compare and a select.
      const int i_row = r + 1;
      const int tgap4 = sp.target_gap * 4, qgap4 = sp.query_gap * 4;
      const int match_t = sp.match * 4 + TR_DIAG, mismatch_t = sp.mismatch * 4 + TR_DIAG;
      int curD = NEG * 4 + TR_DIAG, curL = NEG * 4 + TR_LEFT, curU = (true_border ? i_row * sp.target_gap : NEG) * 4 + TR_UP;
      // "Row 0" (score 0 in all three matrices, ties -> Diag) is what row 1 finds above it: lane 0 gets it as the `old` operand of the
      // lane shift; the first lane of a later job reads the last lane of the job before it, which is no row of that job when its guide
      // is shorter than LPJ and then simply holds row 0 -- always so with three jobs per wave (the host picks LPJ = 21 for guides of
      // up to 20 rows only).  (Two jobs, a 32-base guide in the first: lane 32 is patched in the loop.)
      const bool row0_in_lane31 = LPJ != 32 || __builtin_amdgcn_readlane(L, 0) < 32;
      if (row0_in_lane31 && r == LPJ - 1) { curD = TR_DIAG; curL = NEG * 4 + TR_LEFT; curU = TR_UP; }
      int curP = max(max(curD, curL), curU);
      const int t_first = r + 1, t_last = r < L ? r + ncols : -1;     // the steps at which this row has a column of the strip
      const int nsteps = ncols + L - 1;
      // what the lane above holds: refreshed by a lane shift per step.  Lane 0 has no lane above and keeps what is there -- row 0,
      // put there once; so does a lane whose upper neighbour is switched off (lane 32 when the first job has nothing to do).
      int inD = TR_DIAG, inU = TR_UP, inPa = TR_DIAG, inPb = TR_DIAG;
      int add_match = match_t, add_mismatch = mismatch_t;
      asm volatile("" : "+v"(add_match), "+v"(add_mismatch));        // in vector registers once, not re-materialised per step
      auto shift_in = [](int& dst, int src) { dst = __builtin_amdgcn_update_dpp(dst, src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); };
      // The column masks travel down the lanes with the wavefront: row r is at column t - r in step t, where row r - 1 was a step
      // earlier, so a row takes its mask from the lane above with the same lane shift that brings it the cells.  Round 5: before, every
      // lane read its column's mask from LDS, "two steps ahead" -- but LDS operations complete in order and s_waitcnt counts them in
      // order, so waiting for this step's mask also waited for the previous step's trace byte to land: an LDS round trip per step,
      // ~460 cycles per step at two waves per SIMD (profiles/r05_pmc_tail_before.txt: the waves parked 51 % of their time).  Now only
      // row 1 of each job reads masks, four at a time (one ds_read_b32 per four steps, issued a round ahead); nothing in a step waits.
      const uint32_t* tbm32 = reinterpret_cast<const uint32_t*>(tbm);
      const bool row1 = r == 0, bottom = r == L - 1;
      uint8_t* const trow = &tr[r < ROWS ? r : ROWS - 1][0];  // (a lane that is no row writes to column 0 of the last row)
      int m = 0;
      uint32_t w_cur = tbm32[0];
      asm volatile("" : "+v"(w_cur));                         // arrived before the loop: otherwise the loop's header waits for "all but the newest" every round
      auto cell = [&](const int t, const int jj, int inPp, auto patch_lane32) {
        shift_in(inD, curD);
        shift_in(inU, curU);
        if (decltype(patch_lane32)::value && threadIdx.x == 32) { inPp = TR_DIAG; inD = TR_DIAG; inU = TR_UP; }
        const int own = (int)((w_cur >> (8 * jj)) & 0xFFu);   // tbm[t - 1]: the column row 1 is at
        m = __builtin_amdgcn_update_dpp(own, m, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (row1) m = own;
        // The two LDS stores of a step stand OUTSIDE the branch, for every lane, every step: a lane that has no column of the strip in
        // this step writes to column 0 of its own trace row and to fin[0], which nothing reads.  With the stores inside the branch the
        // compiler cannot count the LDS operations between the read of the next round's masks and its use, assumes none, and waits for
        // the newest of them (s_waitcnt lgkmcnt(1)): an LDS round trip per round.
        const bool on = t >= t_first && t <= t_last;
        int c = 0, tbyte = 0;
        if (on) {
          c = t - r;                                          // strip column 1..ncols
          const int add_t = (qm & m) ? add_match : add_mismatch;
          const int newD = (inPp & ~3) + add_t;
          const int newU = max(inD, inU) + tgap4;             // code bits: TR_DIAG = from Diag, TR_UP = from Up
          const int newL = max(curD, curL) + qgap4;           // code bits: TR_DIAG = from Diag, TR_LEFT = from Left
          // trace byte: bits 0-1 where Diag came from, bit 2 Up came from Diag (else Up), bit 3 Left came from Left (else Diag)
          tbyte = (((newL & 1) << 3) | (inPp & 3)) | ((newU & 2) << 1);
          curD = newD; curU = newU & ~3; curL = (newL & ~3) | TR_LEFT;
          curP = max(max(curD, curL), curU);
        }
        trow[c] = (uint8_t)tbyte;
        const int fc = bottom ? c : 0;
        fin[fc] = curP;
        if (PM) { fin3[0][fc] = curD >> 2; fin3[1][fc] = curL >> 2; fin3[2][fc] = curU >> 2; }
      };
      auto fill = [&](auto patch_lane32) {
        shift_in(inPa, curP);
        // four steps per round (steps past the last one find no row on the strip): the shifted curP of one step is the "previous" of
        // the step after the next, hence the two registers taking turns
        for (int t = 1; t <= nsteps; t += 4) {
          const uint32_t w_next = tbm32[(t + 3) >> 2];       // the masks of the next round's columns
          shift_in(inPb, curP);
          cell(t, 0, inPa, patch_lane32);
          shift_in(inPa, curP);
          cell(t + 1, 1, inPb, patch_lane32);
          shift_in(inPb, curP);
          cell(t + 2, 2, inPa, patch_lane32);
          shift_in(inPa, curP);
          cell(t + 3, 3, inPb, patch_lane32);
          w_cur = w_next;
        }
      };
      if (LPJ != 32 || row0_in_lane31) fill(std::false_type{}); else fill(std::true_type{});
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- hand the strip over to trace_kernel: one candidate descriptor per passing end column, plus the strip's
      //      trace matrix copied from LDS into the job's slab (its target masks are there already) ----
      // lane x takes the x-th selected bit in ascending strand-space column order
      int myb = -1;
      {
        const int sfirst = __ffs(sel) - 1, slast = 31 - __clz(sel);
        int cnt = 0;
        if (dir == 0) { for (int b = sfirst; b <= slast; b++) if ((sel >> b) & 1u) { if (cnt == r) myb = b; cnt++; } }
        else          { for (int b = slast; b >= sfirst; b--) if ((sel >> b) & 1u) { if (cnt == r) myb = b; cnt++; } }
      }
      int j = 0, P = 0;
      int pm_score[3] = {0, 0, 0};
      uint32_t pm_pass = 0;                                            // per-matrix enumeration: bit k = matrix k (Diag, Left, Up) passes
      bool pass = false;
      if (myb >= 0) {
        j = dir ? jb - myb : jb + myb;                                 // strand-space end column
        P = fin[j - c0];
        if (PM) {                                                      // every bottom-row cell >= minScore is an alignment of its own
#pragma unroll
          for (int k3 = 0; k3 < 3; k3++) { pm_score[k3] = fin3[k3][j - c0]; if (pm_score[k3] >= g_min_score) pm_pass |= 1u << k3; }
          pass = pm_pass != 0;
        } else {
          pass = (P >> 2) >= g_min_score;                              // best of the three matrices >= minScore
        }
      }
      const unsigned long long bal = __ballot(pass);
      const uint32_t mine = (uint32_t)(bal >> (job * LPJ)) & 0xFFFFu;  // this job's lanes (only lanes 0..15 can pass)
      if (mine != 0u) {
        uint8_t* slab = a.slab + ji * a.slab_bytes;
        SlabHeader* hd = reinterpret_cast<SlabHeader*>(slab);
        const int stride = (ncols + 4) & ~3;                           // bytes per trace row in the slab (columns 0..ncols)
        const uint32_t tb_bytes = (uint32_t)((ntb + 3) & ~3);
        if (r == 0) hd->pass_mask = mine;
        if (pass) {
          hd->j[r] = (uint16_t)j;
          // item = candidate slot | slab index << 4 | start matrix << 40 | (score + 2^21) << 42
          const uint64_t where = ((ji & 0xFFFFFFFFFull) << 4) | (uint64_t)r;
          if (PM) {
            constexpr int code[3] = {TR_DIAG, TR_LEFT, TR_UP};         // fgbio's order of directions
#pragma unroll
            for (int k3 = 0; k3 < 3; k3++) if ((pm_pass >> k3) & 1u) {
              const uint32_t slot = atomicAdd(&s_nitems[wave], 1u);
              if (slot < (uint32_t)STAGE)
                s_items[wave][slot] = where | ((uint64_t)code[k3] << 40) | ((uint64_t)(uint32_t)(pm_score[k3] + (1 << 21)) << 42);
            }
          } else {
            const uint32_t slot = atomicAdd(&s_nitems[wave], 1u);
            if (slot < (uint32_t)STAGE) s_items[wave][slot] = where | ((uint64_t)(P & 3) << 40) | ((uint64_t)(uint32_t)((P >> 2) + (1 << 21)) << 42);
          }
        }
        if (r < L) {
          uint32_t* drow = reinterpret_cast<uint32_t*>(slab + sizeof(SlabHeader) + tb_bytes + (uint32_t)(r * stride));
          const uint32_t* srow = reinterpret_cast<const uint32_t*>(&tr[r][0]);
          for (int x = 0; x < stride / 4; x++) drow[x] = srow[x];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    }
  }
  flush_items(1);
}

// ------------------------------------------------------------------------------------------------------------------
// align_pk_kernel: align_kernel with TWO jobs in every lane group, one in each 16-bit half of the lanes' registers.
//
// Round 5.  Alone on the chip align_kernel's waves execute vector instructions 59 % of their time at two waves per SIMD
// (profiles/r05_pmc_align_alone_*.txt): with the prologue's dependent loads and the per-step LDS waits gone, what is left is its
// instruction count -- ~30 vector instructions per antidiagonal step, 7.8e7 per hg38-sized pass.  A cell is score x 4 + matrix code;
// with the reference's costs |score| <= 60 x 20, so a cell fits sixteen bits with room to spare, and v_pk_add_i16 / v_pk_max_i16 /
// v_pk_mad_i16 do two cells per instruction: the same ~31 instructions per step now fill the strips of SIX jobs per wave (three lane
// groups of 21 x two halves).  "Minus infinity" is the bottom of the range and the additions saturate (clamp), so a cell that is
// out of range stays below every cell a passing alignment can go through (those lie within +-4 x max|cost| x L of zero, which the
// host checks fits: AlignArgs::pack16); their code bits are lost, which no traceback can see.  The trace nibbles of the two jobs
// share a byte (low nibble: the even job); each job's slab gets the rows with its header saying which nibble is its own.
// Launched instead of align_kernel<false, 21> when the host says so; anything else (longer guides, per-matrix enumeration, costs out
// of range, explicit targets) takes align_kernel.
// ------------------------------------------------------------------------------------------------------------------
typedef short pk2 __attribute__((ext_vector_type(2)));
typedef unsigned short upk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_add_sat(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_add_sat(__builtin_bit_cast(pk2, a), __builtin_bit_cast(pk2, b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(pk2, a), __builtin_bit_cast(pk2, b))); }
__device__ __forceinline__ int pk_min_u(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(upk2, a), __builtin_bit_cast(upk2, b))); }
__device__ __forceinline__ int pk_mad(int a, int b, int c) { return __builtin_bit_cast(int, (pk2)(__builtin_bit_cast(pk2, a) * __builtin_bit_cast(pk2, b) + __builtin_bit_cast(pk2, c))); }
__device__ __forceinline__ int pk_rep(int x) { return (int)(((uint32_t)x & 0xFFFFu) * 0x00010001u); }
__device__ __forceinline__ int pk_half(int x, int h) { return h ? (x >> 16) : (int)(short)(x & 0xFFFF); }

__global__ __launch_bounds__(64) void align_pk_kernel(AlignArgs a) {
  if (!a.low_prio) CALITAS_TAIL_PRIO();
  constexpr int LPJ = 21, GROUPS = 3, JOBS = 6, ROWS = LPJ - 1;
  constexpr int STAGE = STAGE_FLUSH + JOBS * 16;
  constexpr int NEG4 = -32768;                                 // "minus infinity" x 4 in sixteen bits (the additions saturate)
  __shared__ __attribute__((aligned(16))) uint8_t s_tr[GROUPS][ROWS][TR_STRIDE];    // low nibble: the group's even job, high nibble: the odd one
  __shared__ __attribute__((aligned(16))) uint8_t s_hd[JOBS][sizeof(SlabHeader)];
  __shared__ __attribute__((aligned(16))) uint8_t s_tbm[JOBS][TB_LEN];
  __shared__ int s_fin[GROUPS][STRIP_MAX_COLS + 1];             // the bottom row's best of three, both jobs' halves
  __shared__ uint64_t s_items[STAGE];
  __shared__ uint32_t s_nitems;
  const int grp = (int)(threadIdx.x >= 21) + (int)(threadIdx.x >= 42) + (int)(threadIdx.x >= 63);
  const int r = (int)threadIdx.x - grp * LPJ;
  const int wlane = threadIdx.x & 63;
  if (wlane == 0) s_nitems = 0;
  __syncthreads();
  auto flush_items = [&](uint32_t threshold) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t n = s_nitems;
    if (n >= threshold && n != 0) {
      const unsigned long long act = __ballot(1);
      const int leader = __ffsll((long long)act) - 1;
      uint32_t base = 0;
      if (wlane == leader) base = atomicAdd(a.item_count, n);
      base = __shfl(base, leader);
      const int rank = __popcll(act & ((1ull << wlane) - 1ull)), nact = __popcll(act);
      for (uint32_t i = (uint32_t)rank; i < n; i += (uint32_t)nact)
        if (base + i < a.item_capacity) a.items[base + i] = s_items[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (wlane == leader) s_nitems = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };
  const int g = grp < GROUPS ? grp : 0;
  uint8_t (*tr)[TR_STRIDE] = s_tr[g];
  const uint32_t* hdA = reinterpret_cast<const uint32_t*>(s_hd[2 * g]);
  const uint32_t* hdB = reinterpret_cast<const uint32_t*>(s_hd[2 * g + 1]);
  int* fin = s_fin[g];

  uint32_t n_recs = *a.rec_count;
  if (n_recs > a.rec_capacity) n_recs = a.rec_capacity;
  const SearchDev& sp = a.sp;
  uint64_t n_over = *a.job_count;
  { const uint64_t room = (uint64_t)a.rec_capacity * (a.slots_per_rec - 1u); if (n_over > room) n_over = room; }
  const uint64_t n_virtual = (uint64_t)n_recs + n_over;
  auto slab_of = [&](uint64_t v) { return v < n_recs ? v : (uint64_t)a.rec_capacity + (v - n_recs); };

  const uint32_t total_jobs = gridDim.x * JOBS;
  // the group's pair of jobs: virtual jobs vi (even half) and vi + 1 (odd half)
  uint64_t vi = grp < GROUPS ? ((uint64_t)blockIdx.x * GROUPS + (uint64_t)grp) * 2 : n_virtual;
  uint4 pfA = make_uint4(0u, 0u, 0u, 0u), pfB = make_uint4(0u, 0u, 0u, 0u);
  auto prefetch = [&](uint64_t v) {
    if (r < JOB_HEAD16) {
      pfA = reinterpret_cast<const uint4*>(a.slab + slab_of(v) * a.slab_bytes)[r];
      if (v + 1 < n_virtual) pfB = reinterpret_cast<const uint4*>(a.slab + slab_of(v + 1) * a.slab_bytes)[r];
      else pfB = make_uint4(0u, 0u, 0u, 0u);                    // (no odd job: a header of zeros says "no job")
    }
  };
  if (vi < n_virtual) prefetch(vi);
  for (; vi < n_virtual; vi += total_jobs) {
    flush_items(STAGE_FLUSH);
    // ---- stage the two heads in LDS ----
    auto conv = [](uint32_t t) { const uint32_t f = (t >> 4) & 0x01010101u; return t & 0x0F0F0F0Fu & ~(f * 0xFFu); };
    if (r < (int)(sizeof(SlabHeader) / 16)) {
      reinterpret_cast<uint4*>(s_hd[2 * g])[r] = pfA;
      reinterpret_cast<uint4*>(s_hd[2 * g + 1])[r] = pfB;
    } else if (r < JOB_HEAD16) {
      reinterpret_cast<uint4*>(s_tbm[2 * g])[r - (int)(sizeof(SlabHeader) / 16)] = make_uint4(conv(pfA.x), conv(pfA.y), conv(pfA.z), conv(pfA.w));
      reinterpret_cast<uint4*>(s_tbm[2 * g + 1])[r - (int)(sizeof(SlabHeader) / 16)] = make_uint4(conv(pfB.x), conv(pfB.y), conv(pfB.z), conv(pfB.w));
    }
    if (vi + total_jobs < n_virtual) prefetch(vi + total_jobs);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ncolsA = (int)(hdA[5] & 0xFFFFu), ncolsB = (int)(hdB[5] & 0xFFFFu);
    if (ncolsA == 0 && ncolsB == 0) continue;                   // (records whose columns lie in no window of this call)
    {
      const int L = (int)((ncolsA ? hdA[6] : hdB[6]) >> 24);    // (one protospacer length per launch: the host checked)
      const int ncols = max(ncolsA, ncolsB);
      const bool tbdA = ((hdA[6] >> 16) & 0xFFu) != 0u, tbdB = ((hdB[6] >> 16) & 0xFFu) != 0u;
      const int qm = (r < L) ? ((int)reinterpret_cast<const uint8_t*>(hdA + 16)[r] | ((int)reinterpret_cast<const uint8_t*>(hdB + 16)[r] << 16)) : 0;

      // ---- fill: align_kernel's, two cells per register ----
      const int i_row = r + 1;
      const int tgap4 = pk_rep(sp.target_gap * 4), qgap4 = pk_rep(sp.query_gap * 4);
      const int mism_t = pk_rep(sp.mismatch * 4 + TR_DIAG), delta_t = pk_rep((sp.match - sp.mismatch) * 4);
      const int one2 = 0x00010001, keep2 = (int)0xFFFCFFFC;
      int curD = pk_rep(NEG4 + TR_DIAG), curL = pk_rep(NEG4 + TR_LEFT);
      int curU = (int)(((uint32_t)((tbdA ? i_row * sp.target_gap * 4 : NEG4) + TR_UP) & 0xFFFFu) | ((uint32_t)((tbdB ? i_row * sp.target_gap * 4 : NEG4) + TR_UP) << 16));
      if (r == LPJ - 1) { curD = pk_rep(TR_DIAG); curL = pk_rep(NEG4 + TR_LEFT); curU = pk_rep(TR_UP); }   // "row 0" for the group above
      int curP = pk_max(pk_max(curD, curL), curU);
      const int t_first = r + 1, t_last = r < L ? r + ncols : -1;
      const int nsteps = ncols + L - 1;
      int inD = pk_rep(TR_DIAG), inU = pk_rep(TR_UP), inPa = pk_rep(TR_DIAG), inPb = pk_rep(TR_DIAG);
      auto shift_in = [](int& dst, int src) { dst = __builtin_amdgcn_update_dpp(dst, src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); };
      const uint32_t* tbmA = reinterpret_cast<const uint32_t*>(s_tbm[2 * g]);
      const uint32_t* tbmB = reinterpret_cast<const uint32_t*>(s_tbm[2 * g + 1]);
      const bool row1 = r == 0, bottom = r == L - 1;
      uint8_t* const trow = &tr[r < ROWS ? r : ROWS - 1][0];
      int m = 0;
      uint32_t wA = tbmA[0], wB = tbmB[0];
      asm volatile("" : "+v"(wA), "+v"(wB));
      auto cell = [&](const int t, const int jj, int inPp) {
        shift_in(inD, curD);
        shift_in(inU, curU);
        // the two jobs' masks of the column row 1 is at: byte jj of wA in the low half, of wB in the high half
        const int own = (int)__builtin_amdgcn_perm(wB, wA, 0x0C000C00u | (uint32_t)jj | ((uint32_t)(4 + jj) << 16));
        m = __builtin_amdgcn_update_dpp(own, m, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (row1) m = own;
        const bool on = t >= t_first && t <= t_last;
        int c = 0, tbyte = 0;
        if (on) {
          c = t - r;
          const int hit = pk_min_u(qm & m, one2);             // 1 per half where the row's set meets the column's
          const int add_t = pk_mad(hit, delta_t, mism_t);
          const int newD = pk_add_sat(inPp & keep2, add_t);
          const int newU = pk_add_sat(pk_max(inD, inU), tgap4);
          const int newL = pk_add_sat(pk_max(curD, curL), qgap4);
          // trace nibble per half: bits 0-1 where Diag came from, bit 2 Up came from Diag, bit 3 Left came from Left
          const int tn = (inPp & 0x00030003) | ((newU & 0x00020002) << 1) | ((newL & one2) << 3);
          tbyte = tn | (tn >> 12);                            // low nibble: the even job, high nibble: the odd one
          curD = newD; curU = newU & keep2; curL = (newL & keep2) | one2;
          curP = pk_max(pk_max(curD, curL), curU);
        }
        trow[c] = (uint8_t)tbyte;
        fin[bottom ? c : 0] = curP;
      };
      shift_in(inPa, curP);
      for (int t = 1; t <= nsteps; t += 4) {
        const uint32_t nA = tbmA[(t + 3) >> 2], nB = tbmB[(t + 3) >> 2];
        shift_in(inPb, curP);
        cell(t, 0, inPa);
        shift_in(inPa, curP);
        cell(t + 1, 1, inPb);
        shift_in(inPb, curP);
        cell(t + 2, 2, inPa);
        shift_in(inPa, curP);
        cell(t + 3, 3, inPb);
        wA = nA; wB = nB;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // ---- hand over, one job after the other ----
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t* hd32 = h ? hdB : hdA;
        const int ncols_h = h ? ncolsB : ncolsA;
        const int c0 = (int)hd32[4], ntb = (int)(hd32[5] >> 16), dir = (int)(hd32[6] & 0xFFu);
        const int g_min_score = (int)hd32[24];
        const uint32_t sel = ncols_h ? hd32[25] : 0u;
        const int jb = (int)hd32[26];
        int myb = -1;
        {
          const int sfirst = __ffs(sel) - 1, slast = 31 - __clz(sel);
          int cnt = 0;
          if (dir == 0) { for (int b = sfirst; b <= slast; b++) if ((sel >> b) & 1u) { if (cnt == r) myb = b; cnt++; } }
          else          { for (int b = slast; b >= sfirst; b--) if ((sel >> b) & 1u) { if (cnt == r) myb = b; cnt++; } }
        }
        int j = 0, P = 0;
        bool pass = false;
        if (myb >= 0) {
          j = dir ? jb - myb : jb + myb;
          P = pk_half(fin[j - c0], h);
          pass = (P >> 2) >= g_min_score;
        }
        const unsigned long long bal = __ballot(pass);
        const uint32_t mine = (uint32_t)(bal >> (grp * LPJ)) & 0xFFFFu;
        if (mine != 0u) {
          const uint64_t ji = slab_of(vi + (uint64_t)h);
          uint8_t* slab = a.slab + ji * a.slab_bytes;
          SlabHeader* hd = reinterpret_cast<SlabHeader*>(slab);
          const int stride = (ncols + 4) & ~3;                  // the PAIR's row length: both jobs' rows are the group's rows
          const uint32_t tb_bytes = (uint32_t)((ntb + 3) & ~3);
          if (r == 0) { hd->pass_mask = mine; hd->stride = (uint16_t)stride; hd->pad = (uint16_t)h; }
          if (pass) {
            hd->j[r] = (uint16_t)j;
            const uint64_t where = ((ji & 0xFFFFFFFFFull) << 4) | (uint64_t)r;
            const uint32_t slot = atomicAdd(&s_nitems, 1u);
            if (slot < (uint32_t)STAGE) s_items[slot] = where | ((uint64_t)(P & 3) << 40) | ((uint64_t)(uint32_t)((P >> 2) + (1 << 21)) << 42);
          }
          if (r < L) {
            uint32_t* drow = reinterpret_cast<uint32_t*>(slab + sizeof(SlabHeader) + tb_bytes + (uint32_t)(r * stride));
            const uint32_t* srow = reinterpret_cast<const uint32_t*>(&tr[r][0]);
            for (int x = 0; x < stride / 4; x++) drow[x] = srow[x];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  flush_items(1);
}

// Traceback + PAM extension of one candidate end column (item = (record x window slot) slab index << 4 | candidate slot).
template <typename Emit>
__device__ __forceinline__ void trace_one(const AlignArgs& a, const SearchDev& sp, uint64_t it, const uint8_t (*s_qmask)[MAX_L],
                                          const uint8_t (*s_pam)[MAX_PAMS][MAX_PAM_LEN], const uint8_t (*s_pamlen)[MAX_PAMS],
                                          const int (*s_gint)[4], uint32_t* s_ncand, Emit& emit) {
  {
    const int x = (int)(it & 15);
    const uint8_t* slab = a.slab + ((it >> 4) & 0xFFFFFFFFFull) * a.slab_bytes;
    const SlabHeader* hd = reinterpret_cast<const SlabHeader*>(slab);
    if (!((hd->pass_mask >> x) & 1u)) return;
    const uint8_t* tb = slab + sizeof(SlabHeader);
    const uint8_t* tr = tb + ((hd->ntb + 3) & ~3);
    const int L = hd->L, c0 = hd->c0, n = hd->n, gi = hd->guide, stride = hd->stride, nib4 = hd->pad ? 4 : 0;
    const bool true_border = hd->true_border != 0;
    const int j = hd->j[x], gscore = (int)(uint32_t)(it >> 42) - (1 << 21);   // score and start matrix travel in the item
    const int g_npams = s_gint[gi][0], g_maxd = s_gint[gi][1], g_maxp = s_gint[gi][2], g_maxf = s_gint[gi][3];
    atomicAdd(s_ncand, 1u);

    int m = (int)((it >> 40) & 3u), i = L, c = j - c0;
    const int m_start = m;
    uint32_t ops[RAW_MAX_OPS / 16] = {0, 0, 0, 0, 0};
    int nops = 0, diffs = 0;
    bool ok = true;
    while (i > 0) {
      if (nops >= RAW_MAX_OPS) { ok = false; break; }
      int op;
      if (c == 0) {
        // true left border: only the Up matrix is finite there (leading insertions)
        if (!true_border || m != TR_UP) { ok = false; break; }
        op = 2; i--;                                   // 'I'; Up(i,0) traces to Up, Up(1,0) to Diag(0,0)
      } else {
        const int t8 = (tr[(i - 1) * stride + c] >> nib4) & 15;   // (align_pk_kernel: two jobs' trace nibbles share a byte)
        if (m == TR_DIAG) {
          const int tm = tb[c - 1], q = s_qmask[gi][i - 1];
          const bool compat = (q & tm & 15) != 0;
          const bool eq = sp.eqx_by_score ? (compat && !(tm & 16)) : compat;
          op = eq ? 0 : 1;
          m = t8 & 3; i--; c--;
        } else if (m == TR_UP) {
          op = 2; m = ((t8 >> 2) & 1) ? TR_DIAG : TR_UP; i--;       // bit 2: Up came from Diag
        } else {
          op = 3; m = ((t8 >> 3) & 1) ? TR_LEFT : TR_DIAG; c--;
        }
      }
      if (op != 0) diffs++;
      ops[nops >> 4] |= (uint32_t)op << ((nops & 15) * 2);
      nops++;
    }
    if (!ok) { atomicAdd(a.anomalies, 1u); return; }
    if (diffs > g_maxd) return;
    RawAln o;
    o.contig = hd->contig; o.window_k = hd->window_k; o.t_start = (uint16_t)(c0 + c + 1); o.t_end_guide = (uint16_t)j;
    o.dir = hd->dir; o.guide = hd->guide; o.n_ops = (uint8_t)nops;
    o.pad = sp.per_matrix ? (uint8_t)(m_start == TR_DIAG ? 0 : m_start == TR_LEFT ? 1 : 2) : (uint8_t)0;
    uint32_t* ow = reinterpret_cast<uint32_t*>(o.ops);
#pragma unroll
    for (int w = 0; w < RAW_MAX_OPS / 16; w++) ow[w] = ops[w];
    if (g_npams == 0) {
      o.score = gscore; o.pam = -1; o.offset = 0; o.pam_x = 0;
      emit(o);
      return;
    }
    // terminal indel run = first ops of the traceback
    int term = 0;
    {
      const int op0 = ops[0] & 3;
      if (op0 >= 2) { term = 1; while (term < nops && (int)((ops[term >> 4] >> ((term & 15) * 2)) & 3) == op0) term++; }
    }
    int max_extra = sp.max_gaps - term;
    if (g_maxf - diffs < max_extra) max_extra = g_maxf - diffs;
    for (int pi = 0; pi < g_npams; pi++) {
      const int plen = s_pamlen[gi][pi];
      bool have = false; int best_score = 0, best_off = 0; uint32_t best_x = 0;
      for (int off = 0; off <= max_extra; off++) {
        const int toff = j + off;                   // 0-based strand-space offset of the first PAM base
        int limit = g_maxp;
        if (g_maxf - diffs - off < limit) limit = g_maxf - diffs - off;
        if (toff + plen > n || limit < 0) continue;
        int sc = 0, nx = 0; uint32_t xm = 0;
        for (int q = 0; q < plen; q++) {
          const int tm = tb[toff + q - c0];
          const bool match = ((s_pam[gi][pi][q] & tm & 15) != 0) && !(tm & 16);
          const int addend = match ? sp.pam_match : sp.pam_mismatch;
          sc += addend;
          if (!(addend > 0)) { nx++; xm |= 1u << q; }
        }
        if (nx > limit) continue;
        const int total = gscore + sc + off * sp.query_gap;
        if (!have || total > best_score) { have = true; best_score = total; best_off = off; best_x = xm; }
      }
      if (have) {
        o.score = best_score; o.pam = (int8_t)pi; o.offset = (uint8_t)best_off; o.pam_x = (uint16_t)best_x;
        emit(o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// trace_kernel: one lane per passing candidate end column -- traceback through the strip's trace matrix (slab in HBM),
// '='/'X' ops, extendAndFilterRight (SequentialGuideAligner.scala:433-492), one RawAln per (candidate, PAM).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trace_kernel(AlignArgs a, uint32_t* box, uint32_t seq) {
  if (!a.low_prio) CALITAS_TAIL_PRIO();
  __shared__ uint8_t s_qmask[MAX_GUIDES][MAX_L];
  __shared__ uint8_t s_pam[MAX_GUIDES][MAX_PAMS][MAX_PAM_LEN];
  __shared__ uint8_t s_pamlen[MAX_GUIDES][MAX_PAMS];
  __shared__ int s_gint[MAX_GUIDES][4];     // n_pams, max_guide_diffs, max_pam_mismatches, max_diffs_filtering
  for (int i = threadIdx.x; i < a.sp.n_guides * MAX_L; i += blockDim.x) s_qmask[i / MAX_L][i % MAX_L] = a.guides[i / MAX_L].qmask[i % MAX_L];
  for (int i = threadIdx.x; i < a.sp.n_guides * MAX_PAMS * MAX_PAM_LEN; i += blockDim.x) {
    const int gi = i / (MAX_PAMS * MAX_PAM_LEN), rem = i % (MAX_PAMS * MAX_PAM_LEN);
    s_pam[gi][rem / MAX_PAM_LEN][rem % MAX_PAM_LEN] = a.guides[gi].pam_mask[rem / MAX_PAM_LEN][rem % MAX_PAM_LEN];
  }
  for (int i = threadIdx.x; i < a.sp.n_guides * MAX_PAMS; i += blockDim.x) s_pamlen[i / MAX_PAMS][i % MAX_PAMS] = a.guides[i / MAX_PAMS].pam_len[i % MAX_PAMS];
  for (int i = threadIdx.x; i < a.sp.n_guides; i += blockDim.x) {
    s_gint[i][0] = a.guides[i].n_pams; s_gint[i][1] = a.guides[i].max_guide_diffs;
    s_gint[i][2] = a.guides[i].max_pam_mismatches; s_gint[i][3] = a.guides[i].max_diffs_filtering;
  }
  __syncthreads();

  __shared__ RawAln s_out[TRACE_STAGE];
  __shared__ uint32_t s_nout, s_obase, s_ncand;
  if (threadIdx.x == 0) { s_nout = 0; s_ncand = 0; }
  __syncthreads();

  uint32_t n_items = *a.item_count;                                   // passing candidates appended by align_kernel
  if (n_items > a.item_capacity) n_items = a.item_capacity;
  const SearchDev& sp = a.sp;
  // Results are staged in LDS and flushed with one global atomic per flush (see stage_record above for why).
  // ... and where an alignment lands in a.out[] is also listed in the bin its window starts in (binned.hip; returning atomics on
  // distinct words: cheap)
  auto to_bin = [&](uint32_t contig, uint32_t window_k, uint32_t g) {
    const uint32_t bin = a.bin_base[contig] + (uint32_t)(((uint64_t)window_k * (uint64_t)(uint32_t)sp.step) >> a.bin_shift) - a.bin_first;
    if (bin >= a.bin_n) { atomicAdd(a.anomalies, 1u); return; }      // (the host plans the windows from the bins: an internal error, reported)
    const uint32_t at = atomicAdd(a.bin_count + bin, 1u);
    if (at < a.bin_cap) a.bin_idx[(size_t)bin * a.bin_cap + at] = g;
  };
  auto emit = [&](const RawAln& o) {
    const uint32_t slot = atomicAdd(&s_nout, 1u);                     // LDS atomic
    if (slot < (uint32_t)TRACE_STAGE) { s_out[slot] = o; return; }
    const uint32_t g = atomicAdd(a.out_count, 1u);                    // stage full: append directly
    if (g < a.out_capacity) { a.out[g] = o; if (a.bin_idx) to_bin(o.contig, o.window_k, g); }
  };
  auto flush = [&]() {                                                // block-uniform call sites only
    __syncthreads();
    const uint32_t n = min(s_nout, (uint32_t)TRACE_STAGE);
    if (n != 0) {
      if (threadIdx.x == 0) s_obase = atomicAdd(a.out_count, n);
      __syncthreads();
      const uint32_t* src = reinterpret_cast<const uint32_t*>(s_out);
      constexpr uint32_t WPR = sizeof(RawAln) / 4;
      for (uint32_t w = threadIdx.x; w < n * WPR; w += blockDim.x) {
        const uint32_t g = s_obase + w / WPR;
        if (g < a.out_capacity) reinterpret_cast<uint32_t*>(a.out + g)[w % WPR] = src[w];
      }
      if (a.bin_idx)
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x)
          if (s_obase + k < a.out_capacity) to_bin(s_out[k].contig, s_out[k].window_k, s_obase + k);
      __syncthreads();
      if (threadIdx.x == 0) s_nout = 0;
      __syncthreads();
    }
  };

  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < n_items; base += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t idx = base + threadIdx.x;
    if (idx < n_items) trace_one(a, sp, a.items[idx], s_qmask, s_pam, s_pamlen, s_gint, &s_ncand, emit);
    __syncthreads();
    const uint32_t staged = s_nout;     // same value in every thread: nobody appends between the two barriers
    __syncthreads();
    if (staged > (uint32_t)(TRACE_STAGE / 2)) flush();
  }
  flush();
  if (threadIdx.x == 0 && s_ncand) atomicAdd(a.cand_count, s_ncand);
  if (box) {
    // the last workgroup to get here posts the call's counters (records, alignments, anomalies, candidates, ...) to the host: no
    // launch of its own for that on the path the host waits on
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(a.trace_done, 1u) == gridDim.x - 1) {
        for (int i = 0; i < 8; i++) box[1 + i] = __hip_atomic_load(a.rec_count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// Window table of windowIterator (SearchReference.scala:39-71) for one (window size, step): out[win_base[c] + k] = N-trimmed
// 0-based half-open bounds of window k of contig c.  Rebuilt only when the tiling changes.
__global__ void window_table_kernel(const Run* runs, int64_t n_runs, const ContigInfo* contigs, const uint64_t* win_base,
                                    int n_contigs, int W, int step, int2* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= win_base[n_contigs]) return;
  int lo = 0, hi = n_contigs;              // last contig with win_base[c] <= i
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (win_base[mid] <= i) lo = mid; else hi = mid; }
  int64_t a = 0, b = 0;
  window_bounds(runs, n_runs, contigs[lo].gbase, contigs[lo].len, W, step, i - win_base[lo], a, b);
  out[i] = make_int2((int)a, (int)b);
}

// Self-test of the cross-lane primitive the fill relies on: out[i] = value held by lane i-1.
__global__ void dpp_selftest_kernel(int* out) {
  int v = (int)threadIdx.x * 7 + 3;
  out[threadIdx.x] = shift_up_lane(v);
}

// ------------------------------------------------------------------------------------------------------------------
// mailbox (mailbox.hpp)
// ------------------------------------------------------------------------------------------------------------------
__global__ void mailbox_kernel(const uint32_t* src, int n, uint32_t* box, uint32_t seq) {
  CALITAS_TAIL_PRIO();
  for (int i = 0; i < n; i++) box[1 + i] = src[i];
  __threadfence_system();
  __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t mailbox_open(Mailbox& mb) {
  if (mb.host) return hipSuccess;
  void* h = nullptr;
  hipError_t e = hipHostMalloc(&h, (MAILBOX_WORDS + 1) * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return e;
  void* d = nullptr;
  e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) { (void)hipHostFree(h); return e; }
  mb.host = (volatile uint32_t*)h; mb.dev = (uint32_t*)d; mb.seq = 0;
  mb.host[0] = 0;
  return hipSuccess;
}

void mailbox_close(Mailbox& mb) {
  if (mb.host) (void)hipHostFree((void*)mb.host);
  mb.host = nullptr; mb.dev = nullptr;
}

hipError_t mailbox_post(Mailbox& mb, const uint32_t* src, int n, hipStream_t stream) {
  hipError_t e = mailbox_open(mb);
  if (e != hipSuccess) return e;
  if (n > MAILBOX_WORDS) return hipErrorInvalidValue;
  mb.seq++;
  hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(1), 0, stream, src, n, mb.dev, mb.seq);
  return hipGetLastError();
}

hipError_t mailbox_wait(Mailbox& mb, hipStream_t stream) {
  // Bounded: a kernel that never finishes would otherwise leave the caller (and every lane thread) spinning for good.  The limit is far
  // beyond any legitimate wait (the longest device stage of a PAM-less whole-genome pass is under a second).
  constexpr double kDeadlineSeconds = 120.0;
  Backoff wait;
  long long next_check_us = 200;               // now and then: is the stream still alive?
  while (mb.host[0] != mb.seq) {
    wait.pause();
    if (wait.spins >= 256 && wait.waited_us() >= next_check_us) {
      next_check_us = wait.waited_us() + 200;
      const hipError_t e = hipStreamQuery(stream);
      if (e != hipSuccess && e != hipErrorNotReady) return e;
      if (e == hipSuccess && mb.host[0] != mb.seq) {      // everything queued has run, yet nothing arrived
        if (mb.host[0] == mb.seq) break;
        return hipErrorUnknown;
      }
      if ((double)wait.waited_us() * 1e-6 > kDeadlineSeconds) return hipErrorLaunchTimeOut;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------

// start / stop (optional): events attached to the dispatch itself -- no marker packets before and after the kernel on the stream.
hipError_t launch_scan(const ScanArgs& a, int chunk, uint32_t n_tiles, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  if (n_tiles == 0) {
    if (start) { hipError_t e = hipEventRecord(start, stream); if (e != hipSuccess) return e; }
    return stop ? hipEventRecord(stop, stream) : hipSuccess;
  }
  dim3 grid(n_tiles), block(LANES_PER_TILE);
  constexpr unsigned pad = 0;
  switch (chunk) {
    case 64:  hipExtLaunchKernelGGL(scan_kernel<64>, grid, block, pad, stream, start, stop, 0, a); break;
    case 128: hipExtLaunchKernelGGL(scan_kernel<128>, grid, block, pad, stream, start, stop, 0, a); break;
    case 256: hipExtLaunchKernelGGL(scan_kernel<256>, grid, block, pad, stream, start, stop, 0, a); break;
    case 512: hipExtLaunchKernelGGL(scan_kernel<512>, grid, block, pad, stream, start, stop, 0, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void lane_setup_kernel(LaneSetupArgs a) {
  if (blockIdx.x == 0) {
    // The argument block is host memory: a load from it crosses the bus.  16 bytes per lane, all of them in flight at once -- byte by
    // byte the 700 bytes of a typical call took 15 us.
    const uint8_t* args = (const uint8_t*)__builtin_amdgcn_kernarg_segment_ptr();   // (a C-style cast: out of the constant address space)
    if (a.d_guides) {                                          // (null: the row stage's inputs only, the scan's went ahead)
      const uint32_t* g = reinterpret_cast<const uint32_t*>(args + offsetof(LaneSetupArgs, guide));
      uint32_t* dg = reinterpret_cast<uint32_t*>(a.d_guides);
      constexpr uint32_t n4 = sizeof(GuideDev) / 16, rest = (sizeof(GuideDev) % 16) / 4;
      if (threadIdx.x < n4) reinterpret_cast<uint4*>(dg)[threadIdx.x] = reinterpret_cast<const uint4*>(g)[threadIdx.x];
      else if (threadIdx.x < n4 + rest) dg[n4 * 4 + (threadIdx.x - n4)] = g[n4 * 4 + (threadIdx.x - n4)];
      if (threadIdx.x >= 64 && threadIdx.x < 72) a.d_counters[threadIdx.x - 64] = 0u;
    }
    if (a.d_row_counts && threadIdx.x >= 72 && threadIdx.x < 78) reinterpret_cast<uint32_t*>(a.d_row_counts)[threadIdx.x - 72] = 0u;
    if (a.d_blob && threadIdx.x >= 128) {
      const uint4* b = reinterpret_cast<const uint4*>(args + offsetof(LaneSetupArgs, blob));
      for (uint32_t i = threadIdx.x - 128; i < (a.blob_bytes + 15) / 16; i += 128) reinterpret_cast<uint4*>(a.d_blob)[i] = b[i];
    }
  }
  if (a.clear)
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.clear_bytes / 16; i += gridDim.x * 256) a.clear[i] = make_uint4(0u, 0u, 0u, 0u);
}

hipError_t launch_lane_setup(const LaneSetupArgs& a, hipStream_t stream) {
  static_assert(sizeof(GuideDev) % 4 == 0 && sizeof(GuideDev) / 16 + 4 <= 64 && sizeof(LaneSetupArgs) <= 4096 && offsetof(LaneSetupArgs, guide) % 16 == 0 &&
                offsetof(LaneSetupArgs, blob) % 16 == 0, "the lane's small inputs travel as kernel arguments, read in 16-byte pieces");
  if ((a.d_guides != nullptr) != (a.d_counters != nullptr) || a.blob_bytes > LANE_SETUP_BLOB || (a.clear_bytes & 15u)) return hipErrorInvalidValue;
  const unsigned grid = a.clear ? std::min<unsigned>(256u, std::max<unsigned>(1u, a.clear_bytes / (16u * 256u))) : 1u;
  hipLaunchKernelGGL(lane_setup_kernel, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_align(const AlignArgs& a, uint32_t n_blocks, hipStream_t stream) {
  // n_blocks counts 256-lane units (8 jobs)
  const dim3 grid(n_blocks * 4), block(64);                  // one wave per workgroup
  // scan records -> jobs: a lane per record, 256-lane workgroups striding over the records (their number is on the device)
  const uint32_t expand_blocks = std::max<uint32_t>(1u, std::min<uint32_t>(1024u, (a.rec_capacity + 255u) / 256u));
  hipLaunchKernelGGL(expand_kernel, dim3(expand_blocks), dim3(256), 0, stream, a);
  // three jobs per wave when no guide has more than 20 rows (max_guide_len 0: unknown)
  bool three = a.max_guide_len > 0 && a.max_guide_len <= 20;
  if (const char* env = TUNE_GET("CALITAS_ALIGN_LPJ")) three = three && std::atoi(env) == 21;   // (tests / measurements: 32 forces two jobs)
  bool pack = three && !a.sp.per_matrix && a.pack16 != 0;
  if (const char* env = TUNE_GET("CALITAS_ALIGN_PACK")) pack = pack && std::atoi(env) != 0;    // (tests / measurements: 0 = one job per lane group)
  if (pack) { hipLaunchKernelGGL(align_pk_kernel, grid, block, 0, stream, a); return hipGetLastError(); }
  if (a.sp.per_matrix) {
    if (three) hipLaunchKernelGGL((align_kernel<true, 21>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((align_kernel<true, 32>), grid, block, 0, stream, a);
  } else {
    if (three) hipLaunchKernelGGL((align_kernel<false, 21>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((align_kernel<false, 32>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

hipError_t launch_trace(const AlignArgs& a, uint32_t n_blocks, hipStream_t stream, hipEvent_t stop, Mailbox* post) {
  uint32_t* box = nullptr;
  uint32_t seq = 0;
  if (post) {
    hipError_t e = mailbox_open(*post);
    if (e != hipSuccess) return e;
    box = post->dev; seq = ++post->seq;
  }
  hipExtLaunchKernelGGL(trace_kernel, dim3(n_blocks), dim3(256), 0, stream, nullptr, stop, 0, a, box, seq);
  return hipGetLastError();
}

hipError_t launch_window_table(const Run* runs, int64_t n_runs, const ContigInfo* contigs, const uint64_t* win_base, int n_contigs,
                               uint64_t n_windows, int W, int step, int2* out, hipStream_t stream) {
  if (n_windows == 0) return hipSuccess;
  hipLaunchKernelGGL(window_table_kernel, dim3((unsigned)((n_windows + 255) / 256)), dim3(256), 0, stream, runs, n_runs, contigs,
                     win_base, n_contigs, W, step, out);
  return hipGetLastError();
}

hipError_t launch_dpp_selftest(int* out, hipStream_t stream) {
  hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, stream, out);
  return hipGetLastError();
}

}  // namespace calitas
