// scan_rows.hip -- the exact candidate filter of the SearchReference hot path, row-wise bit-parallel form (gfx950).
//
// What it decides (same rule as fgbio's Aligner.align(query, target, minScore) enumeration, called at
// SequentialGuideAligner.scala:261,278,295,299): for every reference position and both strands, whether the bottom row of the
// glocal DP reaches minGuideScore at that end column.  With the reference's linear gap costs that is "semi-global edit distance
// <= E" (SearchReference.scala:432-441), computed exactly with Myers' bit-vector recurrence.
//
// Orientation.  The first version of this kernel (kernels.hip, scan_kernel) kept one DP *column* (the L protospacer rows) in a
// 32-bit word and walked the text base by base: 13 VALU instructions per base and strand for L = 20 cells, 12 of the 32 bits idle,
// plus a table lookup per base.  Here the bit-vector runs along the *text*: a lane owns NW consecutive 32-base words of the
// reference as one long integer (plus NWARM warm-up words from its neighbour), and one step of the recurrence handles one
// protospacer ROW for all of those positions -- every bit of every instruction is a DP cell.  Per 32-base word and row: two
// v_bitop3 (x and xv straight from the text's two bit-planes and the row's base), one v_addc_co (the carry chain runs along the
// text), three more v_bitop3 and a v_or, two v_alignbit (the one-position shifts across word boundaries) and a v_and = 10
// instructions for 32 cells.  The Eq vector of a row ("is this position an A") is a function of the two bit-planes the packed
// reference keeps per 32 bases (low bits, high bits of the 2-bit codes) and never exists as a value: no table, no per-base index
// arithmetic, no precomputed planes.
// The reverse strand is the same recurrence on the bit-reversed, complemented words (right-to-left in the text).
//
// After the last row the lane holds the horizontal deltas of the bottom row; the bottom-row values are L + prefix sums of those
// deltas.  A byte-granular lower bound from masked popcounts (value at the byte's start minus the number of -1 steps in the byte)
// rejects 99.5 % of the words; the survivors go to a small LDS queue that the workgroup resolves position by position at the end
// of the tile (one lane per queued word), and only true candidate columns become ScanRecords -- the same records, bit for bit,
// as the column-wise kernel produced, so everything downstream is unchanged.
//
// Truncation at the left end of a lane's chain (the DP column left of the warm-up words is taken as 0..L) can only raise values,
// and a cell with true value <= E has its whole optimal path inside L + E - 1 <= 32 * NWARM columns, so every reported cell is exact.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <hip/hip_ext.h>

#include "common.hpp"
#include "kernels.hpp"

namespace calitas {

namespace {

constexpr int ROWS_STAGE = 192;   // ScanRecords staged in LDS per tile (flushed with one global atomic)
constexpr int ROWS_QCAP = 128;    // suspect words queued per wave

// GuideDev in the constant address space: loads through it are scalar (s_load) and may be hoisted
typedef const __attribute__((address_space(4))) GuideDev GuideConst;

struct SuspectWord {
  uint32_t p, m;      // +1 / -1 horizontal deltas of the bottom row over the word's 32 positions (chain order)
  int32_t s;          // bottom-row value before the word, biased by -(E+1): a negative running value = candidate column
  uint32_t id;        // bits 0-15 word index in the tile (text order), bit 16 direction, bits 17-23 guide
};

// where a tile's records go: an LDS stage flushed once per tile, the global list beyond that
struct RecordSink {
  ScanRecord* recs;
  uint32_t* rec_count;
  uint32_t rec_capacity;
  ScanRecord* s_recs;
  uint32_t* s_nrec;
};

__device__ __forceinline__ void stage_scan_record(const RecordSink k, uint32_t gword, uint32_t info) {
  ScanRecord r;
  r.gword = gword; r.info = info;
  const uint32_t slot = atomicAdd(k.s_nrec, 1u);        // LDS atomic
  if (slot < (uint32_t)ROWS_STAGE) { k.s_recs[slot] = r; return; }
  // stage full (dense tile): append directly, one global atomic for all the lanes of the wave that are here together -- a
  // returning atomic on one global word completes at ~90 per microsecond chip-wide, and a PAM-less search at max-guide-diffs 8
  // emits 1.8e8 records (one atomic each: 2 s of a pass)
  const unsigned long long here = __ballot(1);
  const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)here) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(k.rec_count, (uint32_t)__popcll(here));
  base = (uint32_t)__shfl((int)base, leader);
  const uint32_t g = base + (uint32_t)__popcll(here & ((1ull << lane) - 1ull));
  if (g < k.rec_capacity) k.recs[g] = r;
}

// Position-by-position walk over one suspect word: bit k of the result = the bottom-row value after position k is <= E.
// (everything by value: a reference to the kernel's argument block would force that block into scratch memory)
__device__ __noinline__ void resolve_suspect(const RecordSink sink, uint32_t w32_base, const SuspectWord q) {
  uint32_t hm = 0;
  int s = q.s;
#pragma unroll 8
  for (int k = 0; k < 32; k++) {
    s += (int)((q.p >> k) & 1u);
    s -= (int)((q.m >> k) & 1u);
    hm |= (uint32_t)(s < 0) << k;
  }
  if (hm == 0u) return;
  const uint32_t dir = (q.id >> 16) & 1u;
  if (dir) hm = __builtin_bitreverse32(hm);              // the reverse-strand chain runs right to left
  const uint32_t gword = (w32_base + (q.id & 0xFFFFu)) * 2u;   // ScanRecords address 16-base words
  const uint32_t tag = q.id & 0xFFFF0000u;               // direction and guide already sit where ScanRecord::info wants them
  if (hm & 0xFFFFu) stage_scan_record(sink, gword, (hm & 0xFFFFu) | tag);
  if (hm >> 16) stage_scan_record(sink, gword + 1u, (hm >> 16) | tag);
}

// One row of the recurrence over a chain of NC words (word 0 = lowest text position in chain order):
//   x  = eq & pv;   xv = eq | mv
//   t  = x + pv                         (the carry chain runs along the text: v_addc_co_u32)
//   mh = pv & ((t ^ pv) | xv);   ph = mv | (~pv & ~t & ~xv)
//   ph, mh move up one position (a 1 enters ph at the chain's low end: the DP column left of the chain is 0, 1, .., L)
//   pv = mh | ~(xv | ph);   mv = ph & xv
// (Where mv = 1 (so pv = 0) the textbook's xh = (t ^ pv) | eq is irrelevant to both of its uses -- ph = mv | ~(xh | pv) is 1 there,
// mh = pv & xh is 0 -- and where mv = 0, eq = xv; so Eq itself is not needed after the first line.)
//
// Eq comes straight from the two bit-planes of the text (lo, hi: bit j = low / high bit of base j): eq = [~]hi & [~]lo folds into the
// two instructions that use it -- x and xv are three-input functions of (lo, hi, pv) and (lo, hi, mv), one v_bitop3_b32 each with
// the row's base in the truth table.  (Round 2's first version kept four precomputed planes per word and the x / xv arrays of a row
// in registers: 166 VGPRs, three waves per SIMD.)  The truth table's bit (s0 << 2 | s1 << 1 | s2) is the result; s0 = lo, s1 = hi.
// The row is in two parts.  Part 1 -- x and xv of every word -- exists once per base (its truth tables are immediates) and is inline
// assembly: as C++ the compiler hoists eq out of the row loop (four planes per word in registers again).  Part 2 is one copy of code
// for all rows: with four copies of the *whole* row behind the switch the compiler keeps the state in two register sets and moves all
// 2 NC words at the loop's back edge (34 v_mov per row and 34 more registers at NC = 17).
template <int BASE> constexpr int eq_table() { return BASE == 0 ? 0x03 : BASE == 1 ? 0x30 : BASE == 2 ? 0x0C : 0xC0; }   // A C G T

// A row whose protospacer base is one of A C G T.  MASKED: the tile has exception bases (exc: N, padding, IUPAC codes in the text):
// code 0 never matches, code 1 ("wild": ~hi & lo as stored, hi & ~lo after the reverse strand's complement) matches every row and
// the aligner decides exactly -- still one three-input function of (lo, hi, exc).
template <int NC, int BASE, bool MASKED, int DIR>
__device__ __forceinline__ void myers_row_part1(const uint32_t (&lo)[NC], const uint32_t (&hi)[NC], const uint32_t (&exc)[MASKED ? NC : 1],
                                                const uint32_t (&pv)[NC], const uint32_t (&mv)[NC], uint32_t (&x)[NC], uint32_t (&xv)[NC]) {
  constexpr int EQ = eq_table<BASE>();
  constexpr int TT_X = EQ & 0xAA, TT_XV = EQ | 0xAA;                       // eq & pv, eq | mv            (s2 = pv / mv)
  constexpr int TT_EM = (EQ & 0x55) | (DIR ? 0x08 : 0x20);                 // (eq & ~exc) | (exc & wild)  (s2 = exc)
#pragma unroll
  for (int w = 0; w < NC; w++) {
    if (MASKED) {
      uint32_t em;
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(em) : "v"(lo[w]), "v"(hi[w]), "v"(exc[w]), "n"(TT_EM));
      asm("v_and_b32 %0, %1, %2 ; base %3" : "=v"(x[w]) : "v"(em), "v"(pv[w]), "n"(BASE));
      asm("v_or_b32 %0, %1, %2 ; base %3" : "=v"(xv[w]) : "v"(em), "v"(mv[w]), "n"(BASE));
    } else {
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(x[w]) : "v"(lo[w]), "v"(hi[w]), "v"(pv[w]), "n"(TT_X));
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(xv[w]) : "v"(lo[w]), "v"(hi[w]), "v"(mv[w]), "n"(TT_XV));
    }
  }
}

// Part 1 of a row whose protospacer base is an IUPAC letter: union of the bases in its set (A=1 C=2 G=4 T=8).
template <int NC, bool MASKED, int DIR>
__device__ __forceinline__ void myers_row_part1_set(const uint32_t (&lo)[NC], const uint32_t (&hi)[NC], const uint32_t (&exc)[MASKED ? NC : 1],
                                                    const uint32_t set, const uint32_t (&pv)[NC], const uint32_t (&mv)[NC], uint32_t (&x)[NC],
                                                    uint32_t (&xv)[NC]) {
  // all-ones / all-zeros per base of the set, in vector registers (an instruction takes one scalar operand at most)
  uint32_t ka = 0u - (set & 1u), kc = 0u - ((set >> 1) & 1u), kg = 0u - ((set >> 2) & 1u), kt = 0u - ((set >> 3) & 1u);
  asm("" : "+v"(ka), "+v"(kc), "+v"(kg), "+v"(kt));
#pragma unroll
  for (int w = 0; w < NC; w++) {
    // eq = hi ? (lo ? kt : kg) : (lo ? kc : ka): three selections (truth table 0xCA = s0 ? s1 : s2); as (~hi & ~lo & ka) | ... in
    // C++ the compiler keeps ~lo, ~hi and the exception term of every word in registers across the rows
    uint32_t hi1, hi0, eq;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(hi1) : "v"(lo[w]), "v"(kt), "v"(kg));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(hi0) : "v"(lo[w]), "v"(kc), "v"(ka));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(eq) : "v"(hi[w]), "v"(hi1), "v"(hi0));
    if (MASKED) {
      uint32_t wt;                                        // exc & wild; kt rides along so that the term is not loop-invariant
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4 ; %5" : "=v"(wt) : "v"(lo[w]), "v"(hi[w]), "v"(exc[w]), "n"(DIR ? 0x08 : 0x20), "v"(kt));
      asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xba" : "=v"(eq) : "v"(eq), "v"(exc[w]), "v"(wt));   // (eq & ~exc) | wt
    }
    x[w] = eq & pv[w]; xv[w] = eq | mv[w];
  }
}

template <int NC>
__device__ __forceinline__ void myers_row_part2(const uint32_t (&x)[NC], const uint32_t (&xv)[NC], uint32_t (&pv)[NC], uint32_t (&mv)[NC]) {
  uint32_t carry = 0u, php = 0x80000000u, mhp = 0u;
#pragma unroll
  for (int w = 0; w < NC; w++) {
    uint32_t cout;
    const uint32_t t = __builtin_addc(x[w], pv[w], carry, &cout);
    carry = cout;
    const uint32_t mh = pv[w] & ((t ^ pv[w]) | xv[w]);
    const uint32_t ph = mv[w] | (~pv[w] & ~t & ~xv[w]);
    const uint32_t phs = __builtin_amdgcn_alignbit(ph, php, 31);
    const uint32_t mhs = __builtin_amdgcn_alignbit(mh, mhp, 31);
    php = ph; mhp = mh;
    pv[w] = mhs | ~(xv[w] | phs);
    mv[w] = phs & xv[w];
  }
}

// One strand of one wave's share of a tile (DIR 0: the text as it is, left to right; DIR 1: its reverse complement).
template <int NW, int NWARM, bool MASKED, int DIR>
__device__ __forceinline__ void scan_wave_strand(const ScanArgs& a, int wave, int wl, uint32_t w32_base, const uint2* s_pl, SuspectWord* s_q,
                                                 uint32_t* s_qn, const RecordSink sink) {
  constexpr int NC = NW + NWARM;            // words of a lane's chain
  constexpr int CSTR = NW + 1;              // uint2 per staged chunk (padded: lane l reads 8-byte word l * CSTR + k, conflict-free)
  const int tid = wave * 64 + wl;           // lane of the tile
  const GuideConst* guides = (const GuideConst*)(a.guides);   // constant address space: scalar loads
  // ---- the two bit-planes of this lane's chain (and its exception bases) ----
  uint32_t lo[NC], hi[NC], exc[MASKED ? NC : 1];
#pragma unroll
  for (int v = 0; v < NC; v++) {
    // text word behind chain word v: DIR 0 reads left to right (warm-up = tail of the left neighbour's chunk),
    // DIR 1 right to left (warm-up = head of the right neighbour's chunk)
    int chunk, k;
    if (DIR == 0) { chunk = (v < NWARM) ? wl : wl + 1; k = (v < NWARM) ? NW - NWARM + v : v - NWARM; }
    else          { chunk = (v < NWARM) ? wl + 2 : wl + 1; k = (v < NWARM) ? NWARM - 1 - v : NW - 1 - (v - NWARM); }
    const uint2 x = s_pl[chunk * CSTR + k];
    lo[v] = x.x; hi[v] = x.y;
    if (MASKED) exc[v] = a.mask[(int64_t)w32_base + (int64_t)((wave * 64 + chunk - 1) * NW + k)];
    if (DIR) {                                // reverse strand: complemented text, read right to left
      lo[v] = ~__builtin_bitreverse32(lo[v]); hi[v] = ~__builtin_bitreverse32(hi[v]);
      if (MASKED) exc[v] = __builtin_bitreverse32(exc[v]);
    }
  }

  for (int gi = 0; gi < a.n_guides; gi++) {
    const int L = guides[gi].L, E = guides[gi].scan_max_edits;
    const uint64_t rows_lo = guides[gi].row_sets[0], rows_hi = guides[gi].row_sets[1];   // 4-bit base set per protospacer row
    uint32_t pv[NC], mv[NC];
#pragma unroll
    for (int v = 0; v < NC; v++) { pv[v] = 0u; mv[v] = 0u; }          // row 0 of the DP is all zeros (free start in the text)
    for (int i = 0; i < L; i++) {
      const uint32_t set = (uint32_t)(((i < 16 ? rows_lo : rows_hi) >> ((i & 15) * 4)) & 15u);
      uint32_t x[NC], xv[NC];
      switch (set) {
        case 1: myers_row_part1<NC, 0, MASKED, DIR>(lo, hi, exc, pv, mv, x, xv); break;
        case 2: myers_row_part1<NC, 1, MASKED, DIR>(lo, hi, exc, pv, mv, x, xv); break;
        case 4: myers_row_part1<NC, 2, MASKED, DIR>(lo, hi, exc, pv, mv, x, xv); break;
        case 8: myers_row_part1<NC, 3, MASKED, DIR>(lo, hi, exc, pv, mv, x, xv); break;
        default: myers_row_part1_set<NC, MASKED, DIR>(lo, hi, exc, set, pv, mv, x, xv);
      }
      myers_row_part2<NC>(x, xv, pv, mv);
    }
    // ---- bottom row: value before chain word 0 is L; hunt for values <= E ----
    int s = L - (E + 1);
#pragma unroll
    for (int v = 0; v < NWARM; v++) s += __builtin_popcount(pv[v]) - __builtin_popcount(mv[v]);
#pragma unroll
    for (int v = NWARM; v < NC; v++) {
      const uint32_t P = pv[v], M = mv[v];
      const int u1 = __builtin_popcount(P & 0xFFu) + s, u2 = __builtin_popcount(P & 0xFFFFu) + s, u3 = __builtin_popcount(P & 0xFFFFFFu) + s;
      const int d1 = __builtin_popcount(M & 0xFFu), d2 = __builtin_popcount(M & 0xFFFFu), d3 = __builtin_popcount(M & 0xFFFFFFu);
      const int d4 = __builtin_popcount(M);
      // lower bound of the running value inside each byte: its value at the byte's start minus the -1 steps in the byte
      const int lb = min(min(s - d1, u1 - d2), min(u2 - d3, u3 - d4));
      if (lb < 0) {
        SuspectWord q;
        q.p = P; q.m = M; q.s = s;
        const int k = v - NWARM;
        int lane_of_tile = tid;
        asm volatile("" : "+v"(lane_of_tile));          // computed here, not once per word ahead of the loops (16 registers)
        q.id = (uint32_t)(lane_of_tile * NW + (DIR ? NW - 1 - k : k)) | ((uint32_t)DIR << 16) | ((uint32_t)gi << 17);
        const uint32_t slot = atomicAdd(s_qn, 1u);     // LDS atomic
        if (slot < (uint32_t)ROWS_QCAP) s_q[slot] = q;
        else resolve_suspect(sink, w32_base, q);        // queue full (dense repeats): resolve in place
      }
      s += __builtin_popcount(P) - d4;
    }
  }
}

// One wave's share of a tile: 64 lanes x NW words, staged (with one halo chunk on each side) in the wave's own LDS region, so the
// four waves of a workgroup never wait for each other.
template <int NW, int NWARM, bool MASKED>
__device__ __forceinline__ void scan_wave_rows(const ScanArgs& a, uint32_t tile, int wave, int wl, const uint2* s_pl, SuspectWord* s_q,
                                               uint32_t* s_qn, const RecordSink sink) {
  const uint32_t w32_base = tile * (uint32_t)(LANES_PER_TILE * NW);   // first 32-base word of the tile
  scan_wave_strand<NW, NWARM, MASKED, 0>(a, wave, wl, w32_base, s_pl, s_q, s_qn, sink);
  scan_wave_strand<NW, NWARM, MASKED, 1>(a, wave, wl, w32_base, s_pl, s_q, s_qn, sink);
}

__device__ __forceinline__ void wave_sync_lds() {   // orders this wave's LDS writes before its later LDS reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup per tile of the packed space, one wave per quarter of it.  Dead tiles (nothing but upper-case N / padding, which
// every window trims away) exit at once; tiles with exception bases take the MASKED instantiation (block-uniform branch).
// After the one barrier at the start (the shared counters) the waves run on their own: private staging region, private suspect
// queue; the wave that finishes last flushes the tile's records.
template <int NW, int NWARM>
__global__ __launch_bounds__(LANES_PER_TILE) void scan_rows_kernel(ScanArgs a) {
  constexpr int CSTR = NW + 1;
  constexpr int WAVE_PL = (64 + 2) * CSTR;            // staged 8-byte words per wave: its 64 chunks and one halo chunk each side
  __shared__ uint2 s_pl[4 * WAVE_PL];
  __shared__ SuspectWord s_q[4][ROWS_QCAP];
  __shared__ ScanRecord s_recs[ROWS_STAGE];
  __shared__ uint32_t s_nrec, s_done, s_qn[4];
  const uint32_t tile = blockIdx.x * a.tile_stride + a.tile_offset;
  const TileInfo ti = a.tiles[tile];
  if (ti.flag == 2u || ti.contig == 0xFFFFFFFFu) return;
  if (a.chrom_index >= 0 && ti.contig != (uint32_t)a.chrom_index) return;
  const int tid = threadIdx.x, wave = tid >> 6, wl = tid & 63;
  if (tid == 0) { s_nrec = 0; s_done = 0; }
  if (wl == 0) s_qn[wave] = 0;
  __syncthreads();
  // ---- stream this wave's quarter of the tile (+ one halo chunk each side) into LDS: 16-byte coalesced loads, padded scatter ----
  uint2* my_pl = &s_pl[wave * WAVE_PL];
  {
    const uint64_t w0 = (uint64_t)tile * (LANES_PER_TILE * NW) + (uint64_t)(wave * 64 * NW);
    const uint4* src = reinterpret_cast<const uint4*>(a.planes + (w0 - NW));
    constexpr int NQ = (64 + 2) * NW / 2;             // 16-byte pieces (two 32-base words each)
    for (int q = wl; q < NQ; q += 64) {
      const uint4 v = src[q];
      const int i = q * 2;
      const int vc = i / NW, k = i % NW;              // NW is even: the pair stays inside one chunk
      uint2* d = &my_pl[vc * CSTR + k];
      d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w);
    }
  }
  wave_sync_lds();
  const uint32_t w32_base = tile * (uint32_t)(LANES_PER_TILE * NW);
  const RecordSink sink{a.recs, a.rec_count, a.rec_capacity, s_recs, &s_nrec};
  if (ti.flag != 0u) scan_wave_rows<NW, NWARM, true>(a, tile, wave, wl, my_pl, s_q[wave], &s_qn[wave], sink);
  else scan_wave_rows<NW, NWARM, false>(a, tile, wave, wl, my_pl, s_q[wave], &s_qn[wave], sink);
  // ---- resolve this wave's queued suspect words, one lane each ----
  wave_sync_lds();
  const uint32_t nq = min(s_qn[wave], (uint32_t)ROWS_QCAP);
  for (uint32_t i = wl; i < nq; i += 64) resolve_suspect(sink, w32_base, s_q[wave][i]);
  // ---- the last wave to get here flushes the tile's records with one global atomic ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  uint32_t arrived = 0;
  if (wl == 0) arrived = atomicAdd(&s_done, 1u);
  arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
  if (arrived != 3u) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const uint32_t n = min(s_nrec, (uint32_t)ROWS_STAGE);
  if (n == 0) return;
  uint32_t base = 0;
  if (wl == 0) base = atomicAdd(a.rec_count, n);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  for (uint32_t i = wl; i < n; i += 64) {
    const uint32_t g = base + i;
    if (g < a.rec_capacity) a.recs[g] = s_recs[i];
  }
}

// codes[] (2 bits per base, interleaved) -> planes[] (per 32 bases: low bits of the codes, high bits of the codes).
__device__ __forceinline__ uint32_t even_bits(uint32_t x) {   // bits 0, 2, 4, .. of x gathered into the low 16 bits
  x &= 0x55555555u;
  x = (x | (x >> 1)) & 0x33333333u;
  x = (x | (x >> 2)) & 0x0F0F0F0Fu;
  x = (x | (x >> 4)) & 0x00FF00FFu;
  x = (x | (x >> 8)) & 0x0000FFFFu;
  return x;
}

__global__ void planes_kernel(const uint32_t* codes, uint2* planes, uint64_t n32) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n32) return;
  const uint32_t c0 = codes[2 * i], c1 = codes[2 * i + 1];
  planes[i] = make_uint2(even_bits(c0) | (even_bits(c1) << 16), even_bits(c0 >> 1) | (even_bits(c1 >> 1) << 16));
}

}  // namespace

hipError_t launch_planes(const uint32_t* codes, uint2* planes, uint64_t n32, hipStream_t stream) {
  if (n32 == 0) return hipSuccess;
  hipLaunchKernelGGL(planes_kernel, dim3((unsigned)((n32 + 255) / 256)), dim3(256), 0, stream, codes, planes, n32);
  return hipGetLastError();
}

// start / stop (optional): events attached to the dispatch itself -- no marker packets before and after the kernel on the stream.
hipError_t launch_scan_rows(const ScanArgs& a, int chunk, int warm_words, uint32_t n_tiles, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
  if (n_tiles == 0) {
    if (start) { hipError_t e = hipEventRecord(start, stream); if (e != hipSuccess) return e; }
    return stop ? hipEventRecord(stop, stream) : hipSuccess;
  }
  const dim3 grid(n_tiles), block(LANES_PER_TILE);
#define CALITAS_LAUNCH_ROWS(NW, NWARM) hipExtLaunchKernelGGL((scan_rows_kernel<NW, NWARM>), grid, block, 0, stream, start, stop, 0, a)
  if (warm_words == 1) {
    switch (chunk) {
      case 64:  CALITAS_LAUNCH_ROWS(2, 1); break;
      case 128: CALITAS_LAUNCH_ROWS(4, 1); break;
      case 256: CALITAS_LAUNCH_ROWS(8, 1); break;
      case 512: CALITAS_LAUNCH_ROWS(16, 1); break;
      default: return hipErrorInvalidValue;
    }
  } else if (warm_words == 2) {
    switch (chunk) {
      case 64:  CALITAS_LAUNCH_ROWS(2, 2); break;
      case 128: CALITAS_LAUNCH_ROWS(4, 2); break;
      case 256: CALITAS_LAUNCH_ROWS(8, 2); break;
      case 512: CALITAS_LAUNCH_ROWS(16, 2); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    return hipErrorInvalidValue;
  }
#undef CALITAS_LAUNCH_ROWS
  return hipGetLastError();
}

}  // namespace calitas
