// hits_dev.hpp -- device code and scratch shared by hits.hip (the general removeOverlaps / sort / rows kernels) and binned.hip (the
// same stages fused per reference bin): the hit record, its coordinates, and the wave-per-row builder of a hits.txt row's middle part.
// Private to those two translation units (everything sits in an anonymous namespace there).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

#include "../../include/calitas_hip.h"
#include "common.hpp"
#include "hits.hpp"
#include "mailbox.hpp"
#include "refpack.hpp"

namespace calitas {

namespace {

constexpr int SCORE_BITS = 14;
constexpr uint32_t CLUSTER_MAX = 1u << 20;

struct HitRec {
  int32_t contig, start, end, gstart, gend, score, rh_end;
  uint32_t minus;
};

struct RowConstDev {
  uint32_t head_off, head_len, tail_off, tail_len, plen_off, plen_len;
  uint32_t q_off[MAX_PAMS + 1], q_len[MAX_PAMS + 1], pu_off[MAX_PAMS + 1], pu_len[MAX_PAMS + 1];
};

// GuideAlignment coordinates (GA:21-31 with the '+' rule in aligner space, mapped per SGA:260-313) and ReferenceHit.end (RH:135-138)
// of one accepted alignment.
__device__ __forceinline__ HitRec hit_record(const RawAln* rp, const GuideDev* guides, const uint64_t* win_base, const int2* win) {
  struct { uint32_t contig, window_k; int32_t score; int t_start, t_end_guide, dir, guide, pam, offset, n_ops; } r;
  r.contig = rp->contig; r.window_k = rp->window_k; r.score = rp->score; r.t_start = rp->t_start; r.t_end_guide = rp->t_end_guide;
  r.dir = rp->dir; r.guide = rp->guide; r.pam = rp->pam; r.offset = rp->offset; r.n_ops = rp->n_ops;
  const GuideDev& g = guides[r.guide];
  const int ng = r.n_ops;
  int pam_len = 0, gap = 0;
  if (r.pam >= 0) { pam_len = g.pam_len[r.pam]; gap = r.offset; }
  // aligner-order op k is traceback op ng - 1 - k.  Everything left of the first / right of the last protospacer column is 'D'
  // (GA:21-31 with the '+' rule in aligner space, SGA:264,281,297,302).
  const OpCounts oc = count_ops(load_ops_words(rp->ops), ng);
  int lead = oc.lead_d, trail = oc.trail_d;
  const int t_guide = oc.not_ins;
  if (lead == ng) { lead = 0; trail = ng; }            // no protospacer column at all: cannot happen, kept total
  const int left_delta = lead, right_delta = trail + gap + pam_len;
  const int tlen = t_guide + gap + pam_len;
  const int start_s = (int)r.t_start - 1, end_s = (int)r.t_end_guide + r.offset + pam_len;   // SGA:515-516
  const int gstart_s = start_s + left_delta, gend_s = end_s - right_delta;
  const int2 w = win[win_base[r.contig] + r.window_k];
  HitRec h;
  h.contig = (int32_t)r.contig; h.score = r.score;
  if (r.dir == 0) { h.start = w.x + start_s; h.end = w.x + end_s; h.gstart = w.x + gstart_s; h.gend = w.x + gend_s; }   // SGA:297, 281
  else            { h.start = w.y - end_s; h.end = w.y - start_s; h.gstart = w.y - gend_s; h.gend = w.y - gstart_s; }   // SGA:303-309, 271-274
  const bool plus = g.pam5 ? (r.dir == 1) : (r.dir == 0);
  h.minus = plus ? 0u : 1u;
  h.rh_end = h.gstart + tlen - 1;                      // RH:135-138
  return h;
}

// (The four stages below exist as functions of one index: each has a kernel of its own, and hits_small_kernel runs them one after
// the other in a single workgroup when the call has at most HITS_SMALL alignments -- four launches less on the path of a small call.)
__device__ __forceinline__ void hit_body(const uint32_t i, const RawAln* fin, const GuideDev* guides, const uint64_t* win_base, const int2* win,
                                         int score_hi, HitRec* hits, uint64_t* keys, uint32_t* vals, uint32_t* wks, uint32_t* flags) {
  const HitRec h = hit_record(fin + i, guides, win_base, win);
  hits[i] = h;
  int sb = score_hi - h.score;
  if (sb < 0 || sb >= (1 << SCORE_BITS) || h.gstart < 0) { atomicOr(flags, HITS_FLAG_SCORE_RANGE); sb = 0; }
  keys[i] = ((uint64_t)(uint32_t)h.contig << 46) | ((uint64_t)(uint32_t)h.gstart << 15) | ((uint64_t)h.minus << 14) | (uint64_t)sb;
  vals[i] = i;
  wks[i] = fin[i].window_k;
}

constexpr int HIT_MAX_LEN = CALITAS_MAX_OPS;   // a hit covers at most this many reference bases (ReferenceHit.end - start + 1)

// ---- rows ------------------------------------------------------------------------------------------------------------
// A row is  head | chromosome \t | middle | tail  where head and tail are the same for every row of the call.
//   mid_kernel: one *wave* per row builds the middle part: lane i owns padded column i of the alignment (op, query, target and
//               alignment characters; the counts of GuideAlignment are popcounts of wave ballots), lane f owns field f's length and
//               -- for the numeric fields -- its digits; a prefix sum over the 25 field lengths places every field in the wave's
//               line buffer, which is copied to a fixed-stride staging buffer with coalesced dword stores.
//   out_kernel: after the exclusive scan of the lengths, a wave assembles row after row at its final offset with
//               coalesced byte stores (head / tail come from LDS).
// (The first version ran one lane per row with its working arrays in a 560-byte LDS slot: 36 KB per single-wave workgroup, each of
// which kept a four-wave workgroup of the next range's scan off its CU, and a 100 us chain of dependent LDS round trips -- DESIGN.md 4.4.)

__device__ __forceinline__ char comp_base(char c) {   // fgbio Sequences.complement on an upper-case base
  switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'U': return 'A';
    case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
    case 'V': return 'B'; case 'B': return 'V'; case 'H': return 'D'; case 'D': return 'H';
    default: return c;
  }
}

// "ACGT"[code] and "=XID"[op] from a constant in a register (indexing the string literals is a load from constant memory per character)
__device__ __forceinline__ char base_letter(uint32_t code) { return (char)((0x54474341u >> (8u * code)) & 0xFFu); }   // A C G T
__device__ __forceinline__ char op_letter(int op) { return (char)((0x4449583Du >> (8 * op)) & 0xFFu); }               // = X I D

__device__ char base_upper_dev(const HitsRef& ref, uint64_t gpos) {
  if ((ref.mask[gpos >> 5] >> (gpos & 31)) & 1u) {
    const int64_t r = run_floor(ref.runs, ref.n_runs, gpos);
    uint8_t ch = 0;
    if (r >= 0 && gpos < ref.runs[r].start + ref.runs[r].len) ch = ref.runs[r].ch;
    if (ch == 0) return 'N';
    return (char)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);
  }
  return base_letter((ref.codes[gpos >> 4] >> ((gpos & 15) * 2)) & 3u);
}

constexpr int HITS_BOX_LATE = 8;      // word of the work's mailbox for flags raised while rows are written (the post uses words 0..6)
constexpr int MID_COLS = 64;          // padded columns a row may have on this path: one per lane
constexpr int MID_LINE = 6 * MID_COLS + 128;   // bytes of a wave's line buffer = the largest mid_bound
constexpr int MID_FWD = 128;          // reference bases staged per row: the alignment and its flanks
constexpr int MID_FIELDS = 25;

struct MidArgs {
  HitsRef ref;
  RowConstDev rc;
  const char* blob;
  const uint32_t* name_off;
  const RawAln* fin;
  const HitRec* hits;
  const GuideDev* guides;
  const uint8_t* keep;       // per sorted position: survives removeOverlaps
  const uint32_t* order;     // sorted values: index into fin / hits
  uint32_t n;
  uint32_t mid_bound;        // bytes reserved for the middle part = staging stride (<= MID_LINE)
  uint32_t n_max;            // most padded columns a row of this search can have (<= MID_COLS)
  uint32_t blob_bytes;       // constant strings, copied to LDS by each block
  uint32_t* n_rows;          // out: number of live rows
  uint32_t n_dev;            // order[k] >= n_dev: a hit the caller built (HitsExt) -- no row to build, its length comes from ext_off
  const uint64_t* ext_off;
  uint32_t* ext_kept;        // out: how many of those were kept
  unsigned long long own_lo, own_hi;   // HitsOwn: rows only for hits with own_lo <= (contig << 32 | coordinate_start) < own_hi (0 / ~0: all)
};

static_assert(offsetof(RawAln, ops) % 4 == 0 && sizeof(RawAln) % 4 == 0, "RawAln::ops must be word aligned");

struct RowIn {             // the fields of one RawAln a row needs, ops as five words (2 bits per op, traceback order)
  uint32_t w0, w1, w2, w3, w4;
  int n_ops, pam, offset;
  uint32_t pam_x;
};
static_assert(RAW_MAX_OPS / 16 == 5, "RowIn holds five ops words");
struct RowGuide { int L, pam5, pam_len; };   // what a row needs of its GuideDev (pam_len: of the row's PAM, 0 without one)

// op i of the row: the word is chosen by comparison (the five words are wave-uniform and live in scalar registers; indexing them
// as an array made the compiler spill them to scratch and load per lane)
__device__ __forceinline__ int row_op(const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t w3, const uint32_t w4, int i) {
  const int k = i >> 4;
  uint32_t w = w0;
  w = (k == 1) ? w1 : w; w = (k == 2) ? w2 : w; w = (k == 3) ? w3 : w; w = (k == 4) ? w4 : w;
  return (int)((w >> ((i & 15) * 2)) & 3u);
}

// Read-only inputs of a row are the same for all lanes of its wave: through the constant address space they are scalar loads into
// scalar registers (everything they point to was written by earlier kernels).
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* uniform_ptr(const T* p) {
  return (const __attribute__((address_space(4))) T*)p;
}

__device__ __forceinline__ void wave_lds_sync() {       // LDS writes of this wave visible to all its lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long bits_below(int i) { return i >= 64 ? ~0ull : (1ull << i) - 1ull; }
// number of set bits of a wave-uniform mask below this lane: two instructions (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int bits_before_lane(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// `v` (wave-uniform) in lane L, `old` elsewhere: one v_writelane_b32
template <int L>
__device__ __forceinline__ int set_lane(int v, int old) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(L));
  return old;
}

// The middle part of one hits.txt row (RH:210-254, columns coordinate_start .. unpadded_target_sequence_length) in `line`, built by
// the 64 lanes of a wave; every argument but `lane` is the same in all of them.  `fwd` holds MID_FWD bytes, `blob` is the LDS copy
// of the constant strings.  Returns the length, or -1 when the row has more columns (or a longer span) than this path lays out.
// EMIT = false: only the length (the same arithmetic, nothing fetched or written; line / fwd may be null) -- what the binned path needs
// to place a row before it is built.
template <bool EMIT = true>
__device__ __forceinline__ int build_middle(uint8_t* line, uint8_t* fwd, const MidArgs& a, const uint8_t* blob, const RowIn r, const HitRec h,
                                            const RowGuide g, const int lane) {
  const int pam_len = g.pam_len, gap = r.pam >= 0 ? r.offset : 0;
  const int ng = r.n_ops, n = ng + gap + pam_len;
  const bool minus = h.minus != 0;
  // one fetch covers the alignment and all four flanks (RH:213-216); a minus-strand hit keeps the complemented bases: every
  // reader below wants them in guide orientation
  const int lo = min(h.start - 8, h.gstart - 10), hi = max(h.end + 8, h.gend + 10);
  if (n > (int)a.n_max || n > MID_COLS || hi - lo > MID_FWD) return -1;
  if (EMIT) {
    const uint64_t c_gbase = uniform_ptr(a.ref.contigs)[h.contig].gbase, c_len = uniform_ptr(a.ref.contigs)[h.contig].len;
    for (int x = lane; x < hi - lo; x += 64) {
      const int64_t p = (int64_t)lo + x;
      char b = 'N';                                                                          // RH:262-264
      if (p >= 0 && p < (int64_t)c_len) b = base_upper_dev(a.ref, c_gbase + (uint64_t)p);
      if (minus) b = comp_base(b);
      fwd[x] = (uint8_t)b;
    }
  }
  // ---- column `lane` of the alignment in guide orientation: guide part (stored in traceback order), gap to the PAM, PAM
  //      (SGA:472-476); reversed for a 5' PAM (SGA:267-269)
  const bool valid = lane < n;
  int op = 0;                                             // 0 '=', 1 'X', 2 'I', 3 'D'
  {
    const int k = g.pam5 ? n - 1 - lane : lane;
    const int guide_op = row_op(r.w0, r.w1, r.w2, r.w3, r.w4, ng - 1 - k), pam_op = (int)((r.pam_x >> ((k - ng - gap) & 15)) & 1u);
    op = k < ng ? guide_op : k < ng + gap ? 3 : pam_op;
    if (!valid) op = 0;
  }
  const unsigned long long MX = __ballot(valid && op == 1), MI = __ballot(valid && op == 2), MD = __ballot(valid && op == 3);
  const unsigned long long V = bits_below(n), nonD = V & ~MD, nonI = V & ~MI;
  const int qi = bits_before_lane(nonD), ti = bits_before_lane(nonI);
  // Alignment.paddedString (SGA:511): query, alignment and target character of this column
  const uint8_t* q = blob + a.rc.q_off[r.pam + 1];
  char qc = '-';
  if (valid && op != 3) qc = (char)q[qi];
  const bool q_low = valid && qc >= 'a', q_up = valid && qc >= 'A' && qc <= 'Z';   // (the query holds letters only: lower = the PAM)
  const unsigned long long ML = __ballot(q_low), MU = __ballot(q_up);
  const char ac = op == 0 ? '|' : op == 1 ? '.' : '~';
  wave_lds_sync();                                        // fwd[] is complete
  // j-th target base of the alignment in guide orientation
  const int t_first = minus ? h.end - 1 - lo : h.start - lo, t_step = minus ? -1 : 1;
  char tc = '-';
  if (EMIT && valid && op != 2) tc = (char)fwd[t_first + t_step * ti];
  // unpaddedTargetWithoutPam (GA:111-115): the target bases under the first .. last upper-case query column
  const int ps = MU ? __ffsll((long long)MU) - 1 : 0, pe = MU ? 63 - __clzll((long long)MU) : -1;
  const unsigned long long span_cols = bits_below(pe + 1) & ~bits_below(ps);
  const int utn = __popcll(nonI & span_cols), ut0 = __popcll(nonI & bits_below(ps));
  // GuideAlignment.count (GA:139-163) as ballots.  Mismatches: '.' columns by the case of the query base.  Gaps ('~' columns): an
  // inserted query base counts by its own case; a deleted one ('-' in the padded guide) by its nearest non-dash neighbours
  // (previousNonDash / nextNonDash, GA:168-182: the scan stops at the first / last column, which is then a dash itself).
  const int gmm = __popcll(MX & ~ML), pam_mm = __popcll(MX & ML), edits = __popcll(MX | MI | MD);
  bool guide_gap = valid && op == 2 && !q_low;            // is_lower(pg[i]) == false
  if (valid && op == 3) {                                 // (no lane gets here in a row without deletions)
    const unsigned long long below = bits_below(lane), left = nonD & below, right = nonD & ~below;   // (this column is not in nonD)
    const bool prev_up = left != 0 && ((MU >> (63 - __clzll((long long)left))) & 1ull);
    const bool next_up = right != 0 && ((MU >> (__ffsll((long long)right) - 1)) & 1ull);
    guide_gap = prev_up || next_up;                       // both_sides = false, lower = false: an upper-case letter on either side
  }
  const int ggp = __popcll(__ballot(guide_gap));
  // Cigar.coalesce + toString: a run starts where the op changes; its text is the length and the op letter
  const int op_prev = __builtin_amdgcn_update_dpp(op, op, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const bool run_start = valid && (lane == 0 || op != op_prev);
  const unsigned long long RS = __ballot(run_start);
  int run_len = 0;
  {
    // next run start above this lane: clear bits 0..lane of RS (lane-dependent shift of a uniform mask)
    const unsigned long long nx = lane >= 63 ? 0ull : (RS >> (lane + 1));
    run_len = nx ? __ffsll((long long)nx) : n - lane;
  }
  const unsigned long long RL = __ballot(run_start && run_len >= 10);       // (a run has at most 64 columns: one or two digits)
  const int cigar_len = 2 * __popcll(RS) + __popcll(RL);
  // ---- the 25 fields: lane f holds the length of field f (every field is followed by a tab) and, for a number, its value
  const int pu_len = (int)a.rc.pu_len[r.pam + 1];
  int flen = 0, val = 0;
  flen = set_lane<2>(1, flen);                            // strand
  flen = set_lane<3>(utn, flen);                          // unpadded target without PAM
  flen = set_lane<4>(10, flen); flen = set_lane<5>(10, flen);    // 10-base flanks of the hit (RH:227-228)
  flen = set_lane<6>(pu_len, flen);                       // pam_used; 7-10: variant_id, variant_description, variant_vcf, allele_frequency: None
  flen = set_lane<17>(n, flen); flen = set_lane<18>(n, flen); flen = set_lane<19>(n, flen);   // padded guide, alignment string, padded target
  flen = set_lane<20>(8, flen); flen = set_lane<21>(8, flen);    // 8-base flanks of the alignment (RH:243-244)
  flen = set_lane<22>(cigar_len, flen);
  val = set_lane<0>(h.gstart, val); val = set_lane<1>(h.gend, val); val = set_lane<11>(h.score, val);
  val = set_lane<12>(gmm, val);                           // guide_mm GA:103
  val = set_lane<13>(ggp, val);                           // guide_gaps GA:104
  val = set_lane<14>(gmm + ggp, val);                     // guide_mm_plus_gaps GA:105
  val = set_lane<15>(pam_mm, val);                        // pam_mm GA:106
  val = set_lane<16>(edits, val);                         // total_mm_plus_gaps = edits GA:101
  val = set_lane<23>(g.L, val);                           // unpadded_guide_sequence_length
  val = set_lane<24>(utn, val);
  const bool numeric = (0x0181F803u >> (lane & 31)) & (lane < 32 ? 1u : 0u);   // fields 0 1 11-16 23 24
  unsigned uval = (unsigned)(val < 0 ? -val : val);
  const int nd = 1 + (uval >= 10u) + (uval >= 100u) + (uval >= 1000u) + (uval >= 10000u) + (uval >= 100000u) + (uval >= 1000000u) +
                 (uval >= 10000000u) + (uval >= 100000000u) + (uval >= 1000000000u);
  if (numeric) flen = nd + (val < 0 ? 1 : 0);
  int incl = lane < MID_FIELDS ? flen + 1 : 0;            // inclusive prefix sum over the field lanes
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  const int foff = incl - (flen + 1);
  const int total = __builtin_amdgcn_readlane(incl, MID_FIELDS - 1);
  if (total > (int)a.mid_bound) return -1;
  if (!EMIT) return total;
  if (lane < MID_FIELDS) line[foff + flen] = '\t';
  if (numeric) {
    uint8_t* w = line + foff;
    if (val < 0) *w++ = '-';
    for (int i = nd - 1; i >= 0; i--) { const unsigned t = uval / 10u; w[i] = (uint8_t)('0' + (uval - 10u * t)); uval = t; }
  }
  auto off_of = [&](int f) { return __builtin_amdgcn_readlane(foff, f); };
  // bases [from, to) of the forward strand in guide orientation (flanks): a minus-strand hit reads them backwards
  auto put_bases = [&](int off, int from, int to) {
    if (lane < to - from) line[off + lane] = fwd[minus ? to - 1 - lane - lo : from + lane - lo];
  };
  const int o2 = off_of(2), o3 = off_of(3), o4 = off_of(4), o5 = off_of(5), o6 = off_of(6), o17 = off_of(17), o18 = off_of(18), o19 = off_of(19),
            o20 = off_of(20), o21 = off_of(21), o22 = off_of(22);
  if (lane == 0) line[o2] = minus ? '-' : '+';
  if (lane < utn) line[o3 + lane] = fwd[t_first + t_step * (ut0 + lane)];
  const int gs = h.gstart, ge = h.gend, as = h.start, ae = h.end;
  if (!minus) { put_bases(o4, gs - 10, gs); put_bases(o5, ge, ge + 10); put_bases(o20, as - 8, as); put_bases(o21, ae, ae + 8); }
  else        { put_bases(o4, ge, ge + 10); put_bases(o5, gs - 10, gs); put_bases(o20, ae, ae + 8); put_bases(o21, as - 8, as); }
  if (lane < pu_len) line[o6 + lane] = blob[a.rc.pu_off[r.pam + 1] + lane];
  if (valid) { line[o17 + lane] = (uint8_t)qc; line[o18 + lane] = (uint8_t)ac; line[o19 + lane] = (uint8_t)tc; }
  if (run_start) {
    uint8_t* w = line + o22 + 2 * bits_before_lane(RS) + bits_before_lane(RL);
    if (run_len >= 10) *w++ = (uint8_t)('0' + run_len / 10);
    *w++ = (uint8_t)('0' + run_len % 10);
    *w = (uint8_t)op_letter(op);
  }
  return total;
}

template <typename T>
hipError_t grow(T** p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  (void)hipFree(*p); *p = nullptr; cap = 0;
  need += need / 4;
  hipError_t e = hipMalloc((void**)p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  return e;
}


}  // namespace

// Length of the middle part of a row = what build_middle computes with ballots, for one lane: the field lengths of RH:210-254 from
// the alignment's op counts (GA:99-115, 139-183) and the run-length encoding of its cigar.  -1: the row builder does not lay it out.
__device__ __forceinline__ int middle_length(const RawAln* rp, const HitRec& h, int L, int pam_len, int pu_len, int n_max, int mid_bound) {
  const int ng = rp->n_ops, pam = rp->pam;
  const int gap = pam >= 0 ? rp->offset : 0;
  const uint32_t pam_x = rp->pam_x;
  const int n = ng + gap + pam_len;
  const int lo = min(h.start - 8, h.gstart - 10), hi = max(h.end + 8, h.gend + 10);
  if (n > n_max || n > MID_COLS || hi - lo > MID_FWD) return -1;
  const OpsWords ow = load_ops_words(rp->ops);
  const OpCounts oc = count_ops(ow, ng);
  const int utn = oc.not_ins - oc.lead_d - oc.trail_d;                 // target bases under the first .. last protospacer column (GA:111-115)
  const int gmm = oc.non_eq - oc.gaps, pam_mm = pam >= 0 ? __popc(pam_x) : 0;   // 'X' columns by the case of the query base (GA:103, 106)
  const int ggp = oc.gaps + gap;                                       // every gap column has a protospacer base on one side (GA:104, 168-182)
  const int edits = oc.non_eq + gap + pam_mm;                          // GA:101
  // Cigar.coalesce + toString over the columns: guide part (aligner order = traceback order reversed), the gap, the PAM; the number of
  // runs and of two-digit run lengths does not depend on the direction the columns are read in (5' PAM)
  int runs = 0, long_runs = 0, prev = -1, len = 0;
  for (int k = 0; k < n; k++) {
    const int op = k < ng ? ow.op(ng - 1 - k) : k < ng + gap ? 3 : (int)((pam_x >> ((k - ng - gap) & 15)) & 1u);
    if (op != prev) { if (len >= 10) long_runs++; runs++; len = 0; prev = op; }
    len++;
  }
  if (len >= 10) long_runs++;
  const int cigar_len = 2 * runs + long_runs;
  auto digits = [](int v) {
    const unsigned u = (unsigned)(v < 0 ? -v : v);
    return (v < 0 ? 1 : 0) + 1 + (u >= 10u) + (u >= 100u) + (u >= 1000u) + (u >= 10000u) + (u >= 100000u) + (u >= 1000000u) + (u >= 10000000u) +
           (u >= 100000000u) + (u >= 1000000000u);
  };
  const int total = MID_FIELDS + digits(h.gstart) + digits(h.gend) + 1 + utn + 10 + 10 + pu_len + digits(h.score) + digits(gmm) + digits(ggp) +
                    digits(gmm + ggp) + digits(pam_mm) + digits(edits) + 3 * n + 8 + 8 + cigar_len + digits(L) + digits(utn);
  return total > mid_bound ? -1 : total;
}

struct HitsWork {
  HitRec* hits = nullptr; size_t hits_cap = 0;
  uint64_t *keys = nullptr, *keys2 = nullptr, *lens = nullptr, *offs = nullptr;
  size_t keys_cap = 0, keys2_cap = 0, lens_cap = 0, offs_cap = 0;
  uint32_t *vals = nullptr, *vals2 = nullptr, *s_cs = nullptr, *wks = nullptr; size_t vals_cap = 0, vals2_cap = 0, cs_cap = 0, wks_cap = 0;
  int32_t *s_start = nullptr, *s_end = nullptr, *s_score = nullptr; size_t ss_cap = 0, se_cap = 0, sc_cap = 0;
  uint8_t *keep = nullptr, *head = nullptr; size_t keep_cap = 0, head_cap = 0;
  void* temp = nullptr; size_t temp_cap = 0;
  char* text = nullptr; size_t text_cap = 0;
  uint32_t* midlen = nullptr; size_t midlen_cap = 0;
  char* blob = nullptr; size_t blob_cap = 0;
  char* names = nullptr; size_t names_cap = 0;
  uint32_t* name_off = nullptr; size_t name_off_cap = 0;
  HitsExtKey* ext_keys = nullptr; size_t ext_keys_cap = 0;   // the caller's own hits of this call (HitsExt)
  uint64_t* ext_off = nullptr; size_t ext_off_cap = 0;
  char* ext_rows = nullptr; size_t ext_rows_cap = 0;
  uint8_t* ext_keep = nullptr; size_t ext_keep_cap = 0;      // HitsExt::rows_for: the walks' verdict per entry, and its page-locked copy
  uint8_t* h_ext_keep = nullptr; size_t h_ext_keep_cap = 0;
  uint64_t* ext_place = nullptr; size_t ext_place_cap = 0;   // HitsExtRows::fill_on_host: where every kept entry's row belongs in the text
  uint64_t* h_ext_place = nullptr; size_t h_ext_place_cap = 0;
  uint64_t* d_counts = nullptr;   // [0] text bytes, [1] low word: kept rows, high word: kept hits of the caller's own, [2] low word: flags
  uint64_t* h_counts = nullptr;   // pinned
  Mailbox mbox;                   // carries d_counts to the host (mailbox.hpp)
  RowConstDev rc{};               // set by hits_prepare
  size_t blob_bytes = 0;
  std::string blob_host;
  bool prepared = false;
};

}  // namespace calitas
