// hits.hip -- the tail of SearchReference.execute for the reference-genome branch, on the device: removeOverlaps
// (SearchReference.scala:653-675), ReferenceHit.sort (ReferenceHit.scala:284) and the 34-column hits.txt rows
// (ReferenceHit.scala:210-254), so that only finished text crosses PCIe.  SGA = SequentialGuideAligner.scala,
// GA = GuideAlignment.scala, RH = ReferenceHit.scala, SR = SearchReference.scala.
//
// Input: the accepted alignments of one guide, in calitas_search order, still on the device (select.hip's output).
//   1. hit_kernel      GuideAlignment coordinates (GA:21-31, SGA:260-313) and ReferenceHit.end (RH:135-138) per alignment; sort
//                      key = (contig, coordinate_start, strand, -score) = ReferenceHit.sort (RH:284); arrival order breaks ties
//                      (stable rocPRIM radix sort).  The hits of one (chromosome, strand) group -- what removeOverlaps works on,
//                      SR:656 -- are a subsequence of that order, and inside the group the order is the group's own sort order.
//   2. prep_kernel     per sorted position: start / end / score / (contig, strand), and whether removeOverlaps' walk restarts
//                      here.  The walk carries one "current hit" through a group; wherever a hit starts at or beyond (largest end
//                      so far - maxOverlap + 1) no earlier hit of the group can overlap it by >= maxOverlap, the walk keeps its
//                      current hit and restarts with a clean state.  Hits are at most CALITAS_MAX_OPS long, so "largest end so
//                      far" only needs a look back over the hits that start within that distance.
//   3. cluster_kernel  the restart points cut every group into clusters (usually 1-3 hits) that are independent of each other;
//                      one lane walks one cluster with the reference's loop (SR:661-671), stepping over the other strand's hits.
//   4. mid_kernel / out_kernel: rows of the kept hits, already in final order (dropped hits have length 0).
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

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

__global__ void hit_kernel(const RawAln* fin, uint32_t n, const GuideDev* guides, const uint64_t* win_base, const int2* win,
                           int score_hi, HitRec* hits, uint64_t* keys, uint32_t* vals, uint32_t* wks, uint32_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RawAln* rp = fin + i;
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
  hits[i] = h;
  int sb = score_hi - r.score;
  if (sb < 0 || sb >= (1 << SCORE_BITS) || h.gstart < 0) { atomicOr(flags, HITS_FLAG_SCORE_RANGE); sb = 0; }
  keys[i] = ((uint64_t)r.contig << 46) | ((uint64_t)(uint32_t)h.gstart << 15) | ((uint64_t)h.minus << 14) | (uint64_t)sb;
  vals[i] = i;
  wks[i] = r.window_k;
}

// ReferenceHit.sort without a sort.  The accepted alignments arrive in (contig, window, ...) order, and hits of two windows that
// share no base cannot be out of order relative to each other (a hit starts inside its window): with reach = the number of
// window steps after which two windows are disjoint, hit i only has to be compared with the hits of the windows less than
// `reach` steps away.  rank(i) = i - (earlier neighbours with a greater key) + (later neighbours with a smaller key); equal keys
// keep their arrival order, as a stable sort would.  One launch instead of the seven of a 64-bit radix / merge sort; the caller
// takes the general sort when a window holds more records than a lane should walk past.
__global__ void rank_kernel(const uint64_t* keys, const uint32_t* wks, uint32_t n, uint32_t reach, uint32_t* order) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  const uint64_t contig = k >> 46;
  const uint32_t wk = wks[i];
  uint32_t rank = i;
  for (uint32_t j = i; j-- > 0;) {
    const uint64_t kj = keys[j];
    if ((kj >> 46) != contig || wk - wks[j] >= reach) break;
    rank -= kj > k;
  }
  for (uint32_t j = i + 1; j < n; j++) {
    const uint64_t kj = keys[j];
    if ((kj >> 46) != contig || wks[j] - wk >= reach) break;
    rank += kj < k;
  }
  order[rank] = i;
}

constexpr int HIT_MAX_LEN = CALITAS_MAX_OPS;   // a hit covers at most this many reference bases (ReferenceHit.end - start + 1)

__global__ void prep_kernel(const HitRec* hits, const uint32_t* order, uint32_t n, int max_overlap, int32_t* s_start, int32_t* s_end,
                            int32_t* s_score, uint32_t* s_cs, uint8_t* head, uint8_t* keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const HitRec h = hits[order[i]];
  const uint32_t cs = ((uint32_t)h.contig << 1) | h.minus;
  s_start[i] = h.gstart; s_end[i] = h.rh_end; s_score[i] = h.score; s_cs[i] = cs;
  keep[i] = 0;
  // restart point?  look back over the hits of this contig that start close enough to reach maxOverlap bases into this one
  bool is_head = true;
  for (uint32_t j = i; j-- > 0;) {
    const HitRec p = hits[order[j]];
    if (p.contig != h.contig || p.gstart + HIT_MAX_LEN - 1 - h.gstart < max_overlap) break;
    if (p.minus == h.minus && p.rh_end - h.gstart >= max_overlap) { is_head = false; break; }
  }
  head[i] = is_head ? 1 : 0;
}

__global__ void cluster_kernel(const int32_t* s_start, const int32_t* s_end, const int32_t* s_score, const uint32_t* s_cs,
                               const uint8_t* head, uint32_t n, int max_overlap, uint8_t* keep, uint32_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const uint32_t cs = s_cs[i];
  // next hit of this (contig, strand) group after position j, or n
  auto next = [&](uint32_t j) {
    for (j++; j < n; j++) {
      const uint32_t c = s_cs[j];
      if (c == cs) return j;
      if ((c >> 1) != (cs >> 1)) break;
    }
    return n;
  };
  uint32_t j = i, steps = 0;
  for (;;) {                                            // SR:661-671
    const uint32_t hit = j;
    j = next(j);
    const int hs = s_start[hit], he = s_end[hit], hsc = s_score[hit];
    bool more = false;
    int ov = 0;
    for (;;) {
      more = j < n && !head[j];
      if (!more) break;
      ov = max(0, min(s_end[j], he) - max(s_start[j], hs));   // RH:141-144
      if (!(ov >= max_overlap && s_score[j] <= hsc)) break;
      j = next(j);
      if (++steps > CLUSTER_MAX) break;
    }
    if (steps > CLUSTER_MAX) { atomicOr(flags, HITS_FLAG_CLUSTER); return; }
    if (!more || ov < max_overlap) keep[hit] = 1;
    if (!more) return;
    if (++steps > CLUSTER_MAX) { atomicOr(flags, HITS_FLAG_CLUSTER); return; }
  }
}

// ---- rows ------------------------------------------------------------------------------------------------------------
// A row is  head | chromosome \t | middle | tail  where head and tail are the same for every row of the call.
//   mid_kernel: one lane per row builds the middle part in its own LDS slot (byte writes, slot stride an odd number of
//               words so the 64 lanes hit 64 banks); the wave then copies the slots to a fixed-stride staging buffer with
//               coalesced dword stores and records the row length.
//   out_kernel: after the exclusive scan of the lengths, a wave assembles row after row at its final offset with
//               coalesced byte stores (head / tail come from LDS).
// Working arrays (ops, padded strings, the fetched reference span) live in the lane's LDS slot behind the output area,
// never in private memory.

__device__ __forceinline__ char comp_base(char c) {   // fgbio Sequences.complement on an upper-case base
  switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'U': return 'A';
    case 'M': return 'K'; case 'K': return 'M'; case 'R': return 'Y'; case 'Y': return 'R';
    case 'V': return 'B'; case 'B': return 'V'; case 'H': return 'D'; case 'D': return 'H';
    default: return c;
  }
}

// "ACGT"[code] and "=XID"[op] from a constant in a register: indexing the string literals is a load from constant memory per
// character, i.e. a dependent round trip per base in loops every lane of the wave runs in lockstep (half of mid_kernel's time).
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

__device__ __forceinline__ uint8_t* put_int(uint8_t* w, int v) {
  if (v < 0) { *w++ = '-'; v = -v; }
  unsigned u = (unsigned)v;
  int nd = 1;
  for (unsigned t = u; t >= 10; t /= 10) nd++;
  for (int i = nd - 1; i >= 0; i--) { w[i] = (uint8_t)('0' + u % 10); u /= 10; }
  return w + nd;
}

// n bytes from one place of the lane's LDS slot to another (they never overlap), eight at a time: the reads of a group are issued
// together and waited for once.  Byte by byte every read is a round trip to the LDS that the next write waits for (~100 cycles each,
// and a row copies ~250 bytes): the compiler cannot batch them itself because it must assume the two pointers alias.
__device__ __forceinline__ uint8_t* copy_bytes(uint8_t* __restrict__ w, const uint8_t* __restrict__ src, int n) {
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    uint8_t b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = src[i + k];
#pragma unroll
    for (int k = 0; k < 8; k++) w[i + k] = b[k];
  }
  if (i < n) {
    uint8_t b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = (i + k < n) ? src[i + k] : (uint8_t)0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (i + k < n) w[i + k] = b[k];
  }
  return w + n;
}

// Bases [from, to) of a forward-strand buffer that starts at contig offset lo, in guide orientation (flanks: 8 or 10 bases).
__device__ __forceinline__ uint8_t* put_bases(uint8_t* __restrict__ w, const uint8_t* __restrict__ fwd, int lo, int from, int to, bool minus) {
  const int n = to - from;
  if (!minus) return copy_bytes(w, fwd + (from - lo), n);
  for (int i = 0; i < n; i += 8) {
    uint8_t b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = (i + k < n) ? fwd[to - 1 - (i + k) - lo] : (uint8_t)0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (i + k < n) w[i + k] = (uint8_t)comp_base((char)b[k]);
  }
  return w + n;
}

__device__ __forceinline__ bool is_lower(char c) { return c >= 'a' && c <= 'z'; }
__device__ __forceinline__ bool is_letter(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

// GuideAlignment.count (GA:139-163)
__device__ int ga_count(const uint8_t* pg, const uint8_t* pa, int len, bool lower, bool both_sides, bool mms, bool gaps) {
  int n = 0;
  for (int i = 0; i < len; i++) {
    if (mms && pa[i] == '.' && is_lower((char)pg[i]) == lower) { n++; continue; }
    if (!(gaps && pa[i] == '~')) continue;
    const char gb = (char)pg[i];
    bool me = gb != '-' && is_lower(gb) == lower;
    if (!me) {
      int pi = i; while (pi > 0 && pg[pi] == '-') pi--;            // previousNonDash GA:168-172
      int ni = i; while (ni < len - 1 && pg[ni] == '-') ni++;      // nextNonDash GA:177-182
      const char prev = (char)pg[pi], next = (char)pg[ni];
      if (both_sides) me = (prev == '-' || is_lower(prev) == lower) && (next == '-' || is_lower(next) == lower);
      else me = (is_letter(prev) && is_lower(prev) == lower) || (is_letter(next) && is_lower(next) == lower);
    }
    if (me) n++;
  }
  return n;
}

// words of the packed reference a row's span can touch (alignment + flanks <= CALITAS_MAX_OPS + 20 bases), staged per lane
constexpr int MID_CODE_WORDS = (CALITAS_MAX_OPS + 20) / 16 + 2, MID_MASK_WORDS = (CALITAS_MAX_OPS + 20) / 32 + 2;
constexpr int MID_WORDS = MID_CODE_WORDS + MID_MASK_WORDS;

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
  uint32_t slot_bytes;       // LDS bytes per lane: multiple of 4, odd number of words
  uint32_t mid_bound;        // bytes reserved for the middle part = staging stride
  uint32_t n_max;            // most padded columns a row of this search can have
  uint32_t blob_bytes;       // constant strings, copied to LDS by each block
  uint32_t* n_rows;          // out: number of live rows
};

static_assert(offsetof(RawAln, ops) % 4 == 0 && sizeof(RawAln) % 4 == 0, "RawAln::ops must be word aligned");

struct RowIn {             // the fields of one RawAln a row needs, ops as five words (2 bits per op, traceback order)
  uint32_t w[RAW_MAX_OPS / 16];
  int n_ops, pam, offset;
  uint32_t pam_x;
};

__device__ __forceinline__ int row_op(const RowIn& r, int i) {
  const int k = i >> 4;
  uint32_t w = r.w[0];
#pragma unroll
  for (int j = 1; j < RAW_MAX_OPS / 16; j++) w = (k == j) ? r.w[j] : w;
  return (int)((w >> ((i & 15) * 2)) & 3u);
}

// The middle part of one hits.txt row (RH:210-254, columns coordinate_start .. unpadded_target_sequence_length) at `out`;
// `scratch` holds 4 * n_max + n_max + 24 bytes and, word aligned behind them, MID_WORDS words; `blob` is the LDS copy of the constant strings.  Returns its length, or -1
// when the alignment has more columns than n_max.
__device__ int format_middle(uint8_t* out, uint8_t* scratch, const MidArgs& a, const uint8_t* blob, const RowIn& r, const HitRec& h,
                             const GuideDev& g) {
  uint8_t* ops = scratch;
  uint8_t* pg = ops + a.n_max;
  uint8_t* pa = pg + a.n_max;
  uint8_t* pt = pa + a.n_max;
  uint8_t* fwd = pt + a.n_max;
  const int pam_len = r.pam >= 0 ? g.pam_len[r.pam] : 0, gap = r.pam >= 0 ? r.offset : 0;
  const int ng = r.n_ops, n = ng + gap + pam_len;
  if (n > (int)a.n_max) return -1;
  // ops in guide orientation: guide part (stored in traceback order), gap to the PAM, PAM (SGA:472-476); reversed for a
  // 5' PAM (SGA:267-269)
  for (int i = 0; i < n; i++) {
    const int k = g.pam5 ? n - 1 - i : i;
    char op;
    if (k < ng) op = op_letter(row_op(r, ng - 1 - k));
    else if (k < ng + gap) op = 'D';
    else op = ((r.pam_x >> (k - ng - gap)) & 1) ? 'X' : '=';
    ops[i] = (uint8_t)op;
  }
  const bool minus = h.minus != 0;
  const uint8_t* q = blob + a.rc.q_off[r.pam + 1];
  // one fetch covers the alignment and all four flanks (RH:213-216)
  const int lo = min(h.start - 8, h.gstart - 10), hi = max(h.end + 8, h.gend + 10);
  {
    // The code words (16 bases each) and mask words (32 bases each) of the span are fetched up front into the lane's scratch -- a
    // dozen independent loads and one wait.  Fetching them inside the loop as it reached a new word made every iteration of the
    // wave a dependent round trip to memory (the lanes cross word boundaries at different bases): ~100 of them per wave, half of
    // the kernel's time.
    const ContigInfo c = a.ref.contigs[h.contig];
    uint32_t* wsc = reinterpret_cast<uint32_t*>(scratch + ((5 * a.n_max + 24 + 3) & ~3u));   // MID_WORDS words behind fwd[]
    const int p0 = max(lo, 0), p1 = (int)min((int64_t)hi, (int64_t)c.len);
    uint64_t cw0 = 0, mw0 = 0;
    if (p0 < p1) {
      const uint64_t g0 = c.gbase + (uint64_t)p0, g1 = c.gbase + (uint64_t)p1 - 1;
      cw0 = g0 >> 4; mw0 = g0 >> 5;
      const int ncw = (int)((g1 >> 4) - cw0) + 1, nmw = (int)((g1 >> 5) - mw0) + 1;
      for (int k = 0; k < ncw && k < MID_CODE_WORDS; k++) wsc[k] = a.ref.codes[cw0 + k];
      for (int k = 0; k < nmw && k < MID_MASK_WORDS; k++) wsc[MID_CODE_WORDS + k] = a.ref.mask[mw0 + k];
    }
    for (int p = lo; p < hi; p++) {
      char b = 'N';                                                                          // RH:262-264
      if (p >= p0 && p < p1) {
        const uint64_t gpos = c.gbase + (uint64_t)p;
        const uint32_t mw = wsc[MID_CODE_WORDS + (int)((gpos >> 5) - mw0)], cw = wsc[(int)((gpos >> 4) - cw0)];
        b = ((mw >> (gpos & 31)) & 1u) ? base_upper_dev(a.ref, gpos) : base_letter((cw >> ((gpos & 15) * 2)) & 3u);
      }
      fwd[p - lo] = (uint8_t)b;
    }
  }
  int qi = 0, mm = 0, gp = 0, ps = -1, pe = -1;
  int tp = minus ? h.end - 1 : h.start;                 // next target base, walking in guide orientation
  for (int i = 0; i < n; i++) {                         // Alignment.paddedString (SGA:511)
    const char op = (char)ops[i];
    char tb = '-', qc = '-';
    if (op != 'I') { tb = (char)fwd[tp - lo]; if (minus) { tb = comp_base(tb); tp--; } else tp++; }
    if (op != 'D') qc = (char)q[qi++];
    pg[i] = (uint8_t)qc; pt[i] = (uint8_t)tb;
    pa[i] = (uint8_t)(op == '=' ? '|' : op == 'X' ? '.' : '~');
    mm += op == 'X'; gp += (op == 'I' || op == 'D');
    if (qc >= 'A' && qc <= 'Z') { if (ps < 0) ps = i; pe = i; }   // unpaddedTargetWithoutPam GA:111-115
  }
  uint8_t* w = out;
  w = put_int(w, h.gstart); *w++ = '\t';
  w = put_int(w, h.gend); *w++ = '\t';
  *w++ = minus ? '-' : '+'; *w++ = '\t';
  int utn = 0;
  for (int i = ps; i >= 0 && i <= pe; i++) if (pt[i] != '-') { *w++ = pt[i]; utn++; }
  *w++ = '\t';
  const int gs = h.gstart, ge = h.gend, as = h.start, ae = h.end;
  if (!minus) { w = put_bases(w, fwd, lo, gs - 10, gs, false); *w++ = '\t'; w = put_bases(w, fwd, lo, ge, ge + 10, false); }   // RH:227-228
  else        { w = put_bases(w, fwd, lo, ge, ge + 10, true);  *w++ = '\t'; w = put_bases(w, fwd, lo, gs - 10, gs, true); }
  *w++ = '\t';
  w = copy_bytes(w, blob + a.rc.pu_off[r.pam + 1], (int)a.rc.pu_len[r.pam + 1]);
  *w++ = '\t';
  *w++ = '\t'; *w++ = '\t'; *w++ = '\t'; *w++ = '\t';   // variant_id, variant_description, variant_vcf, allele_frequency: None
  w = put_int(w, h.score); *w++ = '\t';
  const int gmm = ga_count(pg, pa, n, false, false, true, false);    // guide_mm GA:103
  const int ggp = ga_count(pg, pa, n, false, false, false, true);    // guide_gaps GA:104
  w = put_int(w, gmm); *w++ = '\t';
  w = put_int(w, ggp); *w++ = '\t';
  w = put_int(w, gmm + ggp); *w++ = '\t';               // guide_mm_plus_gaps GA:105
  w = put_int(w, ga_count(pg, pa, n, true, true, true, false)); *w++ = '\t';   // pam_mm GA:106
  w = put_int(w, mm + gp); *w++ = '\t';                 // total_mm_plus_gaps = edits GA:101
  w = copy_bytes(w, pg, n);
  *w++ = '\t';
  w = copy_bytes(w, pa, n);
  *w++ = '\t';
  w = copy_bytes(w, pt, n);
  *w++ = '\t';
  if (!minus) { w = put_bases(w, fwd, lo, as - 8, as, false); *w++ = '\t'; w = put_bases(w, fwd, lo, ae, ae + 8, false); }     // RH:243-244
  else        { w = put_bases(w, fwd, lo, ae, ae + 8, true);  *w++ = '\t'; w = put_bases(w, fwd, lo, as - 8, as, true); }
  *w++ = '\t';
  for (int i = 0; i < n;) {                             // Cigar.coalesce + toString
    int j = i;
    while (j < n && ops[j] == ops[i]) j++;
    w = put_int(w, j - i); *w++ = ops[i]; i = j;
  }
  *w++ = '\t';
  w = put_int(w, g.L); *w++ = '\t';                     // unpadded_guide_sequence_length
  w = put_int(w, utn); *w++ = '\t';
  return (int)(w - out);
}

__global__ __launch_bounds__(64) void mid_kernel(MidArgs a, uint8_t* stage, uint32_t* midlen, uint64_t* lens, uint32_t* flags) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t lane = threadIdx.x, row0 = blockIdx.x * 64, k = row0 + lane;
  int len = 0;
  uint32_t name_len = 0;
  const bool live = k < a.n && a.keep[k] != 0;
  // constant strings (queries, PAMs) into LDS behind the 64 slots
  uint8_t* blob = lds + 64 * a.slot_bytes;
  for (uint32_t i = lane; i < a.blob_bytes; i += 64) blob[i] = (uint8_t)a.blob[i];
  __syncthreads();
  if (live) {
    const uint32_t v = a.order[k];
    const RawAln* rp = a.fin + v;
    RowIn r;
    const uint32_t* ow = reinterpret_cast<const uint32_t*>(rp->ops);   // RawAln::ops sits at a 4-byte aligned offset
#pragma unroll
    for (int j = 0; j < RAW_MAX_OPS / 16; j++) r.w[j] = ow[j];
    r.n_ops = rp->n_ops; r.pam = rp->pam; r.offset = rp->offset; r.pam_x = rp->pam_x;
    const uint32_t guide = rp->guide;
    const HitRec h = a.hits[v];
    uint8_t* out = lds + lane * a.slot_bytes;
    len = format_middle(out, out + a.mid_bound, a, blob, r, h, a.guides[guide]);
    if (len < 0 || len > (int)a.mid_bound) { atomicOr(flags, HITS_FLAG_ROW); len = 0; }
    name_len = a.name_off[h.contig + 1] - a.name_off[h.contig];
  }
  {
    const unsigned long long lv = __ballot(live);
    if (lane == 0 && lv) atomicAdd(a.n_rows, (uint32_t)__popcll(lv));
  }
  if (k < a.n) {
    midlen[k] = (uint32_t)len;
    lens[k] = live ? (uint64_t)(a.rc.head_len + name_len + 1 + (uint32_t)len + a.rc.tail_len) : 0;
  }
  __syncthreads();
  for (int r = 0; r < 64; r++) {
    const int l = __shfl(len, r);
    const uint32_t* src = (const uint32_t*)(lds + r * a.slot_bytes);
    uint32_t* dst = (uint32_t*)(stage + (size_t)(row0 + r) * a.mid_bound);
    for (int wd = lane; wd < (l + 3) / 4; wd += 64) dst[wd] = src[wd];
  }
}

struct OutArgs {
  RowConstDev rc;
  const char* blob;
  const char* names;
  const uint32_t* name_off;
  const HitRec* hits;
  const uint32_t* order;
  const uint32_t* midlen;
  const uint64_t* offs;
  const uint8_t* stage;
  uint32_t n, mid_bound;       // n sorted positions; dropped hits have midlen 0
};

constexpr int OUT_ROWS_PER_WAVE = 8;

__global__ __launch_bounds__(256) void out_kernel(OutArgs a, char* text) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // head | tail
  for (uint32_t i = threadIdx.x; i < a.rc.head_len; i += 256) lds[i] = (uint8_t)a.blob[a.rc.head_off + i];
  for (uint32_t i = threadIdx.x; i < a.rc.tail_len; i += 256) lds[a.rc.head_len + i] = (uint8_t)a.blob[a.rc.tail_off + i];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint8_t* head = lds;
  const uint8_t* tail = lds + a.rc.head_len;
  for (int rr = 0; rr < OUT_ROWS_PER_WAVE; rr++) {
    const uint32_t k = wave * OUT_ROWS_PER_WAVE + rr;
    if (k >= a.n) return;
    if (a.midlen[k] == 0) continue;                     // dropped by removeOverlaps
    const uint32_t contig = (uint32_t)a.hits[a.order[k]].contig;
    const uint32_t nb = a.name_off[contig], nl = a.name_off[contig + 1] - nb;
    const uint32_t s0 = a.rc.head_len, s1 = s0 + nl + 1, s2 = s1 + a.midlen[k], total = s2 + a.rc.tail_len;
    const uint8_t* mid = a.stage + (size_t)k * a.mid_bound;
    char* dst = text + a.offs[k];
    for (uint32_t b = lane; b < total; b += 64) {
      uint8_t c;
      if (b < s0) c = head[b];
      else if (b < s1) c = (b - s0 < nl) ? (uint8_t)a.names[nb + b - s0] : (uint8_t)'\t';
      else if (b < s2) c = mid[b - s1];
      else c = tail[b - s2];
      dst[b] = (char)c;
    }
  }
}

__global__ void total_kernel(const uint64_t* offs, const uint64_t* lens, uint32_t n, uint64_t* total) { *total = offs[n - 1] + lens[n - 1]; }

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

struct HitsWork {
  HitRec* hits = nullptr; size_t hits_cap = 0;
  uint64_t *keys = nullptr, *keys2 = nullptr, *lens = nullptr, *offs = nullptr;
  size_t keys_cap = 0, keys2_cap = 0, lens_cap = 0, offs_cap = 0;
  uint32_t *vals = nullptr, *vals2 = nullptr, *s_cs = nullptr, *wks = nullptr; size_t vals_cap = 0, vals2_cap = 0, cs_cap = 0, wks_cap = 0;
  int32_t *s_start = nullptr, *s_end = nullptr, *s_score = nullptr; size_t ss_cap = 0, se_cap = 0, sc_cap = 0;
  uint8_t *keep = nullptr, *head = nullptr; size_t keep_cap = 0, head_cap = 0;
  void* temp = nullptr; size_t temp_cap = 0;
  char* text = nullptr; size_t text_cap = 0;
  uint8_t* stage = nullptr; size_t stage_cap = 0;
  uint32_t* midlen = nullptr; size_t midlen_cap = 0;
  char* blob = nullptr; size_t blob_cap = 0;
  char* names = nullptr; size_t names_cap = 0;
  uint32_t* name_off = nullptr; size_t name_off_cap = 0;
  uint64_t* d_counts = nullptr;   // [0] text bytes, [1] low word: kept rows, [2] low word: flags
  uint64_t* h_counts = nullptr;   // pinned
  Mailbox mbox;                   // carries d_counts to the host (mailbox.hpp)
  RowConstDev rc{};               // set by hits_prepare
  size_t blob_bytes = 0;
  std::string blob_host;
  bool prepared = false;
};

void hits_destroy(HitsWork* w) {
  if (!w) return;
  (void)hipFree(w->hits); (void)hipFree(w->keys); (void)hipFree(w->keys2); (void)hipFree(w->lens); (void)hipFree(w->offs);
  (void)hipFree(w->vals); (void)hipFree(w->vals2); (void)hipFree(w->s_cs); (void)hipFree(w->wks); (void)hipFree(w->s_start); (void)hipFree(w->s_end);
  (void)hipFree(w->s_score); (void)hipFree(w->keep); (void)hipFree(w->head); (void)hipFree(w->temp); (void)hipFree(w->text);
  (void)hipFree(w->stage); (void)hipFree(w->midlen); (void)hipFree(w->blob); (void)hipFree(w->names); (void)hipFree(w->name_off);
  (void)hipFree(w->d_counts);
  if (w->h_counts) (void)hipHostFree(w->h_counts);
  mailbox_close(w->mbox);
  delete w;
}

bool hits_supported(uint64_t n_contigs, int max_overlap, int score_lo, int score_hi) {
  return n_contigs < (1ull << 18) - 1 && max_overlap >= 1 && score_hi >= score_lo && (int64_t)score_hi - score_lo < (1 << SCORE_BITS);
}

#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)

hipError_t hits_set_names(HitsWork** pw, const std::vector<std::string>& names, hipStream_t stream) {
  if (!*pw) *pw = new HitsWork();
  HitsWork& w = **pw;
  hipError_t e;
  std::string blob;
  std::vector<uint32_t> off(names.size() + 1, 0);
  for (size_t i = 0; i < names.size(); i++) { blob += names[i]; off[i + 1] = (uint32_t)blob.size(); }
  TRY(grow(&w.names, w.names_cap, std::max<size_t>(1, blob.size())));
  TRY(grow(&w.name_off, w.name_off_cap, off.size()));
  // on the stream of the kernels that read them (a lane's stream is non-blocking: nothing orders it against the null stream)
  if (!blob.empty()) TRY(hipMemcpyAsync(w.names, blob.data(), blob.size(), hipMemcpyHostToDevice, stream));
  TRY(hipMemcpyAsync(w.name_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  return hipStreamSynchronize(stream);     // blob / off are locals
}

// The per-call constants (row pieces, cleared counters), queued on `stream` ahead of everything else of the call so that nothing of
// it sits between the filter stage and hit_kernel.  The host copy of the strings lives in the work until the next call.
hipError_t hits_prepare(HitsWork** pw, const RowStrings& st, hipStream_t stream) {
  if (!*pw) *pw = new HitsWork();
  HitsWork& w = **pw;
  hipError_t e;
  if (!w.d_counts) TRY(hipMalloc((void**)&w.d_counts, 3 * sizeof(uint64_t)));
  if (!w.h_counts) TRY(hipHostMalloc((void**)&w.h_counts, 3 * sizeof(uint64_t), hipHostMallocDefault));
  TRY(hipMemsetAsync(w.d_counts, 0, 3 * sizeof(uint64_t), stream));
  RowConstDev rc{};
  std::string& blob = w.blob_host;
  blob.clear();
  auto add = [&](const std::string& s, uint32_t& off, uint32_t& len) { off = (uint32_t)blob.size(); len = (uint32_t)s.size(); blob += s; };
  add(st.head, rc.head_off, rc.head_len); add(st.tail, rc.tail_off, rc.tail_len); add(st.proto_len, rc.plen_off, rc.plen_len);
  for (size_t i = 0; i < st.query.size() && i <= (size_t)MAX_PAMS; i++) {
    add(st.query[i], rc.q_off[i], rc.q_len[i]);
    add(st.pam_used[i], rc.pu_off[i], rc.pu_len[i]);
  }
  TRY(grow(&w.blob, w.blob_cap, blob.size() + 1));
  TRY(hipMemcpyAsync(w.blob, blob.data(), blob.size(), hipMemcpyHostToDevice, stream));
  w.rc = rc; w.blob_bytes = blob.size(); w.prepared = true;
  return hipSuccess;
}

hipError_t hits_run(HitsWork** pw, const HitsRef& ref, const RawAln* d_final, uint32_t n_in, const GuideDev* d_guides,
                    const uint64_t* d_win_base, const int2* d_win, const RowStrings& st, int max_overlap, int score_hi,
                    int max_ops, uint32_t window_reach, hipStream_t stream, HitsResult* res) {
  if (!*pw) *pw = new HitsWork();
  HitsWork& w = **pw;
  hipError_t e;
  *res = HitsResult{};
  const size_t n = n_in;
  if (n == 0) return hipSuccess;
  if (!w.prepared) TRY(hits_prepare(pw, st, stream));  // normally done at the start of the call
  w.prepared = false;
  const RowConstDev rc = w.rc;
  const size_t blob_bytes = w.blob_bytes;
  uint32_t* d_kept = (uint32_t*)(w.d_counts + 1);
  uint32_t* d_flags = (uint32_t*)(w.d_counts + 2);

  TRY(grow(&w.hits, w.hits_cap, n)); TRY(grow(&w.keys, w.keys_cap, n)); TRY(grow(&w.keys2, w.keys2_cap, n));
  TRY(grow(&w.vals, w.vals_cap, n)); TRY(grow(&w.vals2, w.vals2_cap, n)); TRY(grow(&w.s_cs, w.cs_cap, n)); TRY(grow(&w.wks, w.wks_cap, n));
  TRY(grow(&w.lens, w.lens_cap, n)); TRY(grow(&w.offs, w.offs_cap, n)); TRY(grow(&w.s_start, w.ss_cap, n));
  TRY(grow(&w.s_end, w.se_cap, n)); TRY(grow(&w.s_score, w.sc_cap, n)); TRY(grow(&w.keep, w.keep_cap, n)); TRY(grow(&w.head, w.head_cap, n));
  size_t t1 = 0, t3 = 0;
  if (window_reach == 0) TRY(rocprim::radix_sort_pairs(nullptr, t1, w.keys, w.keys2, w.vals, w.vals2, n, 0, 64, stream));
  TRY(rocprim::exclusive_scan(nullptr, t3, w.lens, w.offs, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream));
  {
    const size_t need = std::max(t1, t3);
    if (need > w.temp_cap) { (void)hipFree(w.temp); w.temp = nullptr; w.temp_cap = 0; TRY(hipMalloc(&w.temp, need)); w.temp_cap = need; }
  }
  const dim3 block(256), grid((unsigned)((n + 255) / 256));
  size_t ts;
  // 1: coordinates and the final order
  hipLaunchKernelGGL(hit_kernel, grid, block, 0, stream, d_final, n_in, d_guides, d_win_base, d_win, score_hi, w.hits, w.keys, w.vals, w.wks, d_flags);
  if (window_reach) {   // the order by counting among the neighbouring windows (rank_kernel)
    hipLaunchKernelGGL(rank_kernel, grid, block, 0, stream, (const uint64_t*)w.keys, (const uint32_t*)w.wks, n_in, window_reach, w.vals2);
  } else {              // a window is too crowded for that: the general stable sort
    ts = w.temp_cap;
    TRY(rocprim::radix_sort_pairs(w.temp, ts, w.keys, w.keys2, w.vals, w.vals2, n, 0, 64, stream));
  }
  // 2-3: removeOverlaps
  hipLaunchKernelGGL(prep_kernel, grid, block, 0, stream, (const HitRec*)w.hits, (const uint32_t*)w.vals2, n_in, max_overlap, w.s_start, w.s_end,
                     w.s_score, w.s_cs, w.head, w.keep);
  hipLaunchKernelGGL(cluster_kernel, grid, block, 0, stream, (const int32_t*)w.s_start, (const int32_t*)w.s_end, (const int32_t*)w.s_score,
                     (const uint32_t*)w.s_cs, (const uint8_t*)w.head, n_in, max_overlap, w.keep, d_flags);
  // 4: rows
  const uint32_t n_max = (uint32_t)std::min<int>(CALITAS_MAX_OPS, std::max(1, max_ops));
  const uint32_t mid_bound = (6 * n_max + 128 + 3) & ~3u;
  uint32_t slot = mid_bound + ((5 * n_max + 24 + 3) & ~3u) + 4 * MID_WORDS;   // output | ops, padded strings, fetched bases | staged reference words
  if (((slot / 4) & 1) == 0) slot += 4;
  const uint32_t mid_lds = 64 * slot + (uint32_t)((blob_bytes + 15) & ~(size_t)15);
  if (mid_lds > 64 * 1024) {   // beyond the default dynamic LDS limit: ask for more (160 KB per CU on gfx950) or decline
    if (mid_lds > 160 * 1024 ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds) != hipSuccess) {
      (void)hipGetLastError();
      TRY(hipStreamSynchronize(stream));
      res->flags = HITS_FLAG_ROW;
      return hipSuccess;
    }
  }
  const size_t n_pad = (n + 63) / 64 * 64;
  TRY(grow(&w.stage, w.stage_cap, n_pad * mid_bound));
  TRY(grow(&w.midlen, w.midlen_cap, n_pad));
  MidArgs ma{};
  ma.ref = ref; ma.rc = rc; ma.blob = w.blob; ma.name_off = w.name_off; ma.fin = d_final; ma.hits = w.hits; ma.guides = d_guides;
  ma.keep = w.keep; ma.order = w.vals2; ma.n = n_in; ma.n_rows = d_kept; ma.slot_bytes = slot; ma.mid_bound = mid_bound; ma.n_max = n_max; ma.blob_bytes = (uint32_t)blob_bytes;
  hipLaunchKernelGGL(mid_kernel, dim3((unsigned)(n_pad / 64)), dim3(64), mid_lds, stream, ma, w.stage, w.midlen, w.lens, d_flags);
  ts = w.temp_cap;
  TRY(rocprim::exclusive_scan(w.temp, ts, w.lens, w.offs, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream));
  hipLaunchKernelGGL(total_kernel, dim3(1), dim3(1), 0, stream, (const uint64_t*)w.offs, (const uint64_t*)w.lens, n_in, w.d_counts);
  TRY(mailbox_post(w.mbox, reinterpret_cast<const uint32_t*>(w.d_counts), 6, stream));
  TRY(mailbox_wait(w.mbox, stream));
  for (int k = 0; k < 3; k++) w.h_counts[k] = (uint64_t)w.mbox.host[1 + 2 * k] | ((uint64_t)w.mbox.host[2 + 2 * k] << 32);
  TRY(hipGetLastError());
  res->flags = (uint32_t)w.h_counts[2];
  if (res->flags) return hipSuccess;
  res->n_rows = (uint32_t)w.h_counts[1];
  res->text_bytes = w.h_counts[0];
  TRY(grow(&w.text, w.text_cap, std::max<size_t>(1, (size_t)res->text_bytes)));
  if (res->n_rows) {
    OutArgs oa{};
    oa.rc = rc; oa.blob = w.blob; oa.names = w.names; oa.name_off = w.name_off; oa.hits = w.hits; oa.order = w.vals2; oa.midlen = w.midlen;
    oa.offs = w.offs; oa.stage = w.stage; oa.n = n_in; oa.mid_bound = mid_bound;
    const unsigned rows_per_block = 4 * OUT_ROWS_PER_WAVE;
    hipLaunchKernelGGL(out_kernel, dim3((n_in + rows_per_block - 1) / rows_per_block), dim3(256), rc.head_len + rc.tail_len, stream, oa, w.text);
  }
  TRY(hipGetLastError());
  res->d_text = w.text;
  return hipSuccess;
}

#undef TRY

}  // namespace calitas
