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
//   4. len_kernel / rows_kernel: rows of the kept hits, already in final order (dropped hits have length 0): lengths, offsets, then
//                      every row built at its final place.
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
#include "hits_dev.hpp"

namespace calitas {

namespace {

__global__ void hit_kernel(const RawAln* fin, uint32_t n, const GuideDev* guides, const uint64_t* win_base, const int2* win,
                           int score_hi, HitRec* hits, uint64_t* keys, uint32_t* vals, uint32_t* wks, uint32_t* flags) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hit_body(i, fin, guides, win_base, win, score_hi, hits, keys, vals, wks, flags);
}

// The caller's own hits (HitsExt) behind the device's n_dev: record, sort key, value.  HitRec::minus carries the strand in bit 0 and
// "placed only" in bit 1; start / end / gend are not used for them (their rows come finished).
__global__ void ext_kernel(const HitsExtKey* ext, uint32_t n_ext, uint32_t n_dev, int32_t contig, int score_hi, HitRec* hits, uint64_t* keys,
                           uint32_t* vals, uint32_t* flags) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ext) return;
  const HitsExtKey x = ext[e];
  HitRec h;
  h.contig = contig; h.start = x.coordinate_start; h.end = x.end + 1; h.gstart = x.coordinate_start; h.gend = x.end + 1; h.score = x.score;
  h.rh_end = x.end; h.minus = x.flags & 3u;
  int sb = score_hi - h.score;
  if (sb < 0 || sb >= (1 << SCORE_BITS) || h.gstart < 0) { atomicOr(flags, HITS_FLAG_SCORE_RANGE); sb = 0; }
  const uint32_t i = n_dev + e;
  hits[i] = h;
  keys[i] = ((uint64_t)(uint32_t)h.contig << 46) | ((uint64_t)(uint32_t)h.gstart << 15) | ((uint64_t)(h.minus & 1u) << 14) | (uint64_t)sb;
  vals[i] = i;
}

// ReferenceHit.sort without a sort.  The accepted alignments arrive in (contig, window, ...) order, and hits of two windows that
// share no base cannot be out of order relative to each other (a hit starts inside its window): with reach = the number of
// window steps after which two windows are disjoint, hit i only has to be compared with the hits of the windows less than
// `reach` steps away.  rank(i) = i - (earlier neighbours with a greater key) + (later neighbours with a smaller key); equal keys
// keep their arrival order, as a stable sort would.  One launch instead of the seven of a 64-bit radix / merge sort; the caller
// takes the general sort when a window holds more records than a lane should walk past.
__device__ __forceinline__ void rank_body(const uint32_t i, const uint64_t* keys, const uint32_t* wks, uint32_t n, uint32_t reach, uint32_t* order) {
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

__global__ void rank_kernel(const uint64_t* keys, const uint32_t* wks, uint32_t n, uint32_t reach, uint32_t* order) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank_body(i, keys, wks, n, reach, order);
}


__device__ __forceinline__ void prep_body(const uint32_t i, const HitRec* hits, const uint32_t* order, int max_overlap, int32_t* s_start, int32_t* s_end,
                                          int32_t* s_score, uint32_t* s_cs, uint8_t* head, uint8_t* keep) {
  const HitRec h = hits[order[i]];
  const uint32_t cs = ((uint32_t)h.contig << 2) | h.minus;   // (contig, strand) group; bit 1: a hit of another group, placed only (HitsExt)
  s_start[i] = h.gstart; s_end[i] = h.rh_end; s_score[i] = h.score; s_cs[i] = cs;
  if (h.minus & 2u) { keep[i] = 1; head[i] = 0; return; }   // kept by its own group's walk; no walk starts at it or passes through it
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

__global__ void prep_kernel(const HitRec* hits, const uint32_t* order, uint32_t n, int max_overlap, int32_t* s_start, int32_t* s_end,
                            int32_t* s_score, uint32_t* s_cs, uint8_t* head, uint8_t* keep) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) prep_body(i, hits, order, max_overlap, s_start, s_end, s_score, s_cs, head, keep);
}

__device__ __forceinline__ void cluster_body(const uint32_t i, const int32_t* s_start, const int32_t* s_end, const int32_t* s_score, const uint32_t* s_cs,
                                             const uint8_t* head, uint32_t n, int max_overlap, uint8_t* keep, uint32_t* flags, const HitsOwn own = HitsOwn()) {
  if (!head[i]) return;
  const uint32_t cs = s_cs[i];
  // HitsOwn: a walk that starts where hits may be missing must not decide an owned hit
  const bool unsure = (((unsigned long long)(cs >> 2) << 32) | (uint32_t)s_start[i]) < own.safe;
  auto owned = [&](uint32_t j) { const unsigned long long k = ((unsigned long long)(cs >> 2) << 32) | (uint32_t)s_start[j]; return k >= own.lo && k < own.hi; };
  // next hit of this (contig, strand) group after position j, or n
  auto next = [&](uint32_t j) {
    for (j++; j < n; j++) {
      const uint32_t c = s_cs[j];
      if (c == cs) return j;
      if ((c >> 2) != (cs >> 2)) break;
    }
    return n;
  };
  uint32_t j = i, steps = 0;
  for (;;) {                                            // SR:661-671
    const uint32_t hit = j;
    if (unsure && owned(hit)) { atomicOr(flags, HITS_FLAG_HALO); return; }
    j = next(j);
    const int hs = s_start[hit], he = s_end[hit], hsc = s_score[hit];
    bool more = false;
    int ov = 0;
    for (;;) {
      more = j < n && !head[j];
      if (!more) break;
      ov = max(0, min(s_end[j], he) - max(s_start[j], hs));   // RH:141-144
      if (!(ov >= max_overlap && s_score[j] <= hsc)) break;
      if (unsure && owned(j)) { atomicOr(flags, HITS_FLAG_HALO); return; }
      j = next(j);
      if (++steps > CLUSTER_MAX) break;
    }
    if (steps > CLUSTER_MAX) { atomicOr(flags, HITS_FLAG_CLUSTER); return; }
    if (!more || ov < max_overlap) keep[hit] = 1;
    if (!more) return;
    if (++steps > CLUSTER_MAX) { atomicOr(flags, HITS_FLAG_CLUSTER); return; }
  }
}

__global__ void cluster_kernel(const int32_t* s_start, const int32_t* s_end, const int32_t* s_score, const uint32_t* s_cs,
                               const uint8_t* head, uint32_t n, int max_overlap, uint8_t* keep, uint32_t* flags, const HitsOwn own) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cluster_body(i, s_start, s_end, s_score, s_cs, head, n, max_overlap, keep, flags, own);
}

// At most this many accepted alignments: coordinates, order, restart points and the removeOverlaps walk in ONE launch of one
// workgroup (the stages talk through global memory; between two of them a device-scope fence and a barrier), and the row offsets in
// another (offs_small_kernel).  An E. coli-sized call has 20-40 alignments and ~20 dependent launches of 3-5 us, with 4-10 us between
// any two: the launches are its run time.
constexpr uint32_t HITS_SMALL = 1024;

struct HitsSmallArgs {
  const RawAln* fin; const GuideDev* guides; const uint64_t* win_base; const int2* win;
  HitRec* hits; uint64_t* keys; uint32_t *vals, *wks, *order;
  int32_t *s_start, *s_end, *s_score; uint32_t* s_cs; uint8_t *head, *keep;
  uint32_t* flags;
  uint32_t n, reach;
  int score_hi, max_overlap;
};

__device__ __forceinline__ void stage_sync() {   // what one stage wrote, the next one reads -- possibly in another wave of the workgroup
  __threadfence();
  __syncthreads();
}

__global__ __launch_bounds__(HITS_SMALL) void hits_small_kernel(HitsSmallArgs a) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = threadIdx.x;
  const bool mine = i < a.n;
  if (mine) hit_body(i, a.fin, a.guides, a.win_base, a.win, a.score_hi, a.hits, a.keys, a.vals, a.wks, a.flags);
  stage_sync();
  if (mine) rank_body(i, a.keys, a.wks, a.n, a.reach, a.order);
  stage_sync();
  if (mine) prep_body(i, a.hits, a.order, a.max_overlap, a.s_start, a.s_end, a.s_score, a.s_cs, a.head, a.keep);
  stage_sync();
  if (mine) cluster_body(i, a.s_start, a.s_end, a.s_score, a.s_cs, a.head, a.n, a.max_overlap, a.keep, a.flags);
}

// HitsExt::rows_for: which of the caller's own hits the walks kept, by entry (order[k] - n_dev), for the host to build just their rows.
__global__ void ext_keep_kernel(const uint32_t* order, const uint8_t* keep, uint32_t n, uint32_t n_dev, uint8_t* out) {
  CALITAS_TAIL_PRIO();
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t v = order[k];
  if (v >= n_dev) out[v - n_dev] = keep[k];
}

// HitsExtRows::fill_on_host: where the row of every kept entry of the caller's belongs in the text (the rows kernel leaves the hole).
__global__ void ext_place_kernel(const uint32_t* order, const uint32_t* midlen, const uint64_t* offs, uint32_t n, uint32_t n_dev, uint64_t* place) {
  CALITAS_TAIL_PRIO();
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t v = order[k];
  if (v >= n_dev) place[v - n_dev] = midlen[k] == 0xFFFFFFFFu ? offs[k] : ~0ull;
}

// Rows in two kernels (round 4; round 2 built every middle part into a staging buffer -- mid_kernel -- and assembled the rows from it
// once their offsets were known -- out_kernel: 350 bytes per row written and read again, 20 GB of staging for the 4.1e7 rows of a PAM-less
// whole-genome search):
//   len_kernel   one lane per sorted position: the row's length from the alignment's op counts (middle_length: the arithmetic
//                build_middle does with ballots, without fetching or writing anything) -- what the offsets' scan needs;
//   rows_kernel  after the scan and the host's look at the totals: a wave per row builds the middle part in its LDS line and writes
//                head | chromosome \t | middle | tail at the row's final place.  The two lengths are held against each other for every
//                row (a mismatch raises HITS_FLAG_INTERNAL in the work's late word: two implementations of one arithmetic).
__global__ __launch_bounds__(256) void len_kernel(MidArgs a, uint32_t* midlen, uint64_t* lens, uint32_t* flags) {
  CALITAS_TAIL_PRIO();
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  bool live = false, is_ext = false;
  if (k < a.n) {
    live = a.keep[k] != 0;
    const uint32_t v = a.order[k];
    is_ext = v >= a.n_dev;
    if (is_ext) {                                          // the caller's own hit: its row comes finished (HitsExt), rows_kernel copies it
      midlen[k] = live ? 0xFFFFFFFFu : 0u;
      lens[k] = live ? a.ext_off[v - a.n_dev + 1] - a.ext_off[v - a.n_dev] : 0;
    } else {
      const HitRec h = a.hits[v];
      if (live && (a.own_lo != 0 || a.own_hi != ~0ull)) {   // HitsOwn: rows of the stretch only
        const unsigned long long key = ((unsigned long long)(uint32_t)h.contig << 32) | (uint32_t)h.gstart;
        live = key >= a.own_lo && key < a.own_hi;
      }
      int len = 0;
      if (live) {
        const RawAln* rp = a.fin + v;
        const GuideDev* gp = a.guides + rp->guide;
        const int pam = rp->pam;
        len = middle_length(rp, h, gp->L, pam >= 0 ? gp->pam_len[pam] : 0, (int)a.rc.pu_len[pam + 1], (int)a.n_max, (int)a.mid_bound);
        if (len < 0) { atomicOr(flags, HITS_FLAG_ROW); len = 0; }
      }
      midlen[k] = (uint32_t)len;
      lens[k] = live ? (uint64_t)(a.rc.head_len + (a.name_off[h.contig + 1] - a.name_off[h.contig]) + 1 + (uint32_t)len + a.rc.tail_len) : 0;
    }
  }
  // rows (and kept hits of the caller's own) of the call: one atomic per wave
  const unsigned long long lv = __ballot(live), ev = __ballot(live && is_ext);
  if ((threadIdx.x & 63) == 0) {
    if (lv) atomicAdd(a.n_rows, (uint32_t)__popcll(lv));
    if (ev) atomicAdd(a.ext_kept, (uint32_t)__popcll(ev));
  }
}

struct OutArgs {
  const char* names;
  const uint64_t* offs;
  const uint32_t* midlen;      // per sorted position: 0 dropped, 0xFFFFFFFF a kept hit of the caller's own (copy its finished row)
  const char* ext_rows;
  uint32_t* late;              // the work's mailbox word for flags raised while rows are written (page-locked host memory)
  const uint64_t* counts;      // [0] bytes of the text, [2] low word: flags -- final when this kernel starts (total_kernel ran before it)
  uint64_t text_cap;           // room at `text`: a longer text makes the kernel return at once (the host grows the buffer and runs it again)
};

constexpr int ROWS_PER_WAVE = 4;

__global__ __launch_bounds__(256) void rows_kernel(MidArgs a, OutArgs o, char* text) {
  CALITAS_TAIL_PRIO();
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // per wave: line | fwd; then the constant strings (head, tail, queries, PAMs)
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* line = lds + wave * (MID_LINE + MID_FWD);
  uint8_t* fwd = line + MID_LINE;
  uint8_t* blob = lds + 4 * (MID_LINE + MID_FWD);
  for (uint32_t i = threadIdx.x; i < a.blob_bytes; i += 256) blob[i] = (uint8_t)a.blob[i];
  __syncthreads();
  // queued right behind the kernels that size the text, before the host has looked at their counts: nothing to do when the stage
  // declined, and nothing may be written when the text does not fit the buffer it was given
  if (o.counts[0] > o.text_cap || (uint32_t)o.counts[2] != 0u) return;
  const uint8_t* head = blob + a.rc.head_off;
  const uint8_t* tail = blob + a.rc.tail_off;
  const uint32_t k0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
  for (int rr = 0; rr < ROWS_PER_WAVE; rr++) {
    const uint32_t k = k0 + (uint32_t)rr;
    if (k >= a.n) break;
    const uint32_t want = uniform_ptr(o.midlen)[k];
    if (want == 0) continue;                              // dropped by removeOverlaps (or not owned)
    const uint32_t v = uniform_ptr(a.order)[k];
    char* dst = text + uniform_ptr(o.offs)[k];
    if (want == 0xFFFFFFFFu) {
      const uint32_t e = v - a.n_dev;
      if (!o.ext_rows) continue;                          // (fill_on_host: a hole, the caller has the row)
      const uint64_t b0 = uniform_ptr(a.ext_off)[e], nb = uniform_ptr(a.ext_off)[e + 1] - b0;
      for (uint64_t b = (uint64_t)lane; b < nb; b += 64) dst[b] = o.ext_rows[b0 + b];
      continue;
    }
    const auto* rp = uniform_ptr(a.fin) + v;
    RowIn r;
    const auto* ow = (const __attribute__((address_space(4))) uint32_t*)rp->ops;   // RawAln::ops sits at a 4-byte aligned offset
    r.w0 = ow[0]; r.w1 = ow[1]; r.w2 = ow[2]; r.w3 = ow[3]; r.w4 = ow[4];
    r.n_ops = rp->n_ops; r.pam = rp->pam; r.offset = rp->offset; r.pam_x = rp->pam_x;
    const auto* hp = uniform_ptr(a.hits) + v;
    HitRec h;
    h.contig = hp->contig; h.start = hp->start; h.end = hp->end; h.gstart = hp->gstart; h.gend = hp->gend; h.score = hp->score;
    h.rh_end = hp->rh_end; h.minus = hp->minus;
    const auto* gp = uniform_ptr(a.guides) + rp->guide;
    RowGuide g;
    g.L = gp->L; g.pam5 = gp->pam5; g.pam_len = r.pam >= 0 ? gp->pam_len[r.pam] : 0;
    wave_lds_sync();                                      // the copy-out of the previous row is done with line[]
    const int len = build_middle(line, fwd, a, blob, r, h, g, lane);
    if (len < 0 || (uint32_t)len != want) {               // cannot happen: both kernels run the same arithmetic
      if (lane == 0) __hip_atomic_fetch_or(o.late, HITS_FLAG_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      continue;
    }
    wave_lds_sync();
    const uint32_t nb = uniform_ptr(a.name_off)[h.contig], nl = uniform_ptr(a.name_off)[h.contig + 1] - nb;
    const uint32_t s0 = a.rc.head_len, s1 = s0 + nl + 1, s2 = s1 + (uint32_t)len, total = s2 + a.rc.tail_len;
    for (uint32_t x = (uint32_t)lane; x < total; x += 64) {
      uint8_t ch;
      if (x < s0) ch = head[x];
      else if (x < s1) ch = (x - s0 < nl) ? (uint8_t)o.names[nb + x - s0] : (uint8_t)'\t';
      else if (x < s2) ch = line[x - s1];
      else ch = tail[x - s2];
      dst[x] = (char)ch;
    }
  }
}

// Length of the text = end of the last row; posted with the other two counts (rows, flags) to the host's mailbox (mailbox.hpp) by the
// same thread -- a kernel of its own for the post was one more launch on the path every call waits for.
__global__ void total_kernel(const uint64_t* offs, const uint64_t* lens, uint32_t n, uint64_t* counts, uint32_t* box, uint32_t seq) {
  CALITAS_TAIL_PRIO();
  counts[0] = offs[n - 1] + lens[n - 1];
  __threadfence();
  const uint32_t* src = reinterpret_cast<const uint32_t*>(counts);
  for (int i = 0; i < 6; i++) box[1 + i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // rows and flags come from other kernels' atomics
  __threadfence_system();
  __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Row offsets, the length of the text and the post to the host for at most HITS_SMALL rows: one workgroup instead of the scan's two
// launches and total_kernel.
__global__ __launch_bounds__(HITS_SMALL) void offs_small_kernel(const uint64_t* lens, uint32_t n, uint64_t* offs, uint64_t* counts, uint32_t* box,
                                                                uint32_t seq) {
  CALITAS_TAIL_PRIO();
  __shared__ uint64_t s_sum[HITS_SMALL];
  const uint32_t i = threadIdx.x;
  const uint64_t mine = i < n ? lens[i] : 0;
  s_sum[i] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < HITS_SMALL; d <<= 1) {       // inclusive scan, Hillis-Steele
    const uint64_t add = i >= d ? s_sum[i - d] : 0;
    __syncthreads();
    s_sum[i] += add;
    __syncthreads();
  }
  if (i < n) offs[i] = s_sum[i] - mine;
  if (i == 0) {
    counts[0] = s_sum[HITS_SMALL - 1];
    __threadfence();
    const uint32_t* src = reinterpret_cast<const uint32_t*>(counts);
    for (int k = 0; k < 6; k++) box[1 + k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

uint32_t hits_late(const HitsWork* w) { return w && w->mbox.host ? w->mbox.host[HITS_BOX_LATE] : 0u; }

void hits_destroy(HitsWork* w) {
  if (!w) return;
  (void)hipFree(w->hits); (void)hipFree(w->keys); (void)hipFree(w->keys2); (void)hipFree(w->lens); (void)hipFree(w->offs);
  (void)hipFree(w->vals); (void)hipFree(w->vals2); (void)hipFree(w->s_cs); (void)hipFree(w->wks); (void)hipFree(w->s_start); (void)hipFree(w->s_end);
  (void)hipFree(w->s_score); (void)hipFree(w->keep); (void)hipFree(w->head); (void)hipFree(w->temp); (void)hipFree(w->text);
  (void)hipFree(w->midlen); (void)hipFree(w->blob); (void)hipFree(w->names); (void)hipFree(w->name_off);
  (void)hipFree(w->d_counts); (void)hipFree(w->ext_keys); (void)hipFree(w->ext_off); (void)hipFree(w->ext_rows); (void)hipFree(w->ext_keep); (void)hipFree(w->ext_place);
  if (w->h_ext_keep) (void)hipHostFree(w->h_ext_keep);
  if (w->h_ext_place) (void)hipHostFree(w->h_ext_place);
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
  HitsSetup hs{};
  hipError_t e;
  TRY(hits_prepare_host(pw, st, &hs));
  TRY(hipMemsetAsync(hs.d_counts, 0, 3 * sizeof(uint64_t), stream));
  return hipMemcpyAsync(hs.d_blob, hs.blob, hs.blob_bytes, hipMemcpyHostToDevice, stream);
}

hipError_t hits_prepare_host(HitsWork** pw, const RowStrings& st, HitsSetup* out) {
  if (!*pw) *pw = new HitsWork();
  HitsWork& w = **pw;
  hipError_t e;
  if (!w.d_counts) TRY(hipMalloc((void**)&w.d_counts, 3 * sizeof(uint64_t)));
  if (!w.h_counts) TRY(hipHostMalloc((void**)&w.h_counts, 3 * sizeof(uint64_t), hipHostMallocDefault));
  RowConstDev rc{};
  std::string& blob = w.blob_host;
  blob.clear();
  auto add = [&](const std::string& s, uint32_t& off, uint32_t& len) { off = (uint32_t)blob.size(); len = (uint32_t)s.size(); blob += s; };
  add(st.head, rc.head_off, rc.head_len); add(st.tail, rc.tail_off, rc.tail_len); add(st.proto_len, rc.plen_off, rc.plen_len);
  for (size_t i = 0; i < st.query.size() && i <= (size_t)MAX_PAMS; i++) {
    add(st.query[i], rc.q_off[i], rc.q_len[i]);
    add(st.pam_used[i], rc.pu_off[i], rc.pu_len[i]);
  }
  TRY(grow(&w.blob, w.blob_cap, blob.size() + 16));          // (the lane's setup kernel writes it in 16-byte pieces)
  w.rc = rc; w.blob_bytes = blob.size(); w.prepared = true;
  out->blob = blob.data(); out->blob_bytes = (uint32_t)blob.size(); out->d_blob = w.blob; out->d_counts = w.d_counts;
  return hipSuccess;
}

hipError_t hits_run(HitsWork** pw, const HitsRef& ref, const RawAln* d_final, uint32_t n_in, const GuideDev* d_guides,
                    const uint64_t* d_win_base, const int2* d_win, const RowStrings& st, int max_overlap, int score_hi,
                    int max_ops, uint32_t window_reach, hipStream_t stream, HitsResult* res, const HitsExt* ext, const HitsOwn* own) {
  if (!*pw) *pw = new HitsWork();
  HitsWork& w = **pw;
  hipError_t e;
  *res = HitsResult{};
  const uint32_t n_ext = ext ? ext->n : 0;
  if (ext && ext->kept) *ext->kept = 0;
  if ((uint64_t)n_in + n_ext > 0xFFFFFFF0ull) { res->flags = HITS_FLAG_CLUSTER; return hipSuccess; }
  const size_t n = (size_t)n_in + n_ext;
  if (n == 0) return hipSuccess;
  if (n_ext) window_reach = 0;                         // the caller's hits have no window: the general stable sort
  n_in = (uint32_t)n;                                  // every stage behind hit_kernel runs over both; n_dev is the device's share
  const uint32_t n_dev = (uint32_t)(n - n_ext);
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
  bool ext_on_host = false;                            // HitsExtRows::fill_on_host
  const bool on_demand = n_ext && ext->rows_for;       // the rows of the caller's hits once the walks have decided (hits.hpp)
  if (on_demand && own) return hipErrorInvalidValue;
  // the rows of the caller's hits (offsets, text) to the device: before anything runs when they came with the call, behind the walks on demand
  auto upload_ext_rows = [&](const HitsExtRows& r) -> hipError_t {
    if (!r.row_off || r.row_off[0] != 0) return hipErrorInvalidValue;
    const size_t row_bytes = (size_t)r.row_off[n_ext];
    TRY(grow(&w.ext_off, w.ext_off_cap, (size_t)n_ext + 1));
    TRY(grow(&w.ext_rows, w.ext_rows_cap, std::max<size_t>(1, row_bytes)));
    TRY(hipMemcpyAsync(w.ext_off, r.row_off, ((size_t)n_ext + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    if (r.fill_on_host) { ext_on_host = true; return hipSuccess; }
    if (row_bytes && r.rows) TRY(hipMemcpyAsync(w.ext_rows, r.rows, row_bytes, hipMemcpyHostToDevice, stream));
    else if (row_bytes) {
      if (!r.n_seg || !r.seg || !r.seg_off || r.seg_off[0] != 0 || r.seg_off[r.n_seg] != row_bytes) return hipErrorInvalidValue;
      for (uint32_t sg = 0; sg < r.n_seg; sg++) {
        const uint64_t nb = r.seg_off[sg + 1] - r.seg_off[sg];
        if (nb) TRY(hipMemcpyAsync(w.ext_rows + r.seg_off[sg], r.seg[sg], (size_t)nb, hipMemcpyHostToDevice, stream));
      }
    }
    return hipSuccess;
  };
  if (n_ext) {
    TRY(grow(&w.ext_keys, w.ext_keys_cap, (size_t)n_ext));
    TRY(hipMemcpyAsync(w.ext_keys, ext->keys, (size_t)n_ext * sizeof(HitsExtKey), hipMemcpyHostToDevice, stream));
    if (!on_demand) {
      HitsExtRows given;
      given.row_off = ext->row_off; given.rows = ext->rows; given.n_seg = ext->n_seg; given.seg = ext->seg; given.seg_off = ext->seg_off;
      TRY(upload_ext_rows(given));
    }
  }
  const bool small = n_in <= HITS_SMALL && window_reach != 0 && !own;
  const HitsOwn ho = own ? *own : HitsOwn();
  if (small) {          // 1-3 in one launch
    HitsSmallArgs sa{};
    sa.fin = d_final; sa.guides = d_guides; sa.win_base = d_win_base; sa.win = d_win; sa.hits = w.hits; sa.keys = w.keys; sa.vals = w.vals;
    sa.wks = w.wks; sa.order = w.vals2; sa.s_start = w.s_start; sa.s_end = w.s_end; sa.s_score = w.s_score; sa.s_cs = w.s_cs; sa.head = w.head;
    sa.keep = w.keep; sa.flags = d_flags; sa.n = n_in; sa.reach = window_reach; sa.score_hi = score_hi; sa.max_overlap = max_overlap;
    hipLaunchKernelGGL(hits_small_kernel, dim3(1), dim3(HITS_SMALL), 0, stream, sa);
  } else {
  // 1: coordinates and the final order
  if (n_dev) hipLaunchKernelGGL(hit_kernel, dim3((n_dev + 255) / 256), block, 0, stream, d_final, n_dev, d_guides, d_win_base, d_win, score_hi, w.hits, w.keys, w.vals, w.wks, d_flags);
  if (n_ext) hipLaunchKernelGGL(ext_kernel, dim3((n_ext + 255) / 256), block, 0, stream, (const HitsExtKey*)w.ext_keys, n_ext, n_dev, ext->contig, score_hi, w.hits, w.keys, w.vals, d_flags);
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
                     (const uint32_t*)w.s_cs, (const uint8_t*)w.head, n_in, max_overlap, w.keep, d_flags, ho);
  }
  if (on_demand) {
    // the walks' verdicts on the caller's hits to the host, the rows of the kept ones back: the one place where this stage waits for the
    // host in the middle (a contig's worth of a variant search: 80 000 entries, 8 000 rows, a millisecond of the caller's workers)
    TRY(grow(&w.ext_keep, w.ext_keep_cap, (size_t)n_ext));
    if ((size_t)n_ext > w.h_ext_keep_cap) {
      if (w.h_ext_keep) (void)hipHostFree(w.h_ext_keep);
      w.h_ext_keep = nullptr; w.h_ext_keep_cap = 0;
      const size_t cap = (size_t)n_ext + (size_t)n_ext / 4 + 4096;
      TRY(hipHostMalloc((void**)&w.h_ext_keep, cap, hipHostMallocDefault));
      w.h_ext_keep_cap = cap;
    }
    hipLaunchKernelGGL(ext_keep_kernel, grid, block, 0, stream, (const uint32_t*)w.vals2, (const uint8_t*)w.keep, n_in, n_dev, w.ext_keep);
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(w.h_ext_keep, w.ext_keep, (size_t)n_ext, hipMemcpyDeviceToHost, stream));
    TRY(hipStreamSynchronize(stream));
    HitsExtRows made;
    if (ext->rows_for(w.h_ext_keep, &made) != 0) return hipErrorUnknown;
    TRY(upload_ext_rows(made));
  }
  // 4: rows
  // (a row with more padded columns than a wave has lanes raises HITS_FLAG_ROW and the caller finishes on the host)
  const uint32_t n_max = (uint32_t)std::min<int>(MID_COLS, std::max(1, max_ops));
  const uint32_t mid_bound = (6 * n_max + 128 + 3) & ~3u;
  const uint32_t mid_lds = 4 * (MID_LINE + MID_FWD) + (uint32_t)((blob_bytes + 15) & ~(size_t)15);   // four waves' line buffers | constant strings
  if (mid_lds > 64 * 1024) {   // beyond the default dynamic LDS limit (absurdly long parameter strings): decline
    TRY(hipStreamSynchronize(stream));
    res->flags = HITS_FLAG_ROW;
    return hipSuccess;
  }
  const size_t n_pad = (n + 63) / 64 * 64;
  TRY(grow(&w.midlen, w.midlen_cap, n_pad));
  if (!w.text) TRY(grow(&w.text, w.text_cap, std::min<size_t>((size_t)32 << 20, n * (size_t)(mid_bound + rc.head_len + rc.tail_len + 64))));   // (first call: a guess)
  MidArgs ma{};
  ma.ref = ref; ma.rc = rc; ma.blob = w.blob; ma.name_off = w.name_off; ma.fin = d_final; ma.hits = w.hits; ma.guides = d_guides;
  ma.keep = w.keep; ma.order = w.vals2; ma.n = n_in; ma.n_rows = d_kept; ma.n_dev = n_dev; ma.ext_off = w.ext_off; ma.ext_kept = d_kept + 1; ma.own_lo = ho.lo; ma.own_hi = ho.hi; ma.mid_bound = mid_bound; ma.n_max = n_max; ma.blob_bytes = (uint32_t)blob_bytes;
  hipLaunchKernelGGL(len_kernel, grid, block, 0, stream, ma, w.midlen, w.lens, d_flags);
  TRY(mailbox_open(w.mbox));
  w.mbox.seq++;
  w.mbox.host[HITS_BOX_LATE] = 0;                          // raised by rows_kernel while rows are written; read when the stream is done (hits_late)
  if (small) {
    hipLaunchKernelGGL(offs_small_kernel, dim3(1), dim3(HITS_SMALL), 0, stream, (const uint64_t*)w.lens, n_in, w.offs, w.d_counts, w.mbox.dev, w.mbox.seq);
  } else {
    ts = w.temp_cap;
    TRY(rocprim::exclusive_scan(w.temp, ts, w.lens, w.offs, (uint64_t)0, n, rocprim::plus<uint64_t>(), stream));
    hipLaunchKernelGGL(total_kernel, dim3(1), dim3(1), 0, stream, (const uint64_t*)w.offs, (const uint64_t*)w.lens, n_in, w.d_counts, w.mbox.dev, w.mbox.seq);
  }
  TRY(hipGetLastError());
  // The rows kernel goes out at once, into the buffer as it is (sized by the last call's text: a search is usually followed by one like
  // it): the host's look at the counts -- a round trip of 20-30 us -- is then off the lane's critical path, as in the binned tail.  A
  // text that does not fit makes the kernel return untouched; the buffer grows and the kernel runs again.
  OutArgs oa{};
  oa.names = w.names; oa.offs = w.offs; oa.midlen = w.midlen; oa.ext_rows = ext_on_host ? nullptr : w.ext_rows; oa.late = w.mbox.dev + HITS_BOX_LATE;
  if (ext_on_host) {
    TRY(grow(&w.ext_place, w.ext_place_cap, (size_t)n_ext));
    if ((size_t)n_ext > w.h_ext_place_cap) {
      if (w.h_ext_place) (void)hipHostFree(w.h_ext_place);
      w.h_ext_place = nullptr; w.h_ext_place_cap = 0;
      const size_t cap = (size_t)n_ext + (size_t)n_ext / 4 + 4096;
      TRY(hipHostMalloc((void**)&w.h_ext_place, cap * sizeof(uint64_t), hipHostMallocDefault));
      w.h_ext_place_cap = cap;
    }
    hipLaunchKernelGGL(ext_place_kernel, grid, block, 0, stream, (const uint32_t*)w.vals2, (const uint32_t*)w.midlen, (const uint64_t*)w.offs, n_in, n_dev, w.ext_place);
    TRY(hipGetLastError());
    TRY(hipMemcpyAsync(w.h_ext_place, w.ext_place, (size_t)n_ext * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
  }
  oa.counts = w.d_counts;
  const unsigned rows_per_block = 4 * ROWS_PER_WAVE;
  const dim3 rows_grid((n_in + rows_per_block - 1) / rows_per_block);
  oa.text_cap = w.text_cap;
  hipLaunchKernelGGL(rows_kernel, rows_grid, dim3(256), mid_lds, stream, ma, oa, w.text);
  TRY(hipGetLastError());
  TRY(mailbox_wait(w.mbox, stream));
  for (int k = 0; k < 3; k++) w.h_counts[k] = (uint64_t)w.mbox.host[1 + 2 * k] | ((uint64_t)w.mbox.host[2 + 2 * k] << 32);
  TRY(hipGetLastError());
  res->flags = (uint32_t)w.h_counts[2];
  if (res->flags) return hipSuccess;
  res->n_rows = (uint32_t)w.h_counts[1];
  if (ext && ext->kept) *ext->kept = (uint32_t)(w.h_counts[1] >> 32);
  res->text_bytes = w.h_counts[0];
  if (res->text_bytes > oa.text_cap) {                       // the kernel returned at once: a buffer of the right size (and some more), again
    TRY(hipStreamSynchronize(stream));
    TRY(grow(&w.text, w.text_cap, (size_t)res->text_bytes + (size_t)(res->text_bytes / 8) + 4096));
    oa.text_cap = w.text_cap;
    hipLaunchKernelGGL(rows_kernel, rows_grid, dim3(256), mid_lds, stream, ma, oa, w.text);
    TRY(hipGetLastError());
  }
  res->d_text = w.text;
  res->ext_place = ext_on_host ? w.h_ext_place : nullptr;
  return hipSuccess;
}

#undef TRY

}  // namespace calitas
