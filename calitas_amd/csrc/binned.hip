// binned.hip -- everything behind trace_kernel for one guide, one wave per reference bin (see binned.hpp for the why).
//
// A bin is a power-of-two stretch of ONE contig (32 kb for the default window).  trace_kernel drops every raw alignment into the bin
// its window STARTS in (kernels.hip); bin_hits_kernel's wave for bin b owns the hits whose coordinate_start -- the first key of
// ReferenceHit.sort (ReferenceHit.scala:284) -- lies in [lo, hi) = the bin's stretch, so the rows of consecutive bins are consecutive
// pieces of hits.txt.  What the wave needs to decide those hits exactly:
//   * every window that can hold such a hit: windows starting in (lo - W, hi) -- its own bin and the tail of bin b-1;
//   * removeOverlaps (SearchReference.scala:653-675) walks a (chromosome, strand) group left to right and restarts wherever a hit
//     overlaps nothing before it (hits.hip, prep_kernel).  A hit's fate hangs on its cluster: from the restart point at or left of it
//     to the first hit right of it that it does not swallow.  Hits are at most HIT_MAX_LEN (128) bases long, so the right side needs
//     the hits starting within 128 bases behind hi (windows starting below hi + 128: the head of bin b+1), and the left side needs a
//     restart point whose own look-back (128 bases) is inside the stretch where all hits are known.  The wave takes the windows
//     starting in [lo - 2W, hi + 128): all hits with coordinate_start >= lo - W are known, restart points from lo - W + 128 on are
//     certain, and a hit of the bin that no walk from a certain restart point reaches raises BIN_FLAG_HALO (a tandem repeat with
//     chained hits over more than W - 128 bases): the call then finishes on the general kernels.
// Per wave: (1) the context's raw alignments (<= 128, two per lane's worth of LDS), (2) the greedy of SGA:315-320 window by window --
// a wave-wide maximum of the order keys per round, as filter_wave_kernel does --, (3) GuideAlignment coordinates of the accepted ones
// (<= 64, one per lane), (4) their order by counting, restart points, the cluster walks, (5) the row lengths of the kept hits of the
// bin (the row builder's own arithmetic without the text: build_middle<false>) and the bin's (rows, bytes) with a two-level sum.
// bin_rows_kernel then places every bin from those sums and builds its rows at their final offsets -- no staging copy of the text.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "binned.hpp"
#include "hits_dev.hpp"
#include "select_dev.hpp"

namespace calitas {

namespace {

constexpr int BIN_WAVES = 4;            // waves (= bins) per workgroup
constexpr uint32_t ACC_MAX = 64;        // accepted alignments of a bin's context: one per lane
constexpr uint32_t CHUNK_SHIFT = 8;     // bins per chunk of the two-level sum that places the bins' text
static_assert(BIN_CAP == 64, "a lane holds one alignment of each of the three bins of a context");

struct BinRow {            // a kept hit of a bin, in final order
  uint32_t raw;            // its index in the lane's list of raw alignments
  uint32_t len;            // length of the row's middle part (what build_middle returns)
};

struct BinArgs {
  const RawAln* raw;                 // the lane's raw alignments (trace_kernel's list)
  const uint32_t* bin_idx;           // n_bins x BIN_CAP indices into raw[]
  const uint32_t* bin_count;
  const uint32_t* bin_base;          // per contig (n_contigs + 1), absolute bin indices
  int n_contigs;
  uint32_t bin_first, n_bins, bin_shift;
  const GuideDev* guides;
  const uint64_t* win_base;
  const int2* win;
  int W, step, max_total_diffs, max_overlap;
  BinRow* rows;                      // n_bins x BIN_ROWS
  uint32_t* bin_rows;                // per bin: kept rows
  uint32_t* bin_bytes;               // per bin: their text bytes
  unsigned long long* chunk_bytes;   // per 256 bins (zero at launch)
  uint32_t* chunk_rows;
  uint32_t* chunk_acc;               // accepted alignments (after the per-window filter) of the windows that start in the chunk's bins
  uint32_t* flags;
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, off);
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// contig of an absolute bin index: last c with bin_base[c] <= bin
__device__ __forceinline__ int bin_contig(const uint32_t* bin_base, int n_contigs, uint32_t bin) {
  int lo = 0, hi = n_contigs;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bin_base[mid] <= bin) lo = mid; else hi = mid; }
  return lo;
}

__global__ __launch_bounds__(64 * BIN_WAVES) void bin_hits_kernel(BinArgs a, MidArgs m) {
  CALITAS_TAIL_PRIO();
  extern __shared__ __attribute__((aligned(16))) uint8_t s_blob[];     // the constant strings (queries: the case of a column's query base)
  __shared__ int32_t k_s[BIN_WAVES][ACC_MAX], k_e[BIN_WAVES][ACC_MAX];  // kept intervals of the window being filtered
  __shared__ uint32_t acc[BIN_WAVES][ACC_MAX];                          // accepted alignments in arrival order: index into raw[]
  // accepted alignments by arrival (t_*) and in ReferenceHit.sort order (s_*)
  __shared__ int32_t t_start[BIN_WAVES][ACC_MAX], t_score[BIN_WAVES][ACC_MAX];
  __shared__ int32_t s_start[BIN_WAVES][ACC_MAX], s_end[BIN_WAVES][ACC_MAX], s_score[BIN_WAVES][ACC_MAX];
  __shared__ uint8_t t_minus[BIN_WAVES][ACC_MAX], s_cs[BIN_WAVES][ACC_MAX], s_idx[BIN_WAVES][ACC_MAX], s_head[BIN_WAVES][ACC_MAX],
      s_keep[BIN_WAVES][ACC_MAX], s_done[BIN_WAVES][ACC_MAX];
  for (uint32_t i = threadIdx.x; i < m.blob_bytes; i += 64 * BIN_WAVES) s_blob[i] = (uint8_t)m.blob[i];
  __syncthreads();
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t rel = blockIdx.x * BIN_WAVES + wv;                     // bin of this wave, relative to the range
  if (rel >= a.n_bins) return;
  const uint32_t b = a.bin_first + rel;
  auto finish = [&](uint32_t n_rows, uint32_t bytes, uint32_t n_acc) {
    if (lane == 0) {
      a.bin_rows[rel] = n_rows; a.bin_bytes[rel] = bytes;
      if (n_rows) { atomicAdd(a.chunk_rows + (rel >> CHUNK_SHIFT), n_rows); atomicAdd(a.chunk_bytes + (rel >> CHUNK_SHIFT), (unsigned long long)bytes); }
      if (n_acc) atomicAdd(a.chunk_acc + (rel >> CHUNK_SHIFT), n_acc);
    }
  };
  auto decline = [&](uint32_t why) { if (lane == 0) atomicOr(a.flags, why); finish(0, 0, 0); };
  const uint32_t n_own = a.bin_count[rel];
  const uint32_t n_prev_raw = rel > 0 ? a.bin_count[rel - 1] : 0u;
  if (n_prev_raw == 0 && n_own == 0) { finish(0, 0, 0); return; }       // no window that could hold a hit of this bin has alignments
  const int c = bin_contig(a.bin_base, a.n_contigs, b);
  const uint32_t bb = b - a.bin_base[c];                                // bin inside its contig
  const bool has_prev = bb > 0 && rel > 0, has_next = b + 1 < a.bin_base[c + 1] && rel + 1 < a.n_bins;
  const uint32_t n_prev = has_prev ? n_prev_raw : 0u, n_next = has_next ? a.bin_count[rel + 1] : 0u;
  if (n_prev == 0 && n_own == 0) { finish(0, 0, 0); return; }
  if (n_prev > BIN_CAP || n_own > BIN_CAP || n_next > BIN_CAP) { decline(BIN_FLAG_CROWDED); return; }
  const int64_t lo = (int64_t)bb << a.bin_shift, hi = lo + ((int64_t)1 << a.bin_shift);
  const int64_t ctx_lo = lo - 2 * (int64_t)a.W, ctx_hi = hi + HIT_MAX_LEN;
  // hits are all known from here on; restart points are certain HIT_MAX_LEN further right.  At the start of a contig nothing is missing.
  const int64_t known_from = ctx_lo <= 0 ? -((int64_t)1 << 40) : lo - (int64_t)a.W;

  // ---- 1. the context's raw alignments: lane l holds alignment l of the previous, the own and the next bin (those whose window
  //         starts in [lo - 2W, hi + 128)) in registers ----
  unsigned long long key[3];        // order_key(), 0 = none / taken
  int st[3], en[3];
  uint32_t ed[3], widx[3], src[3], wk[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const uint32_t nq = q == 0 ? n_prev : q == 1 ? n_own : n_next;
    key[q] = 0; st[q] = 0; en[q] = 0; ed[q] = 0; widx[q] = 0xFFFFFFFFu; src[q] = 0; wk[q] = 0;
    if ((uint32_t)lane < nq) {
      const uint32_t idx = a.bin_idx[(size_t)((int64_t)rel + q - 1) * BIN_CAP + (uint32_t)lane];
      const RawAln* rp = a.raw + idx;
      const uint32_t window_k = rp->window_k;
      const int64_t ws = (int64_t)window_k * (int64_t)a.step;            // where the window starts on the contig
      if (ws >= ctx_lo && ws < ctx_hi) {
        const Derived d = derive(rp, a.guides, a.win_base, a.win, 0u, 0u);   // widx = the window's index in the table
        key[q] = order_key(d); st[q] = d.start; en[q] = d.end; ed[q] = d.edits; widx[q] = d.widx; src[q] = idx; wk[q] = window_k;
      }
    }
  }

  // ---- 2. the greedy of SGA:315-320, window by window in ascending order: acc[] = the accepted alignments in arrival order ----
  uint32_t nA = 0, nA_own = 0;
  bool over = false;
  for (uint32_t cur = 0;;) {
    uint32_t mw = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < 3; q++) if (widx[q] != 0xFFFFFFFFu && widx[q] >= cur && widx[q] < mw) mw = widx[q];
    const uint32_t w = wave_min_u32(mw);
    if (w == 0xFFFFFFFFu) break;
    cur = w + 1u;
    uint32_t nk = 0;
    for (uint32_t list = 0; list < 2; list++) {
      const uint32_t first_kept = nk;                                   // overlaps are only tested against the same strand (SGA:317)
      for (;;) {
        unsigned long long mine = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) if (widx[q] == w && (uint32_t)(key[q] >> 63) == list && key[q] > mine) mine = key[q];
        const unsigned long long bk = wave_max_u64(mine);
        if (bk == 0) break;
        const int owner = __ffsll((long long)__ballot(mine == bk)) - 1;   // keys are unique inside a window
        int b_start = 0, b_end = 0;
        uint32_t b_edits = 0, b_src = 0, b_wk = 0;
#pragma unroll
        for (int q = 0; q < 3; q++)
          if (widx[q] == w && key[q] == bk) { b_start = st[q]; b_end = en[q]; b_edits = ed[q]; b_src = src[q]; b_wk = wk[q]; key[q] = 0; }   // the owner takes it
        b_start = __shfl(b_start, owner); b_end = __shfl(b_end, owner); b_edits = (uint32_t)__shfl((int)b_edits, owner);
        b_src = (uint32_t)__shfl((int)b_src, owner); b_wk = (uint32_t)__shfl((int)b_wk, owner);
        if ((int)b_edits > a.max_total_diffs) continue;
        bool clash = false;
        for (uint32_t k = first_kept + (uint32_t)lane; k < nk && k < ACC_MAX; k += 64) {
          const int o = min(b_end, k_e[wv][k]) - max(b_start, k_s[wv][k]);   // GA:119-122
          clash = clash || o > a.max_overlap;
        }
        if (__ballot(clash) == 0) {
          if (nA < ACC_MAX && nk < ACC_MAX) { if (lane == 0) { k_s[wv][nk] = b_start; k_e[wv][nk] = b_end; acc[wv][nA] = b_src; } }
          else over = true;
          nk++; nA++;
          const int64_t ws = (int64_t)b_wk * (int64_t)a.step;            // (statistics) counted by the bin the window starts in
          if (ws >= lo && ws < hi) nA_own++;
          wave_lds_sync();
        }
      }
    }
  }
  if (over) { decline(BIN_FLAG_CROWDED); return; }
  if (nA == 0) { finish(0, 0, 0); return; }
  wave_lds_sync();

  // ---- 3. GuideAlignment coordinates of the accepted alignments (lane i = arrival i) ----
  const bool isA = (uint32_t)lane < nA;
  HitRec h{};
  uint32_t hsrc = 0;
  if (isA) {
    hsrc = acc[wv][lane];
    h = hit_record(a.raw + hsrc, a.guides, a.win_base, a.win);
    t_start[wv][lane] = h.gstart; t_score[wv][lane] = h.score; t_minus[wv][lane] = (uint8_t)h.minus;
  }
  if (__ballot(isA && h.gstart < 0) != 0) { decline(BIN_FLAG_RANGE); return; }
  wave_lds_sync();
  // ---- 4. ReferenceHit.sort among them by counting (equal keys keep their arrival order: a stable sort), restart points, walks ----
  if (isA) {
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nA; j++) {
      const int gs = t_start[wv][j], sc = t_score[wv][j];
      const uint32_t mi = t_minus[wv][j];
      const bool less = gs < h.gstart || (gs == h.gstart && (mi < h.minus || (mi == h.minus && sc > h.score)));
      const bool same = gs == h.gstart && mi == h.minus && sc == h.score;
      rank += (less || (same && j < (uint32_t)lane)) ? 1u : 0u;
    }
    s_start[wv][rank] = h.gstart; s_end[wv][rank] = h.rh_end; s_score[wv][rank] = h.score; s_cs[wv][rank] = (uint8_t)h.minus;
    s_idx[wv][rank] = (uint8_t)lane;
  }
  wave_lds_sync();
  const uint32_t r = (uint32_t)lane;                                    // from here on a lane is a sorted position
  bool head = false;
  if (r < nA) {
    head = true;                                                        // hits.hip prep_body on the wave's arrays
    const int hs = s_start[wv][r];
    const uint32_t cs = s_cs[wv][r];
    for (uint32_t j = r; j-- > 0;) {
      if (s_start[wv][j] + HIT_MAX_LEN - 1 - hs < a.max_overlap) break;
      if (s_cs[wv][j] == cs && s_end[wv][j] - hs >= a.max_overlap) { head = false; break; }
    }
    s_head[wv][r] = head ? 1 : 0; s_keep[wv][r] = 0; s_done[wv][r] = 0;
  }
  wave_lds_sync();
  if (r < nA && head && (int64_t)s_start[wv][r] >= known_from + HIT_MAX_LEN) {   // hits.hip cluster_body: the reference's loop, SR:661-671
    const uint32_t cs = s_cs[wv][r];
    auto next = [&](uint32_t j) { for (j++; j < nA; j++) if (s_cs[wv][j] == cs) return j; return nA; };
    uint32_t j = r;
    for (;;) {
      const uint32_t hit = j;
      s_done[wv][hit] = 1;
      j = next(j);
      const int hs = s_start[wv][hit], he = s_end[wv][hit], hsc = s_score[wv][hit];
      bool more = false;
      int ov = 0;
      for (;;) {
        more = j < nA && !s_head[wv][j];
        if (!more) break;
        ov = max(0, min(s_end[wv][j], he) - max(s_start[wv][j], hs));   // RH:141-144
        if (!(ov >= a.max_overlap && s_score[wv][j] <= hsc)) break;
        s_done[wv][j] = 1;                                               // swallowed
        j = next(j);
      }
      if (!more || ov < a.max_overlap) s_keep[wv][hit] = 1;
      if (!more) break;
    }
  }
  wave_lds_sync();
  const bool mine = r < nA && (int64_t)s_start[wv][r] >= lo && (int64_t)s_start[wv][r] < hi;
  if (__ballot(mine && !s_done[wv][r]) != 0) { decline(BIN_FLAG_HALO); return; }
  const unsigned long long kept = __ballot(mine && s_keep[wv][r] != 0);
  const uint32_t n_rows = (uint32_t)__popcll(kept);
  if (n_rows > BIN_ROWS) { decline(BIN_FLAG_CROWDED); return; }

  // ---- 5. the kept hits of the bin in final order: which alignment, and the length of the row's middle part ----
  uint32_t bytes = 0, k = 0;
  bool bad_row = false;
  for (unsigned long long rest = kept; rest != 0; rest &= rest - 1, k++) {
    const int pos = __ffsll((long long)rest) - 1;                       // sorted position
    const int from = (int)s_idx[wv][pos];                               // the lane that holds its HitRec
    HitRec hh;                                                          // wave-uniform, in scalar registers (build_middle's set_lane wants them there)
    auto from_lane = [&](int v) { return __builtin_amdgcn_readfirstlane(__shfl(v, from)); };
    hh.contig = from_lane(h.contig); hh.start = from_lane(h.start); hh.end = from_lane(h.end); hh.gstart = from_lane(h.gstart);
    hh.gend = from_lane(h.gend); hh.score = from_lane(h.score); hh.rh_end = from_lane(h.rh_end); hh.minus = (uint32_t)from_lane((int)h.minus);
    const uint32_t rsrc = (uint32_t)from_lane((int)hsrc);
    const auto* rp = uniform_ptr(a.raw) + rsrc;
    RowIn rin;
    const auto* ow = (const __attribute__((address_space(4))) uint32_t*)rp->ops;
    rin.w0 = ow[0]; rin.w1 = ow[1]; rin.w2 = ow[2]; rin.w3 = ow[3]; rin.w4 = ow[4];
    rin.n_ops = rp->n_ops; rin.pam = rp->pam; rin.offset = rp->offset; rin.pam_x = rp->pam_x;
    const auto* gp = uniform_ptr(a.guides) + rp->guide;
    RowGuide g;
    g.L = gp->L; g.pam5 = gp->pam5; g.pam_len = rin.pam >= 0 ? gp->pam_len[rin.pam] : 0;
    const int len = build_middle<false>(nullptr, nullptr, m, s_blob, rin, hh, g, lane);
    if (len < 0) { bad_row = true; break; }
    const uint32_t name_len = uniform_ptr(m.name_off)[hh.contig + 1] - uniform_ptr(m.name_off)[hh.contig];
    bytes += m.rc.head_len + name_len + 1u + (uint32_t)len + m.rc.tail_len;
    if (lane == 0) a.rows[(size_t)rel * BIN_ROWS + k] = BinRow{rsrc, (uint32_t)len};
  }
  if (bad_row) { decline(BIN_FLAG_ROW); return; }
  finish(n_rows, bytes, nA_own);
}

struct RowsArgs {
  RowConstDev rc;
  const char* names;
  char* text;
  unsigned long long text_cap;
  const uint32_t* counters;     // the lane's eight counters, posted with the result
  uint32_t* box;                // mailbox (device view)
  uint32_t seq;
  uint32_t n_chunks;
};

// One wave per bin: where the bin's text starts = bytes of the chunks before its chunk + bytes of the bins before it in its chunk;
// then row by row: the middle part in the wave's LDS line (build_middle), head | chromosome | middle | tail at the row's final place.
// The first wave of the grid also posts the totals (final since bin_hits_kernel ended) to the host when it starts.
__global__ __launch_bounds__(64 * BIN_WAVES) void bin_rows_kernel(BinArgs a, MidArgs m, RowsArgs o) {
  CALITAS_TAIL_PRIO();
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // per wave: line | fwd; then the constant strings
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* line = lds + wv * (MID_LINE + MID_FWD);
  uint8_t* fwd = line + MID_LINE;
  uint8_t* blob = lds + BIN_WAVES * (MID_LINE + MID_FWD);
  for (uint32_t i = threadIdx.x; i < m.blob_bytes; i += 64 * BIN_WAVES) blob[i] = (uint8_t)m.blob[i];
  __syncthreads();
  const uint32_t rel = blockIdx.x * BIN_WAVES + wv;
  // totals, and the bytes ahead of this bin's chunk
  const uint32_t my_chunk = rel >> CHUNK_SHIFT;
  unsigned long long tot = 0, before = 0;
  uint32_t rows_tot = 0, acc_tot = 0;
  for (uint32_t ch = (uint32_t)lane; ch < o.n_chunks; ch += 64) {
    const unsigned long long v = a.chunk_bytes[ch];
    tot += v;
    if (ch < my_chunk) before += v;
    if (rel == 0) { rows_tot += a.chunk_rows[ch]; acc_tot += a.chunk_acc[ch]; }
  }
  tot = wave_sum_u64(tot); before = wave_sum_u64(before);
  const uint32_t flags = *a.flags | (tot > o.text_cap ? BIN_FLAG_TEXT : 0u);
  if (rel == 0) {
    rows_tot = (uint32_t)wave_sum_u64(rows_tot); acc_tot = (uint32_t)wave_sum_u64(acc_tot);
    if (lane == 0) {
      for (int i = 0; i < 8; i++) o.box[BIN_BOX_COUNTERS + i] = __hip_atomic_load(o.counters + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      o.box[BIN_BOX_ROWS] = rows_tot; o.box[BIN_BOX_BYTES] = (uint32_t)tot; o.box[BIN_BOX_BYTES + 1] = (uint32_t)(tot >> 32);
      o.box[BIN_BOX_FLAGS] = flags; o.box[BIN_BOX_ACCEPTED] = acc_tot;
      __threadfence_system();
      __hip_atomic_store(o.box, o.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (rel >= a.n_bins || flags != 0) return;
  const uint32_t n = a.bin_rows[rel];
  if (n == 0) return;
  {
    unsigned long long part = 0;
    for (uint32_t x = (my_chunk << CHUNK_SHIFT) + (uint32_t)lane; x < rel; x += 64) part += a.bin_bytes[x];
    before += wave_sum_u64(part);
  }
  const uint8_t* head = reinterpret_cast<const uint8_t*>(m.blob) + o.rc.head_off;     // constant pieces straight from global memory (L2-resident)
  const uint8_t* tail = reinterpret_cast<const uint8_t*>(m.blob) + o.rc.tail_off;
  unsigned long long at = before;
  for (uint32_t k = 0; k < n; k++) {
    const auto* row = uniform_ptr(a.rows) + ((size_t)rel * BIN_ROWS + k);
    const uint32_t ridx = row->raw, want = row->len;
    const auto* rp = uniform_ptr(a.raw) + ridx;
    RowIn rin;
    const auto* ow = (const __attribute__((address_space(4))) uint32_t*)rp->ops;
    rin.w0 = ow[0]; rin.w1 = ow[1]; rin.w2 = ow[2]; rin.w3 = ow[3]; rin.w4 = ow[4];
    rin.n_ops = rp->n_ops; rin.pam = rp->pam; rin.offset = rp->offset; rin.pam_x = rp->pam_x;
    HitRec h;                                             // the same coordinates bin_hits_kernel derived (every lane computes them: uniform)
    {
      const HitRec hv = hit_record(a.raw + ridx, a.guides, a.win_base, a.win);
      h.contig = __builtin_amdgcn_readfirstlane(hv.contig); h.start = __builtin_amdgcn_readfirstlane(hv.start);
      h.end = __builtin_amdgcn_readfirstlane(hv.end); h.gstart = __builtin_amdgcn_readfirstlane(hv.gstart);
      h.gend = __builtin_amdgcn_readfirstlane(hv.gend); h.score = __builtin_amdgcn_readfirstlane(hv.score);
      h.rh_end = __builtin_amdgcn_readfirstlane(hv.rh_end); h.minus = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.minus);
    }
    const auto* gp = uniform_ptr(a.guides) + rp->guide;
    RowGuide g;
    g.L = gp->L; g.pam5 = gp->pam5; g.pam_len = rin.pam >= 0 ? gp->pam_len[rin.pam] : 0;
    wave_lds_sync();                                      // the copy-out of the previous row is done with line[]
    const int len = build_middle<true>(line, fwd, m, blob, rin, h, g, lane);
    if (len < 0 || (uint32_t)len != want) {               // cannot happen: both kernels run the same arithmetic
      if (lane == 0) __hip_atomic_fetch_or(o.box + BIN_BOX_LATE, BIN_FLAG_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    wave_lds_sync();
    const uint32_t nb = uniform_ptr(m.name_off)[h.contig], nl = uniform_ptr(m.name_off)[h.contig + 1] - nb;
    const uint32_t s0 = o.rc.head_len, s1 = s0 + nl + 1, s2 = s1 + (uint32_t)len, total = s2 + o.rc.tail_len;
    char* dst = o.text + at;
    for (uint32_t x = (uint32_t)lane; x < total; x += 64) {
      uint8_t ch;
      if (x < s0) ch = head[x];
      else if (x < s1) ch = (x - s0 < nl) ? (uint8_t)o.names[nb + x - s0] : (uint8_t)'\t';
      else if (x < s2) ch = line[x - s1];
      else ch = tail[x - s2];
      dst[x] = (char)ch;
    }
    at += total;
  }
}

template <typename T>
hipError_t grow_to(T** p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  (void)hipFree(*p); *p = nullptr; cap = 0;
  hipError_t e = hipMalloc((void**)p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  return e;
}

}  // namespace

struct BinnedWork {
  uint32_t* bin_idx = nullptr; size_t bin_idx_cap = 0;
  BinRow* rows = nullptr; size_t rows_cap = 0;
  // one allocation for everything that is cleared per call: bin_count[n_bins] | chunk_bytes | chunk_rows | chunk_acc | flags
  uint8_t* clear = nullptr; size_t clear_cap = 0;
  uint32_t *bin_count = nullptr, *chunk_rows = nullptr, *chunk_acc = nullptr, *flags = nullptr;
  unsigned long long* chunk_bytes = nullptr;
  uint32_t *bin_rows = nullptr, *bin_bytes = nullptr; size_t bin_rows_cap = 0, bin_bytes_cap = 0;
  uint32_t n_bins = 0, n_chunks = 0;
};

void binned_destroy(BinnedWork* w) {
  if (!w) return;
  (void)hipFree(w->bin_idx); (void)hipFree(w->rows); (void)hipFree(w->clear); (void)hipFree(w->bin_rows); (void)hipFree(w->bin_bytes);
  delete w;
}

int binned_shift(int window_size) {
  if (window_size < 2 * HIT_MAX_LEN) return 0;
  int shift = 13;                                                     // 8 kb: room for two default windows of context on the left
  while (shift < 20 && ((int64_t)1 << shift) < 2 * (int64_t)window_size + 4 * HIT_MAX_LEN) shift++;
  return ((int64_t)1 << shift) >= 2 * (int64_t)window_size + 4 * HIT_MAX_LEN ? shift : 0;
}

#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)

hipError_t binned_prepare(BinnedWork** pw, uint32_t n_bins, hipStream_t stream) {
  if (!*pw) *pw = new BinnedWork();
  BinnedWork& w = **pw;
  hipError_t e;
  const uint32_t n_chunks = (n_bins >> CHUNK_SHIFT) + 1;
  TRY(grow_to(&w.bin_idx, w.bin_idx_cap, (size_t)n_bins * BIN_CAP));
  TRY(grow_to(&w.rows, w.rows_cap, (size_t)n_bins * BIN_ROWS));
  TRY(grow_to(&w.bin_rows, w.bin_rows_cap, (size_t)n_bins));
  TRY(grow_to(&w.bin_bytes, w.bin_bytes_cap, (size_t)n_bins));
  // chunk_bytes first (8-byte aligned), then the 32-bit arrays; the whole block is a multiple of 16 bytes (MI355X_MICROARCH: memset sizes)
  const size_t bytes = (((size_t)n_chunks * 8 + (size_t)n_chunks * 4 * 2 + (size_t)n_bins * 4 + 4) + 15) & ~(size_t)15;
  TRY(grow_to(&w.clear, w.clear_cap, bytes));
  w.chunk_bytes = reinterpret_cast<unsigned long long*>(w.clear);
  w.chunk_rows = reinterpret_cast<uint32_t*>(w.clear + (size_t)n_chunks * 8);
  w.chunk_acc = w.chunk_rows + n_chunks;
  w.flags = w.chunk_acc + n_chunks;
  w.bin_count = w.flags + 1;
  w.n_bins = n_bins; w.n_chunks = n_chunks;
  return hipMemsetAsync(w.clear, 0, bytes, stream);
}

void binned_fill_align_args(const BinnedWork* w, const BinnedGeometry& geo, AlignArgs& aa) {
  aa.bin_idx = w->bin_idx; aa.bin_count = w->bin_count; aa.bin_base = geo.d_bin_base; aa.bin_first = geo.bin_first; aa.bin_shift = geo.bin_shift;
  aa.bin_cap = BIN_CAP;
}

const char* binned_text(const HitsWork* hits) { return hits ? hits->text : nullptr; }

static hipError_t launch_rows(BinnedWork& w, HitsWork& hw, const BinArgs& ba, const MidArgs& ma, const uint32_t* d_counters, hipStream_t stream,
                              Mailbox* post, hipEvent_t ev_start, hipEvent_t ev_done) {
  hipError_t e;
  TRY(mailbox_open(*post));
  RowsArgs ro{};
  ro.rc = hw.rc; ro.names = hw.names; ro.text = hw.text; ro.text_cap = hw.text_cap; ro.counters = d_counters; ro.box = post->dev; ro.seq = ++post->seq;
  ro.n_chunks = w.n_chunks;
  post->host[BIN_BOX_LATE] = 0;                              // raised by any wave while rows are written; read when the stream is done
  const uint32_t lds = BIN_WAVES * (MID_LINE + MID_FWD) + (uint32_t)((hw.blob_bytes + 15) & ~(size_t)15);
  const unsigned grid = (std::max<uint32_t>(w.n_bins, 1u) + BIN_WAVES - 1) / BIN_WAVES;
  hipExtLaunchKernelGGL(bin_rows_kernel, dim3(grid), dim3(64 * BIN_WAVES), lds, stream, ev_start, ev_done, 0, ba, ma, ro);
  return hipGetLastError();
}

static void fill_args(BinnedWork& w, HitsWork& hw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                      const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, BinArgs& ba, MidArgs& ma) {
  ba = BinArgs{};
  ba.raw = d_raw; ba.bin_idx = w.bin_idx; ba.bin_count = w.bin_count; ba.bin_base = geo.d_bin_base; ba.n_contigs = geo.n_contigs; ba.bin_first = geo.bin_first;
  ba.n_bins = geo.n_bins; ba.bin_shift = geo.bin_shift; ba.guides = d_guides; ba.win_base = d_win_base; ba.win = d_win;
  ba.W = p.window_size; ba.step = p.step; ba.max_total_diffs = p.max_total_diffs; ba.max_overlap = p.max_overlap;
  ba.rows = w.rows; ba.bin_rows = w.bin_rows; ba.bin_bytes = w.bin_bytes; ba.chunk_bytes = w.chunk_bytes; ba.chunk_rows = w.chunk_rows;
  ba.chunk_acc = w.chunk_acc; ba.flags = w.flags;
  const uint32_t n_max = (uint32_t)std::min<int>(MID_COLS, std::max(1, p.max_ops));
  ma = MidArgs{};
  ma.ref = ref; ma.rc = hw.rc; ma.blob = hw.blob; ma.name_off = hw.name_off; ma.guides = d_guides;
  ma.mid_bound = (6 * n_max + 128 + 3) & ~3u; ma.n_max = n_max; ma.blob_bytes = (uint32_t)hw.blob_bytes;
}

hipError_t binned_run(BinnedWork* pw, HitsWork** phw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                      const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters, hipStream_t stream,
                      Mailbox* post, hipEvent_t ev_hits_done, hipEvent_t ev_rows_start, hipEvent_t ev_rows_done) {
  if (!pw || !*phw || !(*phw)->prepared) return hipErrorInvalidValue;
  BinnedWork& w = *pw;
  HitsWork& hw = **phw;
  hw.prepared = false;
  hipError_t e;
  {
    size_t first_guess = (size_t)32 << 20;                     // BIN_FLAG_TEXT asks for more
    if (const char* env = std::getenv("CALITAS_BINNED_TEXT_KB")) first_guess = (size_t)std::max(1, std::atoi(env)) << 10;   // tests: force the regrow path
    if (hw.text_cap < first_guess) TRY(grow(&hw.text, hw.text_cap, first_guess));
  }
  BinArgs ba; MidArgs ma;
  fill_args(w, hw, geo, ref, d_raw, d_guides, d_win_base, d_win, p, ba, ma);
  const uint32_t lds = (uint32_t)((hw.blob_bytes + 15) & ~(size_t)15);
  if (lds + BIN_WAVES * (MID_LINE + MID_FWD) > 48 * 1024) return hipErrorInvalidValue;      // absurdly long parameter strings: the caller takes the general path
  const unsigned grid = (std::max<uint32_t>(geo.n_bins, 1u) + BIN_WAVES - 1) / BIN_WAVES;
  hipExtLaunchKernelGGL(bin_hits_kernel, dim3(grid), dim3(64 * BIN_WAVES), lds, stream, nullptr, ev_hits_done, 0, ba, ma);
  TRY(hipGetLastError());
  return launch_rows(w, hw, ba, ma, d_counters, stream, post, ev_rows_start, ev_rows_done);
}

hipError_t binned_rerun_rows(BinnedWork* pw, HitsWork** phw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                             const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters,
                             uint64_t bytes, hipStream_t stream, Mailbox* post, hipEvent_t ev_rows_done) {
  if (!pw || !*phw) return hipErrorInvalidValue;
  BinnedWork& w = *pw;
  HitsWork& hw = **phw;
  hipError_t e;
  TRY(hipStreamSynchronize(stream));                           // nobody is writing the old buffer any more
  TRY(grow(&hw.text, hw.text_cap, (size_t)bytes + (size_t)(bytes / 4)));
  BinArgs ba; MidArgs ma;
  fill_args(w, hw, geo, ref, d_raw, d_guides, d_win_base, d_win, p, ba, ma);
  return launch_rows(w, hw, ba, ma, d_counters, stream, post, nullptr, ev_rows_done);
}

#undef TRY

}  // namespace calitas
