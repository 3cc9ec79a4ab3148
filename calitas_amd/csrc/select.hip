// select.hip -- the per-window stage of SequentialGuideAligner.align (SequentialGuideAligner.scala:315-320) on the GPU.
//
// trace_kernel appends extended alignments in arbitrary order.  Here they are
//   1. keyed by (guide, contig, window | strand list, end column, PAM) and radix-sorted (rocPRIM): inside a window that is
//      exactly the reference's enumeration order (fgbio emits ascending end columns, extendAndFilterRight keeps PAM order);
//   2. annotated with what the filter looks at (score, gap bases, edits, contig start/end);
//   3. filtered window by window, one lane per window: stable order by (score desc, gap bases asc) (GuideAlignment.scala:
//      125-129), keep if edits <= maxTotalDiffs and no kept alignment of the same strand overlaps by more than maxOverlap;
//   4. compacted in window order with an exclusive scan, so the host receives only accepted alignments, already ordered.
// Windows with more than GROUP_MAX alignments (satellite repeats, PAM-less dense searches) go to filter_big_kernel, one wave
// per window; only beyond 65 536 records or 4 096 kept alignments in one window does a flag send the caller to the host
// implementation of the same stage for this search.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "select.hpp"

namespace calitas {

namespace {

constexpr int GROUP_MAX = 256;
constexpr unsigned GROUP_SHIFT = 18;   // key bits below the (guide, contig, window) group id

struct Derived { int32_t start, end, score; uint16_t gaps, edits; };

__global__ void key_kernel(const RawAln* raw, uint32_t n, const GuideDev* guides, uint64_t* keys, uint32_t* vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RawAln r = raw[i];
  const uint32_t pam5 = guides[r.guide].pam5;
  const uint64_t list = pam5 ? (r.dir == 1 ? 0 : 1) : (r.dir == 0 ? 0 : 1);   // 0 = forward-strand list (SGA:316)
  keys[i] = ((uint64_t)r.guide << 58) | ((uint64_t)r.contig << 40) | ((uint64_t)r.window_k << 18) | (list << 17) |
            ((uint64_t)r.t_end_guide << 4) | (uint64_t)(r.pam + 1);
  vals[i] = i;
}

__global__ void derive_kernel(const RawAln* raw, const uint32_t* vals, uint32_t n, const GuideDev* guides, const uint64_t* win_base,
                              const int2* win, Derived* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RawAln* rp = raw + vals[i];
  struct { uint32_t contig, window_k; int32_t score; int t_start, t_end_guide, dir, guide, pam, offset, n_ops; uint32_t pam_x; } r;
  r.contig = rp->contig; r.window_k = rp->window_k; r.score = rp->score; r.t_start = rp->t_start; r.t_end_guide = rp->t_end_guide;
  r.dir = rp->dir; r.guide = rp->guide; r.pam = rp->pam; r.offset = rp->offset; r.n_ops = rp->n_ops; r.pam_x = rp->pam_x;
  const OpCounts oc = count_ops(load_ops_words(rp->ops), r.n_ops);     // no per-op loop over a private copy of the record
  int diffs = oc.non_eq, gaps = oc.gaps;
  int pam_len = 0;
  if (r.pam >= 0) { pam_len = guides[r.guide].pam_len[r.pam]; diffs += r.offset + __popc((unsigned)r.pam_x); gaps += r.offset; }
  const int2 w = win[win_base[r.contig] + r.window_k];
  const int start_s = (int)r.t_start - 1, end_s = (int)r.t_end_guide + r.offset + pam_len;
  Derived d;
  if (r.dir == 0) { d.start = w.x + start_s; d.end = w.x + end_s; }
  else            { d.start = w.y - end_s;   d.end = w.y - start_s; }
  d.score = r.score; d.gaps = (uint16_t)gaps; d.edits = (uint16_t)diffs;
  out[i] = d;
}

// One lane per sorted position; only group heads work.  kept[s] = number of survivors of the group starting at s (0 for
// non-heads); their sorted positions, in output order, go to out_pos[s .. s + kept[s]).
__global__ void filter_kernel(const uint64_t* keys, const Derived* der, uint32_t n, int max_total_diffs, int max_overlap,
                              uint8_t* taken, uint32_t* kept, uint32_t* out_pos, uint32_t* flags, uint32_t* big) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint64_t gid = keys[s] >> GROUP_SHIFT;
  if (s > 0 && (keys[s - 1] >> GROUP_SHIFT) == gid) { kept[s] = 0; return; }
  uint32_t e = s + 1;
  while (e < n && (keys[e] >> GROUP_SHIFT) == gid) e++;
  if (e - s > (uint32_t)GROUP_MAX) { big[atomicAdd(flags + 1, 1u)] = s; kept[s] = 0; return; }   // left to filter_big_kernel
  uint32_t mid = s;                                   // first member of the reverse-strand list
  while (mid < e && !((keys[mid] >> 17) & 1)) mid++;
  uint32_t nk = 0;
  for (int list = 0; list < 2; list++) {
    const uint32_t lo = list ? mid : s, hi = list ? e : mid;
    const uint32_t first_kept = nk;                   // overlaps are only tested against the same strand (SGA:317)
    for (uint32_t round = lo; round < hi; round++) {
      int best = -1, best_score = 0, best_gaps = 0;
      for (uint32_t m = lo; m < hi; m++) {
        if (taken[m]) continue;
        const int sc = der[m].score, gp = der[m].gaps;
        if (best < 0 || sc > best_score || (sc == best_score && gp < best_gaps)) { best = (int)m; best_score = sc; best_gaps = gp; }
      }
      if (best < 0) break;
      taken[best] = 1;
      if ((int)der[best].edits > max_total_diffs) continue;
      bool clash = false;
      for (uint32_t k = first_kept; k < nk; k++) {
        const Derived& b = der[out_pos[s + k]];
        const int o = min(der[best].end, b.end) - max(der[best].start, b.start);   // GA:119-122
        if (o > max_overlap) { clash = true; break; }
      }
      if (!clash) out_pos[s + nk++] = (uint32_t)best;
    }
  }
  kept[s] = nk;
}

// Window groups with more than GROUP_MAX records (dense repeats, permissive limits): one wave per group, same greedy.  The
// "taken" bits and the kept intervals of the current strand list live in LDS; a group beyond those capacities raises the flag.
constexpr uint32_t BIG_MAX = 1u << 16;     // records per group
constexpr uint32_t BIG_KEPT = 4096;        // kept alignments per strand list

__global__ __launch_bounds__(64) void filter_big_kernel(const uint64_t* keys, const Derived* der, uint32_t n, int max_total_diffs, int max_overlap,
                                                        uint32_t* kept, uint32_t* out_pos, uint32_t* flags, const uint32_t* big) {
  __shared__ uint32_t s_taken[BIG_MAX / 32];
  __shared__ int s_ks[BIG_KEPT], s_ke[BIG_KEPT];
  const uint32_t lane = threadIdx.x;
  const uint32_t n_big = flags[1];
  for (uint32_t g = blockIdx.x; g < n_big; g += gridDim.x) {
    const uint32_t s = big[g];
    const uint64_t gid = keys[s] >> GROUP_SHIFT;
    uint32_t e = s + 1;
    for (;;) {                                           // end of the group, 64 positions per probe
      const uint32_t q = e + lane;
      const unsigned long long m = __ballot(q < n && (keys[q] >> GROUP_SHIFT) == gid);
      if (m == ~0ull) { e += 64; continue; }
      e += (uint32_t)__ffsll((long long)~m) - 1;
      break;
    }
    uint32_t mid = s;                                    // first member of the reverse-strand list
    for (;;) {
      const uint32_t q = mid + lane;
      const unsigned long long m = __ballot(q < e && !((keys[q] >> 17) & 1));
      if (m == ~0ull) { mid += 64; continue; }
      mid += (uint32_t)__ffsll((long long)~m) - 1;
      break;
    }
    if (e - s > BIG_MAX) { if (lane == 0) { atomicOr(flags, 1u); kept[s] = 0; } continue; }
    for (uint32_t i = lane; i < (e - s + 31) / 32; i += 64) s_taken[i] = 0;
    __syncthreads();
    uint32_t nk = 0;
    bool overflow = false;
    for (int list = 0; list < 2 && !overflow; list++) {
      const uint32_t lo = list ? mid : s, hi = list ? e : mid;
      const uint32_t first_kept = nk;
      for (;;) {
        // best remaining record: score desc, gap bases asc, enumeration order (GA:125-129, stable sort)
        unsigned long long bk = 0;
        for (uint32_t m = lo + lane; m < hi; m += 64) {
          const uint32_t rel = m - s;
          if ((s_taken[rel >> 5] >> (rel & 31)) & 1u) continue;
          const Derived d = der[m];
          const unsigned long long k = ((unsigned long long)(uint32_t)(d.score + (1 << 22)) << 40) | ((unsigned long long)(0xFFFFu - d.gaps) << 24) |
                                       (unsigned long long)(0xFFFFFFu - (m - lo));
          bk = k > bk ? k : bk;
        }
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long o = __shfl_xor(bk, off);
          bk = o > bk ? o : bk;
        }
        if (bk == 0) break;
        const uint32_t best = lo + (0xFFFFFFu - (uint32_t)(bk & 0xFFFFFFu));
        if (lane == 0) { const uint32_t rel = best - s; s_taken[rel >> 5] |= 1u << (rel & 31); }
        const Derived b = der[best];
        bool clash = false;
        if ((int)b.edits <= max_total_diffs) {
          bool mine = false;
          for (uint32_t k = first_kept + lane; k < nk; k += 64) {
            const int o = min(b.end, s_ke[k - first_kept]) - max(b.start, s_ks[k - first_kept]);   // GA:119-122
            mine = mine || o > max_overlap;
          }
          clash = __ballot(mine) != 0;
          if (!clash) {
            if (nk - first_kept >= BIG_KEPT) { overflow = true; break; }
            if (lane == 0) { out_pos[s + nk] = best; s_ks[nk - first_kept] = b.start; s_ke[nk - first_kept] = b.end; }
            nk++;
          }
        }
        __syncthreads();
      }
    }
    if (lane == 0) { if (overflow) { atomicOr(flags, 1u); kept[s] = 0; } else kept[s] = nk; }
    __syncthreads();
  }
}

__global__ void gather_kernel(const RawAln* raw, const uint32_t* vals, const uint32_t* kept, const uint32_t* offs, const uint32_t* out_pos,
                              uint32_t n, RawAln* final_out, uint32_t* counts) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t nk = kept[s];
  for (uint32_t r = 0; r < nk; r++) final_out[offs[s] + r] = raw[vals[out_pos[s + r]]];
  if (s == n - 1) counts[0] = offs[s] + nk;           // total survivors
}

template <typename T>
hipError_t grow(T** p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  (void)hipFree(*p); *p = nullptr; cap = 0;
  hipError_t e = hipMalloc((void**)p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  return e;
}

}  // namespace

struct SelectWork {
  uint64_t *keys = nullptr, *keys2 = nullptr; size_t keys_cap = 0, keys2_cap = 0;
  uint32_t *vals = nullptr, *vals2 = nullptr, *kept = nullptr, *offs = nullptr, *out_pos = nullptr;
  size_t vals_cap = 0, vals2_cap = 0, kept_cap = 0, offs_cap = 0, out_pos_cap = 0;
  Derived* der = nullptr; size_t der_cap = 0;
  uint8_t* taken = nullptr; size_t taken_cap = 0;
  RawAln* final_out = nullptr; size_t final_cap = 0;
  void* temp = nullptr; size_t temp_cap = 0;
  uint32_t* big = nullptr; size_t big_cap = 0;
  uint32_t* counts = nullptr;   // [0] survivors, [1] flags, [2] window groups left to filter_big_kernel
};

void select_destroy(SelectWork* w) {
  if (!w) return;
  (void)hipFree(w->keys); (void)hipFree(w->keys2); (void)hipFree(w->vals); (void)hipFree(w->vals2); (void)hipFree(w->kept);
  (void)hipFree(w->offs); (void)hipFree(w->out_pos); (void)hipFree(w->der); (void)hipFree(w->taken); (void)hipFree(w->final_out);
  (void)hipFree(w->temp); (void)hipFree(w->counts); (void)hipFree(w->big);
  delete w;
}

bool select_supported(uint64_t n_contigs, uint64_t max_windows_per_contig, int window_size, int n_guides) {
  return n_contigs < (1ull << 18) && max_windows_per_contig < (1ull << 22) && window_size < (1 << 13) && n_guides <= 64;
}

hipError_t select_run(SelectWork** pw, const RawAln* d_raw, uint32_t n_raw, const GuideDev* d_guides, const uint64_t* d_win_base,
                      const int2* d_win, int max_total_diffs, int max_overlap, hipStream_t stream, const RawAln** d_final,
                      const uint32_t** d_counts) {
  if (!*pw) *pw = new SelectWork();
  SelectWork& w = **pw;
  hipError_t e;
  const size_t n = n_raw;
#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  if (!w.counts) TRY(hipMalloc((void**)&w.counts, 3 * sizeof(uint32_t)));
  TRY(hipMemsetAsync(w.counts, 0, 3 * sizeof(uint32_t), stream));
  *d_final = nullptr; *d_counts = w.counts;
  if (n == 0) return hipSuccess;
  TRY(grow(&w.keys, w.keys_cap, n)); TRY(grow(&w.keys2, w.keys2_cap, n));
  TRY(grow(&w.vals, w.vals_cap, n)); TRY(grow(&w.vals2, w.vals2_cap, n));
  TRY(grow(&w.kept, w.kept_cap, n)); TRY(grow(&w.offs, w.offs_cap, n)); TRY(grow(&w.out_pos, w.out_pos_cap, n));
  TRY(grow(&w.der, w.der_cap, n)); TRY(grow(&w.taken, w.taken_cap, n)); TRY(grow(&w.big, w.big_cap, n / GROUP_MAX + 1)); TRY(grow(&w.final_out, w.final_cap, n));
  size_t t1 = 0, t2 = 0;
  TRY(rocprim::radix_sort_pairs(nullptr, t1, w.keys, w.keys2, w.vals, w.vals2, n, 0, 64, stream));
  TRY(rocprim::exclusive_scan(nullptr, t2, w.kept, w.offs, 0u, n, rocprim::plus<uint32_t>(), stream));
  {
    size_t need = std::max(t1, t2);
    if (need > w.temp_cap) { (void)hipFree(w.temp); w.temp = nullptr; w.temp_cap = 0; TRY(hipMalloc(&w.temp, need)); w.temp_cap = need; }
  }
  const dim3 block(256), grid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(key_kernel, grid, block, 0, stream, d_raw, n_raw, d_guides, w.keys, w.vals);
  size_t ts = w.temp_cap;
  TRY(rocprim::radix_sort_pairs(w.temp, ts, w.keys, w.keys2, w.vals, w.vals2, n, 0, 64, stream));
  hipLaunchKernelGGL(derive_kernel, grid, block, 0, stream, d_raw, (const uint32_t*)w.vals2, n_raw, d_guides, d_win_base, d_win, w.der);
  TRY(hipMemsetAsync(w.taken, 0, n, stream));
  hipLaunchKernelGGL(filter_kernel, grid, block, 0, stream, (const uint64_t*)w.keys2, (const Derived*)w.der, n_raw, max_total_diffs,
                     max_overlap, w.taken, w.kept, w.out_pos, w.counts + 1, w.big);
  hipLaunchKernelGGL(filter_big_kernel, dim3((unsigned)std::min<size_t>(n / GROUP_MAX + 1, 8192)), dim3(64), 0, stream, (const uint64_t*)w.keys2,
                     (const Derived*)w.der, n_raw, max_total_diffs, max_overlap, w.kept, w.out_pos, w.counts + 1, (const uint32_t*)w.big);
  ts = w.temp_cap;
  TRY(rocprim::exclusive_scan(w.temp, ts, w.kept, w.offs, 0u, n, rocprim::plus<uint32_t>(), stream));
  hipLaunchKernelGGL(gather_kernel, grid, block, 0, stream, d_raw, (const uint32_t*)w.vals2, (const uint32_t*)w.kept,
                     (const uint32_t*)w.offs, (const uint32_t*)w.out_pos, n_raw, w.final_out, w.counts);
  TRY(hipGetLastError());
  if (std::getenv("CALITAS_SELECT_DEBUG")) {
    TRY(hipStreamSynchronize(stream));
    std::vector<uint64_t> k(n); std::vector<uint32_t> v(n), kp(n), op(n), of(n); std::vector<Derived> d(n); std::vector<RawAln> r(n);
    TRY(hipMemcpy(k.data(), w.keys2, n * 8, hipMemcpyDeviceToHost)); TRY(hipMemcpy(v.data(), w.vals2, n * 4, hipMemcpyDeviceToHost));
    TRY(hipMemcpy(kp.data(), w.kept, n * 4, hipMemcpyDeviceToHost)); TRY(hipMemcpy(op.data(), w.out_pos, n * 4, hipMemcpyDeviceToHost));
    TRY(hipMemcpy(of.data(), w.offs, n * 4, hipMemcpyDeviceToHost)); TRY(hipMemcpy(d.data(), w.der, n * sizeof(Derived), hipMemcpyDeviceToHost));
    TRY(hipMemcpy(r.data(), d_raw, n * sizeof(RawAln), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n && i < 60; i++)
      std::fprintf(stderr, "[select] %zu key=%016llx val=%u raw(score=%d tend=%d pam=%d wk=%u dir=%d) der(score=%d start=%d end=%d gaps=%d edits=%d) kept=%u off=%u outpos=%u\n", i,
                   (unsigned long long)k[i], v[i], r[v[i]].score, r[v[i]].t_end_guide, r[v[i]].pam, r[v[i]].window_k, r[v[i]].dir, d[i].score, d[i].start, d[i].end, d[i].gaps, d[i].edits, kp[i], of[i], op[i]);
  }
#undef TRY
  *d_final = w.final_out;
  return hipSuccess;
}

}  // namespace calitas
