// api.cpp -- C ABI of libcalitas_hip.so (include/calitas_hip.h): context, reference upload, the search pipeline.
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <ctime>
#include <numeric>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/calitas_hip.h"
#include "tuning.hpp"
#include "common.hpp"
#include "fasta.hpp"
#include "kernels.hpp"
#include "parallel.hpp"
#include "post.hpp"
#include "refpack.hpp"
#include "ctx.hpp"

static std::string g_create_error;

// Output buffers (alignment arrays, hits.txt text) are tens of MB per pass; handing each one back to the OS and faulting
// a fresh one in costs milliseconds, so calitas_free parks the most recent blocks and out_alloc reuses them.
namespace {
struct BlockHeader { uint64_t magic; uint64_t capacity; uint64_t pinned; uint64_t pad; };
constexpr uint64_t kMagic = 0xCA117A5B10C0FFEEull;
std::mutex g_pool_mutex;
std::vector<BlockHeader*> g_pool;   // parked blocks
constexpr size_t kPoolBlocks = 128;    // a batch call hands out one text buffer per guide (BASELINE config 4: 96 guides)
constexpr uint64_t kPoolBytes = 24ull << 30;   // ... of ~180 MB each at hg38 size; page-locking a fresh one costs tens of milliseconds

struct Reaper {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> jobs;
  bool quit = false, busy = false;
  std::thread t;
  void wait_idle() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return jobs.empty() && !busy; });
  }
  void give(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!t.joinable()) t = std::thread([this] {
        for (;;) {
          std::function<void()> j;
          {
            std::unique_lock<std::mutex> lk2(mu);
            cv.wait(lk2, [&] { return !jobs.empty() || quit; });
            if (jobs.empty()) return;
            j = std::move(jobs.front());
            jobs.pop_front();
            busy = true;
          }
          j();
          j = nullptr;
          { std::lock_guard<std::mutex> lk2(mu); busy = false; }
          cv.notify_all();
        }
      });
      jobs.push_back(std::move(job));
    }
    cv.notify_all();
  }
  ~Reaper() {
    if (!t.joinable()) return;
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    t.join();
  }
};
Reaper g_reaper;
// Bytes of pageable blocks whose pages are still on their way back (release_block): a caller that frees a text of tens of gigabytes
// and asks for the next one at once would hold both for a moment -- on a box with a memory limit that is the OOM killer's cue --,
// so a fresh block of a gigabyte or more waits for them first (out_alloc_impl).
std::atomic<uint64_t> g_deferred_bytes{0};

void release_block_now(BlockHeader* h);
void release_block(BlockHeader* h) {
  h->magic = 0;
  if (h->pinned) { (void)hipHostFree(h); return; }
  // (a text of gigabytes: the caller's free() returns at once, the pages go back on the library's own thread -- 0.13 s per 22 GB even
  // with sixteen threads handing them back)
  if (h->capacity >= (1ull << 30) && !TUNE_GET("CALITAS_FREE_NOW")) {
    const uint64_t cap = h->capacity;
    g_deferred_bytes += cap;
    g_reaper.give([h, cap] { release_block_now(h); g_deferred_bytes -= cap; });
    return;
  }
  release_block_now(h);
}
void release_block_now(BlockHeader* h) {
  // A text of tens of gigabytes: the kernel clears pages as it takes them back (16 GB: 0.77 s inside free() on the GPU boxes, huge
  // pages or not).  MADV_DONTNEED takes the address-space lock shared, so the workers hand the pages back side by side first
  // (0.135 s, tools/thp_bench.cpp) and free() then unmaps an empty range.
  const uint64_t cap = h->capacity;
  if (cap >= (1ull << 30)) {
    const uintptr_t lo = ((uintptr_t)h + 2 * 4096) & ~(uintptr_t)4095, hi = ((uintptr_t)h + sizeof(BlockHeader) + cap) & ~(uintptr_t)4095;
    const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t per = (((hi - lo) / (size_t)T) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([=] { const uintptr_t a = lo + per * (size_t)t, b = std::min<uintptr_t>(hi, a + per); if (a < b) (void)madvise((void*)a, b - a, MADV_DONTNEED); });
    for (auto& x : th) x.join();
  }
  std::free(h);
}

// pinned: page-locked memory (hipHostMalloc) -- the destination of the text copy-back, so that the copy is one DMA at
// PCIe rate from the first use on instead of the runtime pinning pageable memory piecemeal.
void* out_alloc_impl(size_t size, bool pinned = false) {
  if (size < 1) size = 1;
  {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    int best = -1;
    for (size_t i = 0; i < g_pool.size(); i++)
      if ((g_pool[i]->pinned != 0) == pinned && g_pool[i]->capacity >= size && g_pool[i]->capacity <= 2 * size + (1u << 20) &&
          (best < 0 || g_pool[i]->capacity < g_pool[best]->capacity)) best = (int)i;
    if (best >= 0) { BlockHeader* h = g_pool[best]; g_pool.erase(g_pool.begin() + best); return h + 1; }
  }
  size_t cap = size + size / 8;
  if (!pinned && size >= (1ull << 30) && g_deferred_bytes.load() != 0) g_reaper.wait_idle();   // (see g_deferred_bytes)
  if (size >= (1u << 20) && TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] out_alloc: fresh %s block of %zu bytes\n", pinned ? "pinned" : "pageable", cap);
  BlockHeader* h = nullptr;
  if (pinned) {
    if (hipHostMalloc((void**)&h, sizeof(BlockHeader) + cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h = nullptr; pinned = false; }
  }
  if (!h) h = (BlockHeader*)std::malloc(sizeof(BlockHeader) + cap);
  if (!h) return nullptr;
  h->magic = kMagic; h->capacity = cap; h->pinned = pinned ? 1 : 0; h->pad = 0;
  return h + 1;
}

// One pageable block of a gigabyte or more stays parked as well (round 5): the text of a per-contig search -- 21.8 GB for BASELINE
// config 5's shape -- is copied into its block from a bounce buffer, and into pages nobody has touched that copy is the slowest thing the
// call does (0.6 s; 0.13 s more to hand the pages back).  The next such search takes the block as it is (calitas_out_take_big): its pages
// are there.  At most one, at most kBigPark bytes; calitas_release_parked() lets go of it and of everything else that is parked.
constexpr uint64_t kBigPark = 48ull << 30;
BlockHeader* g_big = nullptr;          // (under g_pool_mutex)

void out_free(void* p) {
  if (!p) return;
  BlockHeader* h = (BlockHeader*)p - 1;
  if (h->magic != kMagic) return;     // not ours: refuse rather than corrupt the heap
  if (!h->pinned && h->capacity >= (1ull << 30) && h->capacity <= kBigPark && !TUNE_GET("CALITAS_FREE_NOW")) {
    BlockHeader* old = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_pool_mutex);
      if (!g_big || g_big->capacity < h->capacity) { old = g_big; g_big = h; h = nullptr; }
    }
    if (old) release_block(old);
    if (!h) return;
  }
  // (a block of many gigabytes would only crowd the others out; small page-locked blocks are parked too: hipHostFree + hipHostMalloc
  // of the 11 KB text of an E. coli-sized call cost 0.2 ms per call, more than the search's kernels)
  if ((h->capacity >= (1u << 20) || h->pinned) && h->capacity <= kPoolBytes / 4) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    g_pool.push_back(h);
    uint64_t parked = 0;
    for (auto* b : g_pool) parked += b->capacity;
    if (g_pool.size() <= kPoolBlocks && parked <= kPoolBytes) return;
    size_t small = 0;                   // full: let the smallest parked block go, the big ones are the expensive ones
    for (size_t i = 1; i < g_pool.size(); i++) if (g_pool[i]->capacity < g_pool[small]->capacity) small = i;
    h = g_pool[small];
    g_pool.erase(g_pool.begin() + small);
  }
  release_block(h);
}
}  // namespace
// A pageable block of at least `size` bytes with the first `keep` bytes of p (a pageable block of this allocator, or NULL): realloc,
// so growing a multi-gigabyte text moves pages instead of copying them.
void* calitas_out_grow(void* p, size_t keep, size_t size) {
  if (!p) return out_alloc_impl(size);
  BlockHeader* h = (BlockHeader*)p - 1;
  if (h->magic != kMagic || h->pinned) return nullptr;
  if (h->capacity >= size) return p;
  (void)keep;
  const size_t cap = size + size / 4;
  BlockHeader* n = (BlockHeader*)std::realloc(h, sizeof(BlockHeader) + cap);
  if (!n) return nullptr;
  n->capacity = cap;
  if (cap >= (1ull << 30)) {   // tens of gigabytes are about to be touched for the first time: ask for huge pages (512x fewer faults)
    const uintptr_t lo = ((uintptr_t)n + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)n + sizeof(BlockHeader) + cap) & ~(uintptr_t)4095;
    if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
  }
  return n + 1;
}
// A block that turned out far larger than what it holds (a parked block of 22 GB taken for a text of 2) gives the rest back: p or its
// new address; the first `size` bytes stay.
void* calitas_out_shrink(void* p, size_t size) {
  if (!p) return p;
  BlockHeader* h = (BlockHeader*)p - 1;
  if (h->magic != kMagic || h->pinned || h->capacity < (1ull << 30) || h->capacity <= 2 * (uint64_t)size + (256ull << 20)) return p;
  const size_t cap = size + size / 16 + 4096;
  BlockHeader* n = (BlockHeader*)std::realloc(h, sizeof(BlockHeader) + cap);
  if (!n) return p;
  n->capacity = cap;
  return n + 1;
}
// The parked big block (above) if it has at least min_bytes of room, else NULL.  The caller owns it from here on (calitas_out_grow, calitas_free).
void* calitas_out_take_big(size_t min_bytes) {
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  if (!g_big || g_big->capacity < min_bytes) return nullptr;
  BlockHeader* h = g_big;
  g_big = nullptr;
  return h + 1;
}
extern "C" void calitas_release_parked(void) {
  std::vector<BlockHeader*> all;
  {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    all.swap(g_pool);
    if (g_big) { all.push_back(g_big); g_big = nullptr; }
  }
  for (BlockHeader* h : all) release_block(h);
}
void calitas_reap_later(std::function<void()> job) { g_reaper.give(std::move(job)); }
extern "C" void calitas_reap_wait(void) { g_reaper.wait_idle(); }
void* calitas_out_alloc(size_t size) { return out_alloc_impl(size); }
void* calitas_out_alloc_pinned(size_t size) { return out_alloc_impl(size, true); }
static void* out_alloc(size_t size) { return out_alloc_impl(size); }

int calitas_fail(calitas_ctx* ctx, int code, const std::string& msg) {
  // (a helper thread of a call may report on the same context as the caller's thread: the per-contig passes, the lanes' retries)
  static std::mutex err_mu;
  std::lock_guard<std::mutex> lk(err_mu);
  if (ctx) ctx->err = msg; else g_create_error = msg;
  return code;
}
static int fail(calitas_ctx* ctx, int code, const std::string& msg) { return calitas_fail(ctx, code, msg); }

static void free_reference_device(calitas_ctx* c) {
  if (c->device < 0) return;
  (void)hipFree(c->d_codes); (void)hipFree(c->d_planes); (void)hipFree(c->d_mask); (void)hipFree(c->d_runs); (void)hipFree(c->d_contigs); (void)hipFree(c->d_tiles); (void)hipFree(c->d_tile_list); (void)hipFree(c->d_win_base); (void)hipFree(c->d_win);
  c->d_win_base = nullptr; c->d_win = nullptr; c->win_cap = 0; c->win_W = c->win_step = 0;
  (void)hipFree(c->d_bin_base); c->d_bin_base = nullptr; (void)hipFree(c->d_bin_contig); c->d_bin_contig = nullptr; c->bin_shift = 0; c->bin_base.clear(); c->bin_decl_pams = -1;
  c->d_codes = c->d_mask = nullptr; c->d_planes = nullptr; c->d_runs = nullptr; c->d_contigs = nullptr; c->d_tiles = nullptr; c->d_tile_list = nullptr;
}

extern "C" {

const char* calitas_version(void) { return "calitas-hip 0.1 (gfx950)"; }

const char* calitas_switches(void) {
  static const std::string text = [] {
    std::string t;
    for (const tune::Switch& s : tune::kSwitches) { t += s.name; t += " <"; t += s.values; t += "> "; t += s.what; t += "\n"; }
    return t;
  }();
  return text.c_str();
}

const char* calitas_last_error(const calitas_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void calitas_free(void* p) { out_free(p); }

void* calitas_alloc_host(uint64_t bytes) { return out_alloc_impl((size_t)bytes, true); }

int calitas_create(int device_id, calitas_ctx** out) {
  if (!out) return fail(nullptr, CALITAS_EINVAL, "out is NULL");
  *out = nullptr;
  calitas_ctx* c = new calitas_ctx();
  c->device = device_id;
  c->pool = new WorkerPool(WorkerPool::default_threads());
  if (device_id >= 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      delete c;
      return fail(nullptr, CALITAS_ENODEV, "no HIP device available (the product path has no CPU fallback)");
    }
    if (device_id >= n) { delete c; return fail(nullptr, CALITAS_ENODEV, "device index out of range"); }
    // non-blocking, like the lanes' streams: nothing here relies on the null stream, and an application's own default-stream work
    // (PyTorch) neither waits for a search nor holds one up
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
      std::string m = hipGetErrorString(e);
      delete c;
      return fail(nullptr, CALITAS_EHIP, "device init failed: " + m);
    }
    // [0..5] sit between kernels of one device: a device-scope release is enough (the default, release to system, made whatever
    // waited for the end of a scan wait ~90 us longer); [6..7] bracket the text copy the host waits for
    for (int i = 0; i < 8; i++) (void)hipEventCreateWithFlags(&c->ev[i], i < 6 ? hipEventReleaseToDevice : hipEventDefault);
    (void)hipMalloc((void**)&c->d_counters, 8 * sizeof(uint32_t));
    (void)hipHostMalloc((void**)&c->h_counters, 8 * sizeof(uint32_t), hipHostMallocDefault);
    (void)hipMalloc((void**)&c->d_guides, sizeof(GuideDev) * MAX_GUIDES);
    (void)hipHostMalloc((void**)&c->h_guides, sizeof(GuideDev) * MAX_GUIDES, hipHostMallocDefault);
    // The aligner kernel's wavefront relies on the DPP wave shift; verify it on this device once.
    int* d = nullptr;
    int* h = nullptr;                       // page-locked: no copy of this library lands in pageable memory
    bool ok = hipMalloc((void**)&d, 64 * sizeof(int)) == hipSuccess && hipHostMalloc((void**)&h, 64 * sizeof(int), hipHostMallocDefault) == hipSuccess &&
              launch_dpp_selftest(d, c->stream) == hipSuccess &&
              hipMemcpyAsync(h, d, 64 * sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
              hipStreamSynchronize(c->stream) == hipSuccess;
    (void)hipFree(d);
    for (int i = 1; ok && i < 64; i++) ok = h[i] == (i - 1) * 7 + 3;
    if (h) (void)hipHostFree(h);
    if (!ok || !c->d_counters || !c->h_counters || !c->d_guides || !c->h_guides) {
      calitas_destroy(c);
      return fail(nullptr, CALITAS_EHIP, "device self-test failed (DPP wave_shr / allocation)");
    }
  } else if (device_id != -1) {
    delete c;
    return fail(nullptr, CALITAS_EINVAL, "device_id must be >= 0 or -1");
  }
  *out = c;
  return CALITAS_OK;
}

void calitas_destroy(calitas_ctx* c) {
  if (!c) return;
  if (c->device >= 0) {
    (void)hipSetDevice(c->device);
    free_reference_device(c);
    (void)hipFree(c->d_guides); (void)hipFree(c->d_recs); (void)hipFree(c->d_raw); (void)hipFree(c->d_counters);
    (void)hipFree(c->d_slab); (void)hipFree(c->d_items);
    calitas_destroy_lanes(c);
    if (c->side) { calitas_destroy(c->side); c->side = nullptr; }
    if (c->side2) { calitas_destroy(c->side2); c->side2 = nullptr; }
    for (void* q : c->aw.p) (void)hipFree(q);
    select_destroy(c->select);
    hits_destroy(c->hits);
    hits_destroy(c->hits_alt);
    binned_destroy(c->binned);
    if (c->scan_done) (void)hipEventDestroy(c->scan_done);
    if (c->rows_ready) (void)hipEventDestroy(c->rows_ready);
    if (c->inputs_ready) (void)hipEventDestroy(c->inputs_ready);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    mailbox_close(c->mbox);
    if (c->h_guides) (void)hipHostFree(c->h_guides);
    if (c->h_raw) (void)hipHostFree(c->h_raw);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

static int upload_reference_device(calitas_ctx* ctx) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // searches of the old reference may still be queued on the lanes' streams: nothing of it may be freed under them
  HIP_TRY(ctx, hipDeviceSynchronize());
  free_reference_device(ctx);
  const PackedRef& r = ctx->ref;
  size_t nruns = std::max<size_t>(1, r.runs.size());
  if (const char* e = TUNE_GET("CALITAS_DEVICE_BUDGET_MB")) {   // the budget a caller sharing the card sets covers the reference too
    const uint64_t want = (uint64_t)r.codes.size() * 8 + (uint64_t)r.mask.size() * 4 + nruns * sizeof(Run) + r.tiles.size() * sizeof(TileInfo);
    if (want > (uint64_t)std::atoll(e) << 20)
      return fail(ctx, CALITAS_ENOMEM, "the packed reference (" + std::to_string(want >> 20) + " MB on the device) exceeds CALITAS_DEVICE_BUDGET_MB");
  }
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_codes, r.codes.size() * 4));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_planes, r.codes.size() * 4));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_mask, r.mask.size() * 4));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_runs, nruns * sizeof(Run)));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_contigs, r.contigs.size() * sizeof(ContigInfo)));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tiles, r.tiles.size() * sizeof(TileInfo)));
  HIP_TRY(ctx, hipMemcpy(ctx->d_codes, r.codes.data(), r.codes.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_mask, r.mask.data(), r.mask.size() * 4, hipMemcpyHostToDevice));
  if (!r.runs.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_runs, r.runs.data(), r.runs.size() * sizeof(Run), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_contigs, r.contigs.data(), r.contigs.size() * sizeof(ContigInfo), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_tiles, r.tiles.data(), r.tiles.size() * sizeof(TileInfo), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tile_list, std::max<size_t>(1, r.masked_tiles.size()) * sizeof(uint32_t)));
  if (!r.masked_tiles.empty())
    HIP_TRY(ctx, hipMemcpy(ctx->d_tile_list, r.masked_tiles.data(), r.masked_tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  // the scan streams the codes as two bit-planes per 32 bases (scan_rows.hip); derived on the device, once per reference
  HIP_TRY(ctx, launch_planes(ctx->d_codes, ctx->d_planes, r.total_packed / 32, ctx->stream));
  HIP_TRY(ctx, hipDeviceSynchronize());   // the searches run on non-blocking streams, which nothing orders against these copies
  return CALITAS_OK;
}

// ctx->ref holds a freshly packed reference.  has_ref becomes true only when the device copy is complete: a failed upload
// (CALITAS_ENOMEM for a genome that does not fit) leaves a context that answers CALITAS_ESTATE, not one whose next search
// launches kernels on freed pointers.
static int upload_reference(calitas_ctx* ctx) {
  ctx->ref_serial++;
  ctx->seq_pams = -1; ctx->fit_pams = -1;   // what did (not) fit the old reference may (not) fit this one
  ctx->has_ref = false;
  if (ctx->device >= 0) {
    const int rc = upload_reference_device(ctx);
    if (rc) {
      (void)hipDeviceSynchronize();
      free_reference_device(ctx);
      ctx->ref = PackedRef();
      return rc;
    }
  }
  ctx->has_ref = true;
  return CALITAS_OK;
}

int calitas_set_reference(calitas_ctx* ctx, int32_t n_contigs, const char* const* names, const uint64_t* lengths,
                          const uint8_t* const* bases, const char* genome_build) {
  if (!ctx) return CALITAS_EINVAL;
  if (n_contigs <= 0 || !names || !lengths || !bases) return fail(ctx, CALITAS_EINVAL, "bad contig arguments");
  for (int i = 0; i < n_contigs; i++)
    if (lengths[i] > 0x7FFFFFFFull) return fail(ctx, CALITAS_EINVAL, "contigs longer than 2^31-1 bases are not supported (the reference uses Int coordinates)");
  try {
    pack_reference(ctx->ref, n_contigs, names, lengths, bases, genome_build, 0);
  } catch (std::exception& e) {
    ctx->has_ref = false;
    return fail(ctx, CALITAS_EINVAL, e.what());
  }
  return upload_reference(ctx);
}

int calitas_set_reference_fasta(calitas_ctx* ctx, const char* fasta_path) {
  if (!ctx || !fasta_path) return CALITAS_EINVAL;
  FastaData fd;
  std::string e = read_fasta(fasta_path, fd);
  if (!e.empty()) return fail(ctx, CALITAS_EIO, e);
  std::vector<const char*> names;
  std::vector<uint64_t> lens;
  std::vector<const uint8_t*> bases;
  for (size_t i = 0; i < fd.names.size(); i++) {
    names.push_back(fd.names[i].c_str());
    lens.push_back(fd.seqs[i].size());
    bases.push_back(reinterpret_cast<const uint8_t*>(fd.seqs[i].data()));
  }
  return calitas_set_reference(ctx, (int32_t)names.size(), names.data(), lens.data(), bases.data(), fd.genome_build.c_str());
}


// ---- persistent packed index: skips re-packing 3 GB of ASCII on repeated runs ----
}  // extern "C"
namespace {
struct IndexHeader {
  char magic[8];            // "CALIDX02"
  uint32_t chunk, n_contigs;
  uint64_t tile, total_packed, total_bases, n_runs, n_tiles, n_masked, build_len, names_len;
  uint64_t layout;          // kIndexLayout: a file written by a build with another packed layout is refused, not misread
  uint64_t checksum;        // index_checksum() over contigs, codes, mask, runs, tiles, masked_tiles
};
constexpr uint64_t kIndexLayout = 2 | ((uint64_t)LANES_PER_TILE << 8) | ((uint64_t)sizeof(Run) << 24) | ((uint64_t)sizeof(TileInfo) << 32) | ((uint64_t)sizeof(ContigInfo) << 40);

// 64-bit multiply-xor hash over 8-byte words in four independent lanes (memory-bound; the arrays are 4-byte aligned and a multiple of 4 bytes long)
uint64_t hash_bytes(const void* p, size_t n, uint64_t seed) {
  const unsigned char* b = (const unsigned char*)p;
  uint64_t h[4] = {seed ^ 0x9E3779B97F4A7C15ull, seed ^ 0xC2B2AE3D27D4EB4Full, seed ^ 0x165667B19E3779F9ull, seed ^ 0x27D4EB2F165667C5ull};
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    std::memcpy(w, b + i, 32);
    for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ w[k]) * 0x9FB21C651E98DF25ull; h[k] ^= h[k] >> 29; }
  }
  uint64_t tail = 0x1234567ull + n;
  for (; i < n; i++) tail = (tail ^ b[i]) * 0x100000001B3ull;
  uint64_t r = tail;
  for (int k = 0; k < 4; k++) { r = (r ^ h[k]) * 0xFF51AFD7ED558CCDull; r ^= r >> 32; }
  return r;
}
uint64_t index_checksum(const PackedRef& r) {
  uint64_t c = hash_bytes(r.contigs.data(), r.contigs.size() * sizeof(ContigInfo), 1);
  c = hash_bytes(r.codes.data(), r.codes.size() * 4, c);
  c = hash_bytes(r.mask.data(), r.mask.size() * 4, c);
  c = hash_bytes(r.runs.data(), r.runs.size() * sizeof(Run), c);
  c = hash_bytes(r.tiles.data(), r.tiles.size() * sizeof(TileInfo), c);
  return hash_bytes(r.masked_tiles.data(), r.masked_tiles.size() * 4, c);
}

// Everything the kernels index with goes to the device unchecked, so a loaded index is checked here: "" or what is wrong.
std::string validate_index(const PackedRef& r) {
  if (!(r.chunk == 64 || r.chunk == 128 || r.chunk == 256 || r.chunk == 512) || r.tile != (uint64_t)r.chunk * LANES_PER_TILE) return "tile geometry";
  const uint64_t T = r.tile, n_tiles = r.total_packed / T;
  if (r.contigs.empty() || r.total_packed / 16 > 0xFFFFFFFFull || n_tiles < 3) return "size";
  if (r.codes.size() != r.total_packed / 16 || r.mask.size() != r.total_packed / 32 || r.tiles.size() != n_tiles) return "array sizes";
  uint64_t next = T, sum = 0;            // tile 0 is padding
  std::vector<uint32_t> owner(n_tiles, 0xFFFFFFFFu);
  for (size_t c = 0; c < r.contigs.size(); c++) {
    const ContigInfo& ci = r.contigs[c];
    if (ci.gbase % T != 0 || ci.gbase < next || ci.len > 0x7FFFFFFFull) return "contig placement";
    uint64_t nt = (ci.len + (uint64_t)r.chunk + T - 1) / T;   // the contig and >= one chunk of padding
    if (nt == 0) nt = 1;
    if (ci.gbase / T + nt + 1 > n_tiles) return "contig beyond the packed space";
    for (uint64_t t = 0; t < nt; t++) owner[ci.gbase / T + t] = (uint32_t)c;
    next = ci.gbase + nt * T;
    sum += ci.len;
  }
  if (sum != r.total_bases) return "total_bases";
  for (uint64_t t = 0; t < n_tiles; t++)
    if (r.tiles[t].contig != owner[t] || r.tiles[t].flag > 2u) return "tile table";
  uint64_t prev_end = 0;
  for (const Run& u : r.runs) {
    if (u.len == 0 || u.start < prev_end || u.start + u.len > r.total_packed) return "run table";
    prev_end = u.start + u.len;
  }
  uint64_t prev_tile = 0;
  for (size_t i = 0; i < r.masked_tiles.size(); i++) {
    const uint32_t t = r.masked_tiles[i];
    if (t >= n_tiles || (i && t <= prev_tile) || r.tiles[t].flag != 1u) return "masked tile list";
    prev_tile = t;
  }
  return "";
}
template <typename T> bool wr(FILE* f, const T* p, size_t n) { return n == 0 || std::fwrite(p, sizeof(T), n, f) == n; }
template <typename T> bool rd(FILE* f, T* p, size_t n) { return n == 0 || std::fread(p, sizeof(T), n, f) == n; }
}  // namespace
extern "C" {

int calitas_save_index(const calitas_ctx* ctx, const char* path) {
  if (!ctx || !path) return CALITAS_EINVAL;
  calitas_ctx* c = const_cast<calitas_ctx*>(ctx);
  if (!ctx->has_ref) return fail(c, CALITAS_ESTATE, "calitas_set_reference has not been called");
  const PackedRef& r = ctx->ref;
  if (!r.absent.empty()) return fail(c, CALITAS_EINVAL, "a reference with absent contigs (one process's share of a multi-GPU job) is not saved as an index");
  std::string names;
  for (auto& n : r.names) { names += n; names += '\n'; }
  IndexHeader h{};
  std::memcpy(h.magic, "CALIDX02", 8);
  h.layout = kIndexLayout; h.checksum = index_checksum(r);
  h.chunk = (uint32_t)r.chunk; h.n_contigs = (uint32_t)r.contigs.size(); h.tile = r.tile; h.total_packed = r.total_packed;
  h.total_bases = r.total_bases; h.n_runs = r.runs.size(); h.n_tiles = r.tiles.size(); h.n_masked = r.masked_tiles.size();
  h.build_len = r.genome_build.size(); h.names_len = names.size();
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(c, CALITAS_EIO, std::string("cannot write ") + path);
  bool ok = wr(f, &h, 1) && wr(f, r.genome_build.data(), r.genome_build.size()) && wr(f, names.data(), names.size()) &&
            wr(f, r.contigs.data(), r.contigs.size()) && wr(f, r.codes.data(), r.codes.size()) && wr(f, r.mask.data(), r.mask.size()) &&
            wr(f, r.runs.data(), r.runs.size()) && wr(f, r.tiles.data(), r.tiles.size()) && wr(f, r.masked_tiles.data(), r.masked_tiles.size());
  ok = (std::fclose(f) == 0) && ok;
  return ok ? CALITAS_OK : fail(c, CALITAS_EIO, std::string("short write to ") + path);
}

int calitas_load_index(calitas_ctx* ctx, const char* path) {
  if (!ctx || !path) return CALITAS_EINVAL;
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(ctx, CALITAS_EIO, std::string("cannot read ") + path);
  IndexHeader h{};
  PackedRef r;
  std::string names;
  bool ok = rd(f, &h, 1) && std::memcmp(h.magic, "CALIDX02", 8) == 0 && h.layout == kIndexLayout && h.tile == (uint64_t)h.chunk * LANES_PER_TILE &&
            h.total_packed % std::max<uint64_t>(1, h.tile) == 0 && h.n_tiles == h.total_packed / std::max<uint64_t>(1, h.tile) &&
            h.build_len < (1u << 16) && h.names_len < (1ull << 32);
  if (ok) {   // the counts size the allocations below: they must add up to the file's own length
    const uint64_t want = sizeof(IndexHeader) + h.build_len + h.names_len + (uint64_t)h.n_contigs * sizeof(ContigInfo) + h.total_packed / 16 * 4 +
                          h.total_packed / 32 * 4 + h.n_runs * sizeof(Run) + h.n_tiles * sizeof(TileInfo) + h.n_masked * 4;
    ok = h.n_runs < (1ull << 40) && h.n_masked <= h.n_tiles && std::fseek(f, 0, SEEK_END) == 0 && (uint64_t)std::ftell(f) == want &&
         std::fseek(f, (long)sizeof(IndexHeader), SEEK_SET) == 0;
  }
  if (ok) {
    r.chunk = (int)h.chunk; r.tile = h.tile; r.total_packed = h.total_packed; r.total_bases = h.total_bases;
    r.genome_build.resize(h.build_len); names.resize(h.names_len);
    r.contigs.resize(h.n_contigs); r.codes.resize(h.total_packed / 16); r.mask.resize(h.total_packed / 32);
    r.runs.resize(h.n_runs); r.tiles.resize(h.n_tiles); r.masked_tiles.resize(h.n_masked);
    ok = rd(f, &r.genome_build[0], h.build_len) && rd(f, &names[0], h.names_len) && rd(f, r.contigs.data(), r.contigs.size()) &&
         rd(f, r.codes.data(), r.codes.size()) && rd(f, r.mask.data(), r.mask.size()) && rd(f, r.runs.data(), r.runs.size()) &&
         rd(f, r.tiles.data(), r.tiles.size()) && rd(f, r.masked_tiles.data(), r.masked_tiles.size());
  }
  std::fclose(f);
  if (ok) {
    size_t a = 0;
    while (a < names.size()) { size_t b = names.find('\n', a); if (b == std::string::npos) break; r.names.push_back(names.substr(a, b - a)); a = b + 1; }
    ok = r.names.size() == r.contigs.size();
  }
  if (ok && index_checksum(r) != h.checksum) return fail(ctx, CALITAS_EIO, std::string("calitas index fails its checksum: ") + path);
  if (!ok) return fail(ctx, CALITAS_EIO, std::string("not a calitas index (or truncated): ") + path);
  const std::string bad = validate_index(r);
  if (!bad.empty()) return fail(ctx, CALITAS_EIO, std::string("corrupt calitas index (") + bad + "): " + path);
  ctx->ref = std::move(r);
  return upload_reference(ctx);
}

int calitas_reference_info(const calitas_ctx* ctx, int32_t* n_contigs, uint64_t* total_bases, uint64_t* packed_bytes) {
  if (!ctx || !ctx->has_ref) return CALITAS_ESTATE;
  if (n_contigs) *n_contigs = (int32_t)ctx->ref.contigs.size();
  if (total_bases) *total_bases = ctx->ref.total_bases;
  if (packed_bytes) {                                          // 2 bits per base of the contigs that are resident here (all of them, usually)
    uint64_t bases = 0;
    for (size_t i = 0; i < ctx->ref.contigs.size(); i++) if (!ctx->ref.is_absent(i)) bases += ctx->ref.contigs[i].len;
    *packed_bytes = (bases + 3) / 4;
  }
  return CALITAS_OK;
}

const char* calitas_genome_build(const calitas_ctx* ctx) { return (ctx && ctx->has_ref) ? ctx->ref.genome_build.c_str() : ""; }

int calitas_contig_name(const calitas_ctx* ctx, int32_t i, const char** name, uint64_t* length) {
  if (!ctx || !ctx->has_ref || i < 0 || i >= (int32_t)ctx->ref.contigs.size()) return CALITAS_EINVAL;
  if (name) *name = ctx->ref.names[i].c_str();
  if (length) *length = ctx->ref.contigs[i].len;
  return CALITAS_OK;
}

int calitas_expand_rows(calitas_ctx* ctx, const char* compact, uint64_t n, uint64_t rows, const char* head, const char* tail, char* out,
                        uint64_t out_capacity, uint64_t* written) {
  if (!ctx) return CALITAS_EINVAL;
  if ((!compact && n) || !head || !tail || !out || !written) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  const std::string h(head), t(tail);
  if (t.empty() || t.back() != '\n') return fail(ctx, CALITAS_EINVAL, "the tail of a row ends with a newline");
  if (out_capacity < n + rows * (uint64_t)(h.size() + t.size() - 1)) return fail(ctx, CALITAS_EINVAL, "out is too small");
  const size_t w = expand_rows(compact, (size_t)n, rows, h, t, out, ctx->pool);
  if (w == (size_t)-1) return fail(ctx, CALITAS_EINVAL, "the compact text does not hold exactly the given number of newline-terminated rows");
  *written = w;
  return CALITAS_OK;
}

int calitas_fetch_bases(const calitas_ctx* ctx, int32_t i, uint64_t start, uint32_t len, char* out) {
  if (!ctx || !ctx->has_ref || i < 0 || i >= (int32_t)ctx->ref.contigs.size() || !out) return CALITAS_EINVAL;
  const ContigInfo& c = ctx->ref.contigs[i];
  if (start + len > c.len) return CALITAS_EINVAL;
  for (uint32_t k = 0; k < len; k++) out[k] = ctx->ref.base_upper(c.gbase + start + k);
  return CALITAS_OK;
}

int calitas_window_table(const calitas_ctx* ctx, int32_t window_size, int32_t step, int32_t min_length, int32_t chrom_index,
                         int32_t** out, uint64_t* n_windows) {
  if (!ctx || !ctx->has_ref || !out || !n_windows || step <= 0 || window_size <= 0) return CALITAS_EINVAL;
  const PackedRef& r = ctx->ref;
  std::vector<int32_t> rows;
  for (size_t c = 0; c < r.contigs.size(); c++) {
    if (chrom_index >= 0 && (int)c != chrom_index) continue;
    uint64_t nw = window_count(r.contigs[c].len, step);
    for (uint64_t k = 0; k < nw; k++) {
      int64_t a, b;
      if (!window_bounds(r.runs.data(), (int64_t)r.runs.size(), r.contigs[c].gbase, r.contigs[c].len, window_size, step, k, a, b)) continue;
      if (b - a < min_length) continue;
      rows.push_back((int32_t)c); rows.push_back((int32_t)a); rows.push_back((int32_t)b);
    }
  }
  *n_windows = rows.size() / 3;
  *out = (int32_t*)out_alloc(std::max<size_t>(1, rows.size()) * sizeof(int32_t));
  if (!rows.empty()) std::memcpy(*out, rows.data(), rows.size() * sizeof(int32_t));
  return CALITAS_OK;
}

}  // extern "C"

extern "C" {

int calitas_search(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                   calitas_aln_t** out, uint64_t* n_out) {
  return calitas_search_impl(ctx, n_guides, guides, params, out, n_out);
}

int calitas_search_hits(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                        const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows) {
  if (!ctx) return CALITAS_EINVAL;
  if (!guide || !params || !tsv) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  return calitas_search_hits_impl(ctx, guide, guide_id ? guide_id : "", params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows);
}

int calitas_search_hits_into(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                             const char* aligner_version, const char* time_stamp, char* dst, uint64_t dst_capacity, uint64_t* tsv_bytes,
                             uint64_t* n_rows) {
  if (!ctx) return CALITAS_EINVAL;
  if (!guide || !params || !dst) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  if (tsv_bytes) *tsv_bytes = 0;
  if (n_rows) *n_rows = 0;
  return calitas_search_hits_into_impl(ctx, guide, guide_id ? guide_id : "", params, aligner_version, time_stamp, dst, dst_capacity, tsv_bytes, n_rows);
}

int calitas_pin_host(calitas_ctx* ctx, void* p, uint64_t bytes) {
  if (!ctx || !p || !bytes) return CALITAS_EINVAL;
  if (ctx->device < 0) return fail(ctx, CALITAS_ENODEV, "host-only context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault));
  return CALITAS_OK;
}

int calitas_unpin_host(calitas_ctx* ctx, void* p) {
  if (!ctx || !p) return CALITAS_EINVAL;
  HIP_TRY(ctx, hipHostUnregister(p));
  return CALITAS_OK;
}

int calitas_search_hits_stream(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                               const char* aligner_version, const char* time_stamp, calitas_text_sink_t sink, void* user,
                               uint64_t* tsv_bytes, uint64_t* n_rows) {
  if (!ctx) return CALITAS_EINVAL;
  if (!guide || !params || !sink) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  if (tsv_bytes) *tsv_bytes = 0;
  if (n_rows) *n_rows = 0;
  if (params->first_window != 0 || params->n_windows != 0)
    return fail(ctx, CALITAS_EINVAL, "a window range (first_window / n_windows) is for calitas_search, calitas_search_hits(_into) and calitas_search_hits_batch");
  return calitas_search_hits_stream_impl(ctx, guide, guide_id ? guide_id : "", params, aligner_version, time_stamp, sink, user, tsv_bytes, n_rows);
}

int calitas_search_hits_batch(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const char* const* guide_ids,
                              const calitas_params_t* params, const char* aligner_version, const char* time_stamp, char** tsv,
                              uint64_t* tsv_bytes, uint64_t* n_rows) {
  if (!ctx) return CALITAS_EINVAL;
  if (n_guides <= 0 || !guides || !params || !tsv) return fail(ctx, CALITAS_EINVAL, "bad argument");
  return calitas_search_hits_batch_impl(ctx, n_guides, guides, guide_ids, params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows);
}

int calitas_scan_candidates(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                            uint32_t** records, uint64_t* n_records) {
  return calitas_scan_candidates_impl(ctx, n_guides, guides, params, records, n_records);
}

int calitas_scan_candidates_columnwise(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                                       uint32_t** records, uint64_t* n_records) {
  return calitas_scan_candidates_impl(ctx, n_guides, guides, params, records, n_records, true);
}

int calitas_reference_tiles(const calitas_ctx* ctx, uint64_t* n_tiles, uint64_t* n_dead, uint64_t* n_masked, uint64_t* tile_bases) {
  if (!ctx || !ctx->has_ref) return CALITAS_ESTATE;
  uint64_t n = 0, dead = 0, masked = 0;
  for (const TileInfo& t : ctx->ref.tiles) {
    if (t.contig == 0xFFFFFFFFu) continue;
    n++;
    if (t.flag == 2u) dead++; else if (t.flag == 1u) masked++;
  }
  if (n_tiles) *n_tiles = n;
  if (n_dead) *n_dead = dead;
  if (n_masked) *n_masked = masked;
  if (tile_bases) *tile_bases = ctx->ref.tile;
  return CALITAS_OK;
}

int calitas_contig_packed_base(const calitas_ctx* ctx, int32_t i, uint64_t* gbase) {
  if (!ctx || !ctx->has_ref || i < 0 || i >= (int32_t)ctx->ref.contigs.size() || !gbase) return CALITAS_EINVAL;
  *gbase = ctx->ref.contigs[i].gbase;
  return CALITAS_OK;
}

int calitas_get_timing(const calitas_ctx* ctx, calitas_timing_t* out) {
  if (!ctx || !out) return CALITAS_EINVAL;
  *out = ctx->timing;
  return CALITAS_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// host-side stages
// -------------------------------------------------------------------------------------------------------------------

int calitas_window_filter(const calitas_aln_t* alns, int32_t n, int32_t max_total_diffs, int32_t max_overlap, int32_t* order,
                          int32_t* n_kept) {
  if (!alns || n < 0 || !order || !n_kept) return CALITAS_EINVAL;
  // forward-strand list first, then reverse-strand list, each in the order given
  std::vector<calitas_aln_t> v;
  std::vector<int> src;
  for (int s = 0; s < 2; s++)
    for (int i = 0; i < n; i++) if ((alns[i].strand == '-') == (s == 1)) { v.push_back(alns[i]); src.push_back(i); }
  std::vector<int> kept;
  window_filter(v.data(), (int)v.size(), max_total_diffs, max_overlap, kept);
  for (size_t i = 0; i < kept.size(); i++) order[i] = src[kept[i]];
  *n_kept = (int32_t)kept.size();
  return CALITAS_OK;
}

int calitas_hits_tsv(const calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                     const calitas_aln_t* alns, uint64_t n_alns, const char* aligner_version, const char* time_stamp, char** tsv,
                     uint64_t* n_rows) {
  return calitas_hits_tsv_ext(ctx, guide, guide_id, params, alns, n_alns, nullptr, 0, aligner_version, time_stamp, tsv, n_rows);
}

int calitas_hits_tsv_ext(const calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                         const calitas_aln_t* alns, uint64_t n_alns, const calitas_ext_hit_t* ext, uint64_t n_ext,
                         const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* n_rows) {
  if (n_ext && !ext) return CALITAS_EINVAL;
  if (!ctx || !guide || !params || !tsv || (n_alns && !alns)) return CALITAS_EINVAL;
  calitas_ctx* c = const_cast<calitas_ctx*>(ctx);
  if (!ctx->has_ref) return fail(c, CALITAS_ESTATE, "calitas_set_reference has not been called");
  GuideHost gh;
  std::string e = make_guide_host(*guide, gh);
  if (!e.empty()) return fail(c, CALITAS_EINVAL, e);
  std::string version, stamp;
  calitas_default_version_and_stamp(aligner_version, time_stamp, version, stamp);
  *tsv = hits_tsv(ctx->ref, gh, guide_id ? guide_id : "", *params, alns, n_alns, version, stamp, n_rows, ctx->pool, out_alloc, ext, n_ext);
  if (!*tsv) return fail(c, CALITAS_EINVAL, "out of memory");
  return CALITAS_OK;
}

int calitas_padded_strings(const calitas_ctx* ctx, const calitas_guide_t* guide, const calitas_aln_t* aln, char* padded_guide,
                           char* padded_alignment, char* padded_target) {
  if (!ctx || !ctx->has_ref || !guide || !aln || !padded_guide || !padded_alignment || !padded_target) return CALITAS_EINVAL;
  GuideHost gh;
  std::string e = make_guide_host(*guide, gh);
  if (!e.empty()) return fail(const_cast<calitas_ctx*>(ctx), CALITAS_EINVAL, e);
  std::string pg, pa, pt;
  padded_strings(ctx->ref, gh, *aln, pg, pa, pt);
  std::strcpy(padded_guide, pg.c_str()); std::strcpy(padded_alignment, pa.c_str()); std::strcpy(padded_target, pt.c_str());
  return CALITAS_OK;
}

}  // extern "C"
