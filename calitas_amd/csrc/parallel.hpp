// parallel.hpp -- a small persistent worker pool for the host-side stages (per-window filter, hit rows).
// The reference spreads windows over a ThreadPoolExecutor (SearchReference.scala:75-94); here the GPU does the
// alignment work and the pool only covers the residual host stages, which are independent per window / per contig.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>
#include <time.h>

#include <chrono>

#include "tuning.hpp"

namespace calitas {

// One step of a host wait on the critical path: pure spinning for the first ~50 us (a round trip through the mailbox is 10-30 us and a
// blocking wait would add its wake-up latency to each of them), then the core is offered to whoever else can run on it (eight ranks
// of a job, each with a caller and a lane thread, may share sixteen cores), and from 5 ms on -- a long kernel, a 20 GB copy -- the
// thread sleeps between looks.
struct Backoff {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  long long waited_us() const { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
  void pause() {
    if (++spins < 256) { __builtin_ia32_pause(); return; }   // (the first ~10 us: not even a look at the clock)
    const long long us = waited_us();
    if (us < 50) __builtin_ia32_pause();
    else if (us < 5000) sched_yield();
    else { timespec ts{0, 50000}; nanosleep(&ts, nullptr); }
  }
};

// A job whose pieces are claimed by whoever shows up (WorkerPool::offer): work() takes pieces until none is left and returns.
struct SharedJob {
  virtual ~SharedJob() {}
  virtual void work() = 0;
  virtual bool exhausted() const { return false; }              // nothing left to take: the pool forgets the job
};

class WorkerPool {
 public:
  explicit WorkerPool(int n_threads) {
    n_ = std::max(1, n_threads);
    for (int i = 1; i < n_; i++) threads_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      wake_seq_++;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int size() const { return n_; }

  // Runs fn(0) ... fn(size() - 1), each once, on whichever threads are free -- the caller among them -- and returns when all have run.
  // The argument is the share's number, not a thread's: what fn(t) writes is share t's, and two shares may run on one thread one
  // after the other.  Callers on different threads run side by side (the variant branch builds windows, lifts alignments and makes
  // rows on three threads while the reference passes of the same call expand their text): their shares are taken in the order the
  // calls came.  Nobody waits for a worker that has not arrived (a GPU box bounds this process by a CPU quota on a host shared with
  // other tenants: one woken thread in a few hundred starts 4-8 ms late, profiles/r04_slow_calls.txt) -- only for shares in hand.
  void run(const std::function<void(int)>& fn) {
    if (n_ == 1) { fn(0); return; }
    auto job = std::make_shared<Shares>();
    job->fn = &fn; job->total = n_;
    {
      std::lock_guard<std::mutex> lk(m_);
      active_.push_back(job);
      wake_seq_++;
    }
    cv_.notify_all();
    take_shares(*job);
    Backoff wait;
    while (job->done.load(std::memory_order_acquire) < job->total) wait.pause();   // (shares in hand: their holders are running)
    std::lock_guard<std::mutex> lk(m_);
    active_.erase(std::find(active_.begin(), active_.end(), job));
  }

  // Offers a job to the workers that are free and returns at once: each of them calls job->work() once, when it gets to it.  The
  // caller works on the job itself and decides when it is complete.  The job is kept alive by the shared_ptr until the last late
  // worker has looked at it.  Several offers can be live at once (the lanes of a chunked or batch call expand their texts side by
  // side): every worker visits each of them once, oldest first; an offer is forgotten when it says it is exhausted, and the oldest
  // one when more than kMaxOffers are live (its caller finishes it alone, as it would with no worker free).
  void offer(const std::shared_ptr<SharedJob>& job) {
    if (n_ == 1) return;
    {
      std::lock_guard<std::mutex> lk(m_);
      offers_.erase(std::remove_if(offers_.begin(), offers_.end(), [](const Offer& o) { return o.job->exhausted(); }), offers_.end());
      if (offers_.size() >= kMaxOffers) offers_.erase(offers_.begin());
      offers_.push_back(Offer{++offer_seq_, job});
      wake_seq_++;
    }
    cv_.notify_all();
  }

  // Static block partition of [0, n) into size() consecutive blocks: body(begin, end, block number).
  void for_blocks(size_t n, const std::function<void(size_t, size_t, int)>& body) {
    run([&](int tid) {
      size_t per = (n + n_ - 1) / n_;
      size_t b = std::min(n, per * tid), e = std::min(n, b + per);
      if (b < e) body(b, e, tid);
    });
  }

  static int default_threads() {
    if (const char* e = TUNE_GET("CALITAS_THREADS")) { int v = std::atoi(e); if (v > 0) return std::min(v, 256); }
    unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));  // the GPU boxes give one GPU a 16-core share
  }

 private:
  struct Shares {                                               // one run(): its shares are numbered 0 .. total - 1
    const std::function<void(int)>* fn = nullptr;               // (the caller's: alive until every share has run)
    int total = 0;
    std::atomic<int> next{0}, done{0};
  };
  static void take_shares(Shares& j) {
    for (;;) {
      const int t = j.next.fetch_add(1, std::memory_order_relaxed);
      if (t >= j.total) return;
      (*j.fn)(t);
      j.done.fetch_add(1, std::memory_order_release);
    }
  }
  void loop() {
    unsigned long seen = 0, seen_offer = 0;
    for (;;) {
      std::shared_ptr<Shares> shares;
      std::shared_ptr<SharedJob> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
          if (stop_) return;
          for (auto& a : active_) if (a->next.load(std::memory_order_relaxed) < a->total) { shares = a; break; }
          if (shares) break;
          for (auto it = offers_.begin(); it != offers_.end();) {
            if (it->job->exhausted()) { it = offers_.erase(it); continue; }
            if (it->seq > seen_offer) { seen_offer = it->seq; job = it->job; break; }
            ++it;
          }
          if (job) break;
          seen = wake_seq_;
          cv_.wait(lk, [&] { return wake_seq_ != seen; });
        }
      }
      if (shares) take_shares(*shares);
      else if (job) job->work();
    }
  }
  int n_ = 1;
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::shared_ptr<Shares>> active_;                 // the calls of run() under way, oldest first
  struct Offer { unsigned long seq; std::shared_ptr<SharedJob> job; };
  static constexpr size_t kMaxOffers = 16;
  std::vector<Offer> offers_;                                   // the live offers, oldest first
  unsigned long offer_seq_ = 0;
  unsigned long wake_seq_ = 0;
  bool stop_ = false;
};

}  // namespace calitas
