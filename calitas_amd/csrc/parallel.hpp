// parallel.hpp -- a small persistent worker pool for the host-side stages (per-window filter, hit rows).
// The reference spreads windows over a ThreadPoolExecutor (SearchReference.scala:75-94); here the GPU does the
// alignment work and the pool only covers the residual host stages, which are independent per window / per contig.
#pragma once
#include <algorithm>
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
};

class WorkerPool {
 public:
  explicit WorkerPool(int n_threads) {
    n_ = std::max(1, n_threads);
    for (int i = 1; i < n_; i++) threads_.emplace_back([this, i] { loop(i); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int size() const { return n_; }

  // Runs fn(tid) on every worker (tid 0 is the calling thread) and waits for all of them.
  // (Callers on different threads take turns: the variant branch builds windows and rows while the reference passes of the same
  // call copy their text, variants.cpp.)
  void run(const std::function<void(int)>& fn) {
    if (n_ == 1) { fn(0); return; }
    std::lock_guard<std::mutex> turn(run_mu_);
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      pending_ = n_ - 1;
      gen_++;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

  // Offers a job to the workers that are free and returns at once: each of them calls job->work() once, when it gets to it.  The
  // caller works on the job itself and decides when it is complete -- it does not wait for workers that have not arrived.  (A GPU
  // box shares its host with other tenants and bounds this process by a CPU quota, not a CPU set: one woken thread in a few hundred
  // lands behind somebody else's time slice and starts 4-8 ms late, profiles/r04_slow_calls.txt.  With run() the whole job waits
  // for it; here it finds the pieces gone.)  The job is kept alive by the shared_ptr until the last late worker has looked at it.
  void offer(const std::shared_ptr<SharedJob>& job) {
    if (n_ == 1) return;
    {
      std::lock_guard<std::mutex> lk(m_);
      offer_ = job;
      offer_seq_++;
    }
    cv_.notify_all();
  }

  // Static block partition of [0, n) over the workers.
  void for_blocks(size_t n, const std::function<void(size_t, size_t, int)>& body) {
    run([&](int tid) {
      size_t per = (n + n_ - 1) / n_;
      size_t b = std::min(n, per * tid), e = std::min(n, b + per);
      if (b < e) body(b, e, tid);
    });
  }

  static int default_threads() {
    if (const char* e = tune::get("CALITAS_THREADS")) { int v = std::atoi(e); if (v > 0) return std::min(v, 256); }
    unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));  // the GPU boxes give one GPU a 16-core share
  }

 private:
  void loop(int tid) {
    unsigned long seen = 0, seen_offer = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      std::shared_ptr<SharedJob> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen || offer_seq_ != seen_offer; });
        if (stop_) return;
        if (gen_ == seen) {                                     // an offered job: no one waits for this worker
          seen_offer = offer_seq_;
          job = offer_;
        } else {
          seen = gen_;
          fn = fn_;
        }
      }
      if (job) { job->work(); continue; }
      if (fn) (*fn)(tid);
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  int n_ = 1;
  std::vector<std::thread> threads_;
  std::mutex m_, run_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  std::shared_ptr<SharedJob> offer_;
  unsigned long offer_seq_ = 0;
  unsigned long gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};

}  // namespace calitas
