// parallel.hpp -- a small persistent worker pool for the host-side stages (per-window filter, hit rows).
// The reference spreads windows over a ThreadPoolExecutor (SearchReference.scala:75-94); here the GPU does the
// alignment work and the pool only covers the residual host stages, which are independent per window / per contig.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "tuning.hpp"

namespace calitas {

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
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_;
      }
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
  unsigned long gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};

}  // namespace calitas
