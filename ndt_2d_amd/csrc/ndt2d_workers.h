// One persistent host thread per device of a multi-device matcher beyond the first
// (ndt2d_matcher_create_multi): the calling thread drives device 0, thread r drives device r,
// so that the devices' uploads and launches go out side by side instead of one after the
// other (a launch costs the host ~5 us, an upload + launch ~20 us: eight devices dealt by one
// thread started 140 us apart).  Host C++ only, no HIP types.
//
// run(fn) calls fn(rank) for every rank -- rank 0 on the caller -- and returns when all have
// returned.  A thread that has just worked spins for kSpinNs waiting for the next call (a
// filter or a mapper calls at a steady rate; a wake-up through the kernel costs 30-60 us),
// then parks on a condition variable.  barrier(): all ranks meet inside a run() (the particle
// path's sums meet on the host between a device's two launches).
#ifndef NDT2D_WORKERS_H_
#define NDT2D_WORKERS_H_

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

namespace ndt2d
{

class DeviceWorkers
{
public:
  explicit DeviceWorkers(size_t n_ranks) : n_(n_ranks)
  {
    for (size_t r = 1; r < n_; ++r) threads_.emplace_back([this, r] { loop(r); });
  }

  ~DeviceWorkers()
  {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
      gen_.fetch_add(1);
    }
    cv_go_.notify_all();
    for (std::thread & t : threads_) t.join();
  }

  DeviceWorkers(const DeviceWorkers &) = delete;
  DeviceWorkers & operator=(const DeviceWorkers &) = delete;

  size_t ranks() const { return n_; }

  // fn(rank) on every rank; fn must not throw.  Not reentrant (one call at a time: a matcher
  // is driven by one thread).
  template <class F>
  void run(F & fn)
  {
    if (n_ <= 1)
    {
      fn(static_cast<size_t>(0));
      return;
    }
    call_ = [](void * ctx, size_t rank) { (*static_cast<F *>(ctx))(rank); };
    ctx_ = &fn;
    pending_.store(static_cast<uint32_t>(n_ - 1));
    arrived_.store(0);
    {
      // (the generation changes under the mutex: a thread between its last look at it and its
      // wait cannot miss the change)
      std::lock_guard<std::mutex> lock(mu_);
      gen_.fetch_add(1);
    }
    cv_go_.notify_all();
    fn(static_cast<size_t>(0));
    // the others: they took the same path, normally they are done or nearly so
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins)
    {
      pause();
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_done_.wait_for(lock, std::chrono::milliseconds(1),
                          [this] { return pending_.load(std::memory_order_acquire) == 0; });
      }
    }
  }

  // All ranks of the current run() meet here (each calls it once per barrier, in the same
  // order).  false: `give_up` went up (a rank failed and will not come) or the wait exceeded
  // `timeout_ms` -- the caller then abandons its part.
  bool barrier(const std::atomic<bool> & give_up, int timeout_ms = 10000)
  {
    if (n_ <= 1) return !give_up.load();
    const uint32_t ticket = arrived_.fetch_add(1, std::memory_order_acq_rel) + 1;
    const uint32_t target = static_cast<uint32_t>(((ticket - 1) / n_ + 1) * n_);   // end of this round
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; arrived_.load(std::memory_order_acquire) < target; ++spins)
    {
      if (give_up.load(std::memory_order_acquire)) return false;
      pause();
      if ((spins & 4095u) == 4095u &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms))
      {
        return false;
      }
    }
    return !give_up.load(std::memory_order_acquire);
  }

private:
  static void pause()
  {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }

  void loop(size_t rank)
  {
    uint32_t seen = 0;
    for (;;)
    {
      // the next call: spin while the last one is recent, then park
      const auto t0 = std::chrono::steady_clock::now();
      uint32_t g = gen_.load(std::memory_order_acquire);
      for (uint32_t spins = 0; g == seen; ++spins)
      {
        pause();
        if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::nanoseconds(kSpinNs))
        {
          std::unique_lock<std::mutex> lock(mu_);
          cv_go_.wait(lock, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        }
        g = gen_.load(std::memory_order_acquire);
      }
      seen = g;
      if (stop_) return;
      call_(ctx_, rank);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1)
      {
        std::lock_guard<std::mutex> lock(mu_);
        cv_done_.notify_one();
      }
    }
  }

  static constexpr int64_t kSpinNs = 200000;
  const size_t n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_go_, cv_done_;
  std::atomic<uint32_t> gen_{0};
  std::atomic<uint32_t> pending_{0};
  std::atomic<uint32_t> arrived_{0};
  bool stop_ = false;
  void (*call_)(void *, size_t) = nullptr;
  void * ctx_ = nullptr;
};

}  // namespace ndt2d

#endif  // NDT2D_WORKERS_H_
