// ndt2d::DeviceWorkers (ndt_2d_amd/csrc/ndt2d_workers.h) without a GPU: every rank runs once per
// run(), rank 0 on the caller; the barrier lets the ranks meet (several times per run); a rank
// that gives up releases the others; thousands of back-to-back runs; parked threads wake up.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "ndt2d_workers.h"

int main()
{
  const size_t n = 8;
  ndt2d::DeviceWorkers pool(n);
  const std::thread::id caller = std::this_thread::get_id();
  std::vector<int> hits(n, 0);
  bool rank0_on_caller = true;
  for (int rep = 0; rep < 5000; ++rep)
  {
    auto fn = [&](size_t r) {
      ++hits[r];
      if (r == 0 && std::this_thread::get_id() != caller) rank0_on_caller = false;
      if (r != 0 && std::this_thread::get_id() == caller) rank0_on_caller = false;
    };
    pool.run(fn);
  }
  for (size_t r = 0; r < n; ++r)
  {
    if (hits[r] != 5000) { std::printf("rank %zu ran %d times\n", r, hits[r]); return 1; }
  }
  if (!rank0_on_caller) { std::printf("rank 0 did not run on the caller\n"); return 2; }

  // the rows meet: every rank publishes a value, all see the same total after the barrier -- twice
  for (int rep = 0; rep < 2000; ++rep)
  {
    std::vector<double> row(n, 0.0), total(n, 0.0), total2(n, 0.0);
    std::atomic<bool> give_up{false};
    std::atomic<int> met{0};
    auto fn = [&](size_t r) {
      row[r] = static_cast<double>(r + 1) * (rep + 1);
      if (!pool.barrier(give_up)) return;
      double acc = 0.0;
      for (size_t q = 0; q < n; ++q) acc += row[q];
      total[r] = acc;
      if (!pool.barrier(give_up)) return;   // nobody rewrites its row before everybody has read
      row[r] = -row[r];
      if (!pool.barrier(give_up)) return;
      acc = 0.0;
      for (size_t q = 0; q < n; ++q) acc += row[q];
      total2[r] = acc;
      ++met;
    };
    pool.run(fn);
    const double want = 36.0 * (rep + 1);
    for (size_t r = 0; r < n; ++r)
    {
      if (total[r] != want || total2[r] != -want || met.load() != static_cast<int>(n))
      {
        std::printf("rep %d rank %zu: %g %g (want %g), met %d\n", rep, r, total[r], total2[r], want, met.load());
        return 3;
      }
    }
  }

  // one rank fails before the barrier: the others are released, nobody hangs
  {
    std::atomic<bool> give_up{false};
    std::atomic<int> released{0};
    auto fn = [&](size_t r) {
      if (r == 5)
      {
        give_up.store(true);
        return;
      }
      if (!pool.barrier(give_up)) ++released;
    };
    const auto t0 = std::chrono::steady_clock::now();
    pool.run(fn);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (released.load() != static_cast<int>(n) - 1 || ms > 2000.0) { std::printf("give-up: %d released in %.1f ms\n", released.load(), ms); return 4; }
  }
  // ... and a rank that never comes: the barrier's own bound ends the wait
  {
    std::atomic<bool> give_up{false};
    std::atomic<int> timed_out{0};
    auto fn = [&](size_t r) {
      if (r == 2) return;
      if (!pool.barrier(give_up, 50)) ++timed_out;
    };
    pool.run(fn);
    if (timed_out.load() != static_cast<int>(n) - 1) { std::printf("timeout: %d\n", timed_out.load()); return 5; }
  }
  // parked threads (no call for longer than the spin time) wake up
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  {
    std::atomic<int> ran{0};
    auto fn = [&](size_t) { ++ran; };
    pool.run(fn);
    if (ran.load() != static_cast<int>(n)) return 6;
  }
  // a pool of one rank has no thread: run() is the call itself
  {
    ndt2d::DeviceWorkers one(1);
    int k = 0;
    auto fn = [&](size_t r) { k += static_cast<int>(r) + 1; };
    one.run(fn);
    std::atomic<bool> give_up{false};
    if (k != 1 || !one.barrier(give_up)) return 7;
  }
  std::printf("workers ok\n");
  return 0;
}
