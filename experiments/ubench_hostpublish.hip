// What would it cost to let EVERY block of the default search publish its own record to
// host-coherent memory (the host adds them up) instead of the ticket + last-block reduction?
//   hipcc --offload-arch=gfx950 -O2 -o experiments/bin/ubench_hostpublish experiments/ubench_hostpublish.hip
// Launch -> the host has seen the flag(s), medians over 3000 repetitions:
//   one block, one record + flag (today's final publish)
//   N blocks (240, 560), each 12 doubles + its own flag, the host polling all N flags
//   N blocks spinning ~15 us first (as the search's blocks do), then publishing
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_publish(unsigned long long * host_flags, double * host_recs, unsigned long long seq, int spin)
{
  double v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 0.5;
  const unsigned b = blockIdx.x;
  if (threadIdx.x < 12)
  {
    __hip_atomic_store(host_recs + 16 * b + threadIdx.x, (double)seq + (v == 12345.0 ? 1.0 : 0.0), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(host_flags + 8 * b, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

typedef std::chrono::steady_clock clk;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main()
{
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long * flags;
  double * recs;
  const int kMax = 1024;
  CK(hipHostMalloc((void **)&flags, kMax * 64, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostMalloc((void **)&recs, kMax * 128, hipHostMallocCoherent | hipHostMallocMapped));
  unsigned long long * d_flags;
  double * d_recs;
  CK(hipHostGetDevicePointer((void **)&d_flags, flags, 0));
  CK(hipHostGetDevicePointer((void **)&d_recs, recs, 0));
  unsigned long long seq = 0;
  const int cases[][2] = {{1, 0}, {240, 0}, {560, 0}, {1, 30000}, {240, 30000}, {560, 30000}};
  for (auto & c : cases)
  {
    const int n = c[0], spin = c[1];
    std::vector<double> t;
    for (int rep = 0; rep < 3050; ++rep)
    {
      ++seq;
      auto a = clk::now();
      hipLaunchKernelGGL(k_publish, dim3(n), dim3(64), 0, s, d_flags, d_recs, seq, spin);
      double sum = 0.0;
      for (int b = 0; b < n; ++b)
      {
        volatile unsigned long long * f = flags + 8 * b;
        while (*f != seq) __builtin_ia32_pause();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int k = 0; k < 12; ++k) sum += recs[16 * b + k];
      }
      auto e = clk::now();
      if (sum < 0) printf("?");
      if (rep >= 50) t.push_back(us(a, e));
    }
    std::sort(t.begin(), t.end());
    printf("%4d blocks, spin %5d: launch -> all records on the host and added: median %.2f us  p99 %.2f us\n", n, spin,
           t[t.size() / 2], t[t.size() * 99 / 100]);
  }
  return 0;
}
