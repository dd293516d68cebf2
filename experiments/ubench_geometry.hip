// Launch -> first result on the host for an (almost) empty kernel of various launch geometries:
// does the default search's geometry (240 blocks x 960 threads, 37 KB of dynamic LDS, 3.7 KB of
// kernel arguments) cost start-up time by itself?
//   hipcc --offload-arch=gfx950 -O2 -o experiments/bin/ubench_geometry experiments/ubench_geometry.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { double v[416]; };
__global__ void k_geom(unsigned long long * host_flag, unsigned long long seq, int use_lds)
{
  extern __shared__ double lds[];
  if (use_lds && threadIdx.x < 8) lds[threadIdx.x] = (double)seq;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    __hip_atomic_store(host_flag, seq + (use_lds && lds[3] < 0 ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void k_geom_args(unsigned long long * host_flag, unsigned long long seq, Big big)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    __hip_atomic_store(host_flag, seq + (big.v[(int)(seq & 255)] < -1e300 ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
typedef std::chrono::steady_clock clk;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main()
{
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long * flag, * d_flag;
  CK(hipHostMalloc((void **)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&d_flag, flag, 0));
  CK(hipFuncSetAttribute((const void *)k_geom, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  unsigned long long seq = 0;
  Big big = {};
  struct Case { int blocks, threads, lds; const char * name; };
  const Case cases[] = {{1, 64, 0, "1 x 64"}, {240, 64, 0, "240 x 64"}, {240, 960, 0, "240 x 960"},
                        {240, 960, 37 * 1024, "240 x 960, 37 KB LDS"}, {240, 960, 74 * 1024, "240 x 960, 74 KB LDS"},
                        {2048, 256, 0, "2048 x 256"}, {-1, 0, 0, "240 x 960, 3.3 KB of arguments"}};
  for (const Case & c : cases)
  {
    std::vector<double> t;
    for (int rep = 0; rep < 3050; ++rep)
    {
      ++seq;
      auto a = clk::now();
      if (c.blocks < 0) hipLaunchKernelGGL(k_geom_args, dim3(240), dim3(960), 0, s, d_flag, seq, big);
      else hipLaunchKernelGGL(k_geom, dim3(c.blocks), dim3(c.threads), c.lds, s, d_flag, seq, c.lds > 0 ? 1 : 0);
      volatile unsigned long long * f = flag;
      while (*f != seq) __builtin_ia32_pause();
      auto e = clk::now();
      CK(hipStreamSynchronize(s));
      if (rep >= 50) t.push_back(us(a, e));
    }
    std::sort(t.begin(), t.end());
    printf("%-34s launch -> flag on the host: median %.2f us  p99 %.2f us\n", c.name, t[t.size() / 2], t[t.size() * 99 / 100]);
  }
  return 0;
}
