// Does it pay to launch the consumer of a staged image BEFORE the host has produced the image --
// the kernel waits on a word in host memory -- instead of after?  (addScans: 24 us of host build
// during which the GPU idles, then an install kernel that is 8.5 us from launch to end.)
// Prints the time from "image ready" to "consumer's done flag seen by the host" for 16 / 90 KB:
//   launch after the image is ready  |  kernel launched 25 us earlier, waiting for the go word.
// Block 0 polls the host word (over PCIe) and passes it on through a device word the others poll.
//   hipcc --offload-arch=gfx950 -O2 experiments/ubench_prelaunch.hip -o experiments/bin/ubench_prelaunch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void consume(const double2 * src, size_t n16, double * out, volatile unsigned long long * done_flag,
                        unsigned long long seq, unsigned int * counter, const unsigned long long * go_host,
                        unsigned long long * go_dev)
{
  if (go_host != nullptr)
  {
    if (threadIdx.x == 0)
    {
      if (blockIdx.x == 0)
      {
        long spins = 0;
        while (__hip_atomic_load(go_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq && spins < (1l << 22))
        {
          __builtin_amdgcn_s_sleep(4);
          ++spins;
        }
        __hip_atomic_store(go_dev, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      else
      {
        while (__hip_atomic_load(go_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
  }
  double acc = 0.0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x)
  {
    const double2 v = src[i];
    acc += v.x + v.y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __threadfence();
    if (atomicAdd(counter, 1u) == gridDim.x - 1)
    {
      *counter = 0;
      __hip_atomic_store(const_cast<unsigned long long *>(done_flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main()
{
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t cap = 128 * 1024;
  double * pinned = nullptr, * pinned_dev = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void **>(&pinned), cap, hipHostMallocDefault));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&pinned_dev), pinned, 0));
  unsigned long long * words = nullptr, * words_dev = nullptr;   // [0] done flag, [8] go word
  CK(hipHostMalloc(reinterpret_cast<void **>(&words), 256, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&words_dev), words, 0));
  std::memset(words, 0, 256);
  double * out = nullptr; unsigned int * counter = nullptr; unsigned long long * go_dev = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&out), 8 * 64 * 256)); CK(hipMalloc(reinterpret_cast<void **>(&counter), 4));
  CK(hipMalloc(reinterpret_cast<void **>(&go_dev), 8));
  CK(hipMemset(counter, 0, 4)); CK(hipMemset(go_dev, 0, 8)); CK(hipDeviceSynchronize());
  std::vector<double> src(cap / 8, 1.5);
  unsigned long long seq = 0;
  for (size_t bytes : {size_t(16 * 1024), size_t(90 * 1024)})
  {
    for (int mode = 0; mode < 2; ++mode)
    {
      std::vector<double> t;
      for (int rep = -20; rep < 400; ++rep)
      {
        ++seq;
        const unsigned blocks = static_cast<unsigned>(std::min<size_t>((bytes / 16 + 255) / 256, 64));
        if (mode == 1)
        {
          hipLaunchKernelGGL(consume, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const double2 *>(pinned_dev),
                             bytes / 16, out, words_dev, seq, counter, words_dev + 8, go_dev);
        }
        const double w0 = now_us();
        while (now_us() - w0 < 25.0) {}                        // the host "builds"
        std::memcpy(pinned, src.data(), bytes);                // ... and stages
        const double ready = now_us();
        if (mode == 1)
        {
          __atomic_store_n(words + 8, seq, __ATOMIC_RELEASE);
        }
        else
        {
          hipLaunchKernelGGL(consume, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const double2 *>(pinned_dev),
                             bytes / 16, out, words_dev, seq, counter, static_cast<const unsigned long long *>(nullptr), go_dev);
        }
        while (*reinterpret_cast<volatile unsigned long long *>(words) != seq) {}
        if (rep >= 0) t.push_back(now_us() - ready);
      }
      std::sort(t.begin(), t.end());
      printf("%3zu KB, %s: image ready -> consumer done %.2f us (p90 %.2f)\n", bytes / 1024,
             mode == 0 ? "launched when the image is ready" : "launched 25 us earlier, waiting   ", t[t.size() / 2],
             t[t.size() * 9 / 10]);
    }
  }
  return 0;
}
