// Host-side latency of the primitives a small matchScan call is made of, one MI355X.
//   hipcc --offload-arch=gfx950 -O2 -o experiments/bin/ubench_latency experiments/ubench_latency.hip
// Prints medians in microseconds over 2000 repetitions each.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(double * out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0; }
__global__ void k_spin(double * out, int iters)
{
  double v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 0.5;
  if (v == 12345.0) out[1] = v;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0;
}
// result + flag into (pinned, coherent) host memory
__global__ void k_flag(volatile unsigned long long * host_flag, double * host_rec, unsigned long long seq)
{
  if (threadIdx.x < 12) host_rec[threadIdx.x] = (double)seq;
  __threadfence_system();
  if (threadIdx.x == 0) *host_flag = seq;
}

// the product's pattern: system-scope stores, wait for their acknowledgement, then the flag
__global__ void k_flag_ack(unsigned long long * host_flag, double * host_rec, unsigned long long seq)
{
  if (threadIdx.x < 12) __hip_atomic_store(host_rec + threadIdx.x, (double)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// record and sequence number in ONE store instruction (16 lanes x 8 B = two 64-byte lines,
// each carrying a copy of seq in its last word), nothing waited for: does the host ever see
// a line's seq without the line's data?
__global__ void k_flag_inline(unsigned long long * host_line, unsigned long long seq)
{
  if (threadIdx.x < 16)
  {
    const bool is_seq = (threadIdx.x & 7) == 7;
    const unsigned long long v = is_seq ? seq : (seq * 1000003ull + threadIdx.x);
    __hip_atomic_store(host_line + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

typedef std::chrono::steady_clock clk;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

template <class F>
double med(F f, int n = 2000)
{
  std::vector<double> t;
  for (int i = 0; i < n + 50; ++i)
  {
    auto a = clk::now();
    f();
    auto b = clk::now();
    if (i >= 50) t.push_back(us(a, b));
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main()
{
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double * d;
  CK(hipMalloc(&d, 1 << 20));
  double * pin;
  CK(hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault));
  double * pin_coh;
  CK(hipHostMalloc(&pin_coh, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  volatile unsigned long long * flag = reinterpret_cast<volatile unsigned long long *>(pin_coh + 64);
  double * d_pin_coh = nullptr;
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_pin_coh), pin_coh, 0));
  std::vector<double> pageable(1 << 17);
  hipEvent_t e0, e1, edone;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&edone, hipEventDisableTiming));
  unsigned long long seq = 0;

  printf("launch empty + streamSync            %7.2f us\n", med([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("launch empty x2 + streamSync         %7.2f us\n", med([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("launch empty x3 + streamSync         %7.2f us\n", med([&] { for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("launch 256 blocks empty + sync       %7.2f us\n", med([&] { hipLaunchKernelGGL(k_empty, 256, 1024, 0, s, d); hipStreamSynchronize(s); }));
  printf("ev0 + launch + ev1 + streamSync      %7.2f us\n", med([&] { hipEventRecord(e0, s); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipEventRecord(e1, s); hipStreamSynchronize(s); }));
  printf("launch + eventRecord + eventSync     %7.2f us\n", med([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipEventRecord(edone, s); hipEventSynchronize(edone); }));
  printf("H2D 4KB pinned + launch + sync       %7.2f us\n", med([&] { hipMemcpyAsync(d, pin, 4096, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("H2D 4KB pageable + launch + sync     %7.2f us\n", med([&] { hipMemcpyAsync(d, pageable.data(), 4096, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("H2D 80KB pinned + launch + sync      %7.2f us\n", med([&] { hipMemcpyAsync(d, pin, 81920, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("H2D 80KB pageable + launch + sync    %7.2f us\n", med([&] { hipMemcpyAsync(d, pageable.data(), 81920, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipStreamSynchronize(s); }));
  printf("launch + D2H 96B pageable + sync     %7.2f us\n", med([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipMemcpyAsync(pageable.data(), d, 96, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }));
  printf("launch + D2H 96B pinned + sync       %7.2f us\n", med([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s, d); hipMemcpyAsync(pin, d, 96, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }));
  printf("launch writes pinned-coherent + sync %7.2f us\n", med([&] { ++seq; hipLaunchKernelGGL(k_flag, 1, 64, 0, s, (volatile unsigned long long *)(d_pin_coh + 64), d_pin_coh, seq); hipStreamSynchronize(s); }));
  printf("launch writes pinned + spin on flag  %7.2f us\n", med([&] { ++seq; hipLaunchKernelGGL(k_flag, 1, 64, 0, s, (volatile unsigned long long *)(d_pin_coh + 64), d_pin_coh, seq); while (*flag != seq) { } }));
  {
    unsigned long long * uflag = reinterpret_cast<unsigned long long *>(d_pin_coh + 64);
    printf("launch stores+ack+flag, spin         %7.2f us\n", med([&] { ++seq; hipLaunchKernelGGL(k_flag_ack, 1, 64, 0, s, uflag, d_pin_coh, seq); while (*flag != seq) { } }));
    volatile unsigned long long * line = reinterpret_cast<volatile unsigned long long *>(pin_coh + 128);
    unsigned long long * d_line = reinterpret_cast<unsigned long long *>(d_pin_coh + 128);
    long torn = 0;
    printf("launch record+seq in one store, spin %7.2f us\n", med([&] {
      ++seq;
      hipLaunchKernelGGL(k_flag_inline, 1, 64, 0, s, d_line, seq);
      while (line[7] != seq || line[15] != seq) { }
      for (int k = 0; k < 16; ++k)
        if ((k & 7) != 7 && line[k] != seq * 1000003ull + k) ++torn;
    }, 20000));
    printf("   words seen stale behind their line's seq: %ld of %d\n", torn, 20050 * 14);
  }
  printf("H2D 4KB pinned + launch + spin flag  %7.2f us\n", med([&] { ++seq; hipMemcpyAsync(d, pin, 4096, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_flag, 1, 64, 0, s, (volatile unsigned long long *)(d_pin_coh + 64), d_pin_coh, seq); while (*flag != seq) { } }));
  printf("2 launches + spin flag               %7.2f us\n", med([&] { ++seq; hipLaunchKernelGGL(k_empty, 256, 256, 0, s, d); hipLaunchKernelGGL(k_flag, 1, 64, 0, s, (volatile unsigned long long *)(d_pin_coh + 64), d_pin_coh, seq); while (*flag != seq) { } }));
  printf("3 launches + spin flag               %7.2f us\n", med([&] { ++seq; hipLaunchKernelGGL(k_empty, 256, 256, 0, s, d); hipLaunchKernelGGL(k_empty, 256, 256, 0, s, d); hipLaunchKernelGGL(k_flag, 1, 64, 0, s, (volatile unsigned long long *)(d_pin_coh + 64), d_pin_coh, seq); while (*flag != seq) { } }));
  printf("10us kernel + sync                   %7.2f us\n", med([&] { hipLaunchKernelGGL(k_spin, 256, 256, 0, s, d, 1500); hipStreamSynchronize(s); }));
  printf("10us kernel + flag kernel + spin     %7.2f us\n", med([&] { ++seq; hipLaunchKernelGGL(k_spin, 256, 256, 0, s, d, 1500); hipLaunchKernelGGL(k_flag, 1, 64, 0, s, (volatile unsigned long long *)(d_pin_coh + 64), d_pin_coh, seq); while (*flag != seq) { } }));
  // 2.4 MB pinned / pageable H2D and zero-copy are measured by experiments/particles_hostcall.py
  printf("hipSetDevice                         %7.2f us\n", med([&] { hipSetDevice(0); }));
  hipPointerAttribute_t attr;
  printf("hipPointerGetAttributes (pageable)   %7.2f us\n", med([&] { (void)hipPointerGetAttributes(&attr, pageable.data()); (void)hipGetLastError(); }));
  printf("hipPointerGetAttributes (pinned)     %7.2f us\n", med([&] { (void)hipPointerGetAttributes(&attr, pin); }));
  printf("H2D 2.4MB pinned + sync              %7.2f us\n", med([&] { hipMemcpyAsync(d, pin, 1 << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }, 300));
  return 0;
}
