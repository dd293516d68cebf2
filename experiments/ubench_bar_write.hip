// Can the host write a staged image straight into device memory (fine-grained device
// allocation through the PCIe BAR), so that the kernel that consumes it reads local memory
// instead of pinned host memory across PCIe?  Prints: host copy time for 16 / 90 / 400 KB into
// (a) pinned host memory, (b) the device allocation; and the duration + launch-to-flag latency
// of a kernel that reads the image from (a) in place, from (b), and after a hipMemcpyAsync.
//   hipcc --offload-arch=gfx950 -O2 experiments/ubench_bar_write.hip -o experiments/bin/ubench_bar_write
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <csignal>
#include <csetjmp>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void consume(const double2 * src, size_t n16, double * out, volatile unsigned long long * flag, unsigned long long seq, unsigned int * counter)
{
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
      __hip_atomic_store(const_cast<unsigned long long *>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

int main()
{
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t cap = 512 * 1024;
  double * pinned = nullptr; double * pinned_dev = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void **>(&pinned), cap, hipHostMallocDefault));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&pinned_dev), pinned, 0));
  double * fine = nullptr;
  hipError_t fe = hipExtMallocWithFlags(reinterpret_cast<void **>(&fine), cap, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained) -> %s\n", hipGetErrorString(fe));
  double * plain = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&plain), cap));
  double * out = nullptr; unsigned int * counter = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&out), 8 * 512 * 256)); CK(hipMalloc(reinterpret_cast<void **>(&counter), 4));
  CK(hipMemset(out, 0, 8 * 512 * 256)); CK(hipMemset(counter, 0, 4)); CK(hipDeviceSynchronize());
  unsigned long long * flag = nullptr, * flag_dev = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void **>(&flag), 64, hipHostMallocDefault));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&flag_dev), flag, 0));
  *flag = 0;
  std::vector<double> src(cap / 8);
  for (size_t i = 0; i < src.size(); ++i) src[i] = 1.0 + i * 1e-9;

  bool fine_writable = false;
  if (fe == hipSuccess)
  {
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) { std::memcpy(fine, src.data(), 4096); fine_writable = true; }
    else printf("host write to the fine-grained device allocation faulted\n");
    signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
  }
  printf("fine-grained device memory host-writable: %d\n", fine_writable ? 1 : 0);
  unsigned long long seq = 0;
  const size_t sizes[3] = {16 * 1024, 90 * 1024, 400 * 1024};
  for (size_t bytes : sizes)
  {
    for (int mode = 0; mode < 3; ++mode)
    {
      // 0: pinned, kernel reads in place; 1: host writes device memory, kernel reads it; 2: pinned + hipMemcpyAsync + kernel
      if (mode == 1 && !fine_writable) continue;
      std::vector<double> t_host, t_total;
      const unsigned blocks = static_cast<unsigned>(std::min<size_t>((bytes / 16 + 255) / 256, 512));
      for (int rep = -20; rep < 300; ++rep)
      {
        src[rep & 1023] += 1.0;
        const double a = now_us();
        if (mode == 1) { std::memcpy(fine, src.data(), bytes); __builtin_ia32_sfence(); }
        else std::memcpy(pinned, src.data(), bytes);
        const double b = now_us();
        const double2 * in = reinterpret_cast<const double2 *>(mode == 0 ? pinned_dev : mode == 1 ? fine : plain);
        if (mode == 2) CK(hipMemcpyAsync(plain, pinned, bytes, hipMemcpyHostToDevice, st));
        ++seq;
        hipLaunchKernelGGL(consume, dim3(blocks), dim3(256), 0, st, in, bytes / 16, out, flag_dev, seq, counter);
        while (*reinterpret_cast<volatile unsigned long long *>(flag) != seq) {}
        const double c = now_us();
        if (rep >= 0) { t_host.push_back(b - a); t_total.push_back(c - a); }
      }
      std::sort(t_host.begin(), t_host.end()); std::sort(t_total.begin(), t_total.end());
      printf("%4zu KB  mode %d (%s): host copy %6.2f us, copy + launch -> flag %6.2f us\n", bytes / 1024, mode,
             mode == 0 ? "pinned, read in place" : mode == 1 ? "host writes device memory" : "pinned + memcpyAsync",
             t_host[t_host.size() / 2], t_total[t_total.size() / 2]);
    }
  }
  // check the sum once (mode 1 data correctness)
  if (fine_writable)
  {
    CK(hipMemsetAsync(out, 0, 8 * 512 * 256, st));
    for (int i = 0; i < 2048; ++i) src[i] = 3.0 + i;
    std::memcpy(fine, src.data(), 16384); __builtin_ia32_sfence();
    ++seq;
    hipLaunchKernelGGL(consume, dim3(4), dim3(256), 0, st, reinterpret_cast<const double2 *>(fine), 1024, out, flag_dev, seq, counter);
    CK(hipStreamSynchronize(st));
    std::vector<double> back(1024);
    CK(hipMemcpy(back.data(), out, 8 * 1024, hipMemcpyDeviceToHost));
    double got = 0, want = 0;
    for (double v : back) got += v;
    for (int i = 0; i < 2048; ++i) want += src[i];
    printf("sum through device memory written by the host: got %.6f want %.6f\n", got, want);
  }
  return 0;
}
