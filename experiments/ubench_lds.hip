// Micro-benchmark: LDS read throughput per CU for the address patterns of the lane
// search's map look-up (gfx950).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 experiments/ubench_lds.hip -o experiments/bin/ubench_lds
//
// 1024-thread blocks (16 waves per CU, as the search runs), one block per CU; every
// wave issues `iters` x 64 LDS reads back to back, eight in flight.  Reported: CU
// cycles per wave-instruction (lower bound of what one look-up costs the LDS pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int WIDTH>  // 1 = ds_read_u8, 4 = ds_read_b32, 8 = ds_read_b64
__global__ void __launch_bounds__(1024) k(unsigned * out, int iters, int pattern, int stride)
{
  extern __shared__ unsigned char lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += 1024) reinterpret_cast<unsigned *>(lds)[i] = i;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // an 8 x 8 patch of candidates 0.32 sub-cells apart: cell (x, y) = (lane/8, lane%8) * 0.32
  unsigned cx = (unsigned)((lane >> 3) * 0.32f + 0.3f), cy = (unsigned)((lane & 7) * 0.32f + 0.6f);
  unsigned a;
  if (pattern == 0) a = 0;                               // all lanes one address (broadcast)
  else if (pattern == 1) a = lane * WIDTH;               // consecutive (conflict free)
  else a = (cy + wave * 3) * stride + cx + wave * 5;     // the map look-up: rows `stride` apart
  a &= ~(unsigned)(WIDTH - 1);
  unsigned acc = 0;
  typedef const __attribute__((address_space(3))) unsigned char * p8;
  typedef const __attribute__((address_space(3))) unsigned * p32;
  typedef const __attribute__((address_space(3))) unsigned long long * p64;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int r = 0; r < 64; ++r)
    {
      const unsigned addr = a + r * 16 + (it & 3) * 2048;  // 64 distinct immediate offsets
      if (WIDTH == 1) acc += *reinterpret_cast<p8>(addr);
      if (WIDTH == 4) acc += *reinterpret_cast<p32>(addr);
      if (WIDTH == 8) acc += (unsigned)*reinterpret_cast<p64>(addr);
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int WIDTH>
void run(const char * name, int pattern, int stride, unsigned * out, int cus, double ghz)
{
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<WIDTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k<WIDTH>, dim3(cus), dim3(1024), 160 * 1024, 0, out, 10, pattern, stride);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<WIDTH>, dim3(cus), dim3(1024), 160 * 1024, 0, out, iters, pattern, stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_cu = 16.0 * iters * 64;
  printf("%-44s %8.3f ms  %6.2f CU-cycles per wave-instruction (at %.2f GHz)\n", name, ms,
         ms * 1e-3 * ghz * 1e9 / wave_instr_per_cu, ghz);
}

int main(int argc, char ** argv)
{
  const double ghz = argc > 1 ? atof(argv[1]) : 2.0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned * out;
  hipMalloc(&out, sizeof(unsigned) * 1024 * cus);
  printf("%d CUs, clock assumed %.2f GHz (prop %.2f)\n", cus, ghz, prop.clockRate * 1e-6);
  run<1>("ds_read_u8  broadcast", 0, 256, out, cus, ghz);
  run<1>("ds_read_u8  consecutive bytes", 1, 256, out, cus, ghz);
  run<1>("ds_read_u8  patch, row stride 256", 2, 256, out, cus, ghz);
  run<1>("ds_read_u8  patch, row stride 264", 2, 264, out, cus, ghz);
  run<1>("ds_read_u8  patch, row stride 260", 2, 260, out, cus, ghz);
  run<4>("ds_read_b32 broadcast", 0, 256, out, cus, ghz);
  run<4>("ds_read_b32 consecutive", 1, 256, out, cus, ghz);
  run<4>("ds_read_b32 patch, row stride 256", 2, 256, out, cus, ghz);
  run<4>("ds_read_b32 patch, row stride 264", 2, 264, out, cus, ghz);
  run<8>("ds_read_b64 broadcast", 0, 256, out, cus, ghz);
  run<8>("ds_read_b64 consecutive", 1, 256, out, cus, ghz);
  return 0;
}
