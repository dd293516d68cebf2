// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the
// VALU instructions the NDT kernels are made of, gfx950.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 experiments/ubench_f64.hip -o experiments/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void __launch_bounds__(256) k(double * out, int iters, double seed)
{
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
         a6 = a0 + 6, a7 = a0 + 7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const double c = seed * 0.5;
  for (int it = 0; it < iters; ++it)
  {
    if (OP == 0)
    {
      REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(c));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a7) : "v"(c));)
    }
    if (OP == 1)
    {
      REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(c));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a7) : "v"(c));)
    }
    if (OP == 2)
    {
      REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a3) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a7) : "v"(c));)
    }
    if (OP == 3)
    {
      REP16(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i1) : "v"(a1));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i3) : "v"(a3));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i4) : "v"(a4)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i5) : "v"(a5));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i6) : "v"(a6)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i7) : "v"(a7));)
    }
    if (OP == 4)
    {
      REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a0), "v"(c), "v"(i0), "v"(i1) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a1), "v"(c), "v"(i2), "v"(i3) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a2), "v"(c), "v"(i4), "v"(i5) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a3), "v"(c), "v"(i6), "v"(i7) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a4), "v"(c), "v"(i0), "v"(i1) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a5), "v"(c), "v"(i2), "v"(i3) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a6), "v"(c), "v"(i4), "v"(i5) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a7), "v"(c), "v"(i6), "v"(i7) : "vcc");)
    }
    if (OP == 5)
    {
      REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i7)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(i7));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(i7)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(i7));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(i4) : "v"(i7)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i5) : "v"(i7));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(i6) : "v"(i7)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(i7));)
    }
    if (OP == 6)
    {
      REP16(asm volatile("v_rndne_f64 %0, %0" : "+v"(a0)); asm volatile("v_rndne_f64 %0, %0" : "+v"(a1));
            asm volatile("v_rndne_f64 %0, %0" : "+v"(a2)); asm volatile("v_rndne_f64 %0, %0" : "+v"(a3));
            asm volatile("v_rndne_f64 %0, %0" : "+v"(a4)); asm volatile("v_rndne_f64 %0, %0" : "+v"(a5));
            asm volatile("v_rndne_f64 %0, %0" : "+v"(a6)); asm volatile("v_rndne_f64 %0, %0" : "+v"(a7));)
    }
    if (OP == 7)
    {
      REP16(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a0) : "v"(i0)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a1) : "v"(i0));
            asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a2) : "v"(i0)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a3) : "v"(i0));
            asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a4) : "v"(i0)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a5) : "v"(i0));
            asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a6) : "v"(i0)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a7) : "v"(i0));)
    }
    if (OP == 8)
    {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(i0), "v"(i1) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(i0), "v"(i1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(i0), "v"(i1) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(i0), "v"(i1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a4) : "v"(i0), "v"(i1) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a5) : "v"(i0), "v"(i1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a6) : "v"(i0), "v"(i1) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a7) : "v"(i0), "v"(i1) : "vcc");)
    }
    if (OP == 9)
    {
      REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i0) : "v"(i7)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i1) : "v"(i7));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i2) : "v"(i7)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i3) : "v"(i7));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i4) : "v"(i7)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i5) : "v"(i7));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i6) : "v"(i7)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i0) : "v"(i7));)
    }
    if (OP == 10)
    {
      REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i0) : "v"(i7)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i1) : "v"(i7));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i2) : "v"(i7)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i3) : "v"(i7));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i4) : "v"(i7)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i5) : "v"(i7));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i6) : "v"(i7)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i0) : "v"(i7));)
    }
    if (OP == 11)
    {
      float f0 = i0, f1 = i1;
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(a7)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a1) : "v"(a7));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a2) : "v"(a7)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a3) : "v"(a7));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a4) : "v"(a7)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a5) : "v"(a7));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a6) : "v"(a7)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(a7));)
      (void)f0; (void)f1;
    }
    if (OP == 12)
    {
      REP16(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a0) : "v"(i0)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a1) : "v"(i1));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a2) : "v"(i2)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a3) : "v"(i3));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a4) : "v"(i4)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a5) : "v"(i5));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a6) : "v"(i6)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a7) : "v"(i7));)
    }
    if (OP == 13)
    {
      REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i7) : ); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i1) : "v"(i7));
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i2) : "v"(i7)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i3) : "v"(i7));
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i4) : "v"(i7)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i5) : "v"(i7));
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i6) : "v"(i7)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(i7));)
    }
    if (OP == 14)
    {
      REP16(asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i0) : "v"(i6), "v"(i7)); asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i1) : "v"(i6), "v"(i7));
            asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i2) : "v"(i6), "v"(i7)); asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i3) : "v"(i6), "v"(i7));
            asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i4) : "v"(i6), "v"(i7)); asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i5) : "v"(i6), "v"(i7));
            asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i0) : "v"(i6), "v"(i7)); asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(i1) : "v"(i6), "v"(i7));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int OP>
void run(const char * name, int insts_per_iter)
{
  double * out;
  hipMalloc(&out, 1024 * 256 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // 1 wave per SIMD: 256 CUs x 4 SIMDs -> 256 blocks of 256 threads; then 2/SIMD
  for (int blocks : {256, 512, 1024})
  {
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks / 256.0;
    const double inst = (double)iters * insts_per_iter * waves_per_simd;  // per SIMD
    printf("%-16s waves/SIMD=%.0f  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz)\n",
           name, waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
  }
  hipFree(out);
}

int main()
{
  run<0>("v_add_f64", 128);
  run<1>("v_mul_f64", 128);
  run<2>("v_fma_f64", 128);
  run<3>("v_cvt_i32_f64", 128);
  run<4>("cmp_f64+cndmask", 256);
  run<5>("v_add_u32", 128);
  run<6>("v_rndne_f64", 128);
  run<7>("v_ldexp_f64", 128);
  run<8>("v_mad_u64_u32", 128);
  run<9>("v_mul_lo_u32", 128);
  run<10>("v_mad_u32_u24", 128);
  run<11>("v_pk_fma_f32", 128);
  run<12>("v_cvt_f64_i32", 128);
  run<13>("v_cndmask_b32", 128);
  run<14>("v_med3_i32", 128);
  return 0;
}
