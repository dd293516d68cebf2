// Micro-benchmark: ISSUE cost (cycles per wave-instruction per SIMD) of every VALU
// instruction class the two hot kernels (match_lane_compact_kernel,
// score_poses_compact_kernel) are made of, on gfx950 -- the table bench.py prices the
// VALU-issue roofline with (profiles/r05_ubench_issue.json).  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 experiments/ubench_issue.hip -o experiments/bin/ubench_issue
//   experiments/bin/ubench_issue > gpurun_out/ubench_issue.jsonl
//
// Method: a wave runs `iters` x 128 copies of ONE instruction over 8 independent
// register chains (so that with a single wave per SIMD the instruction's own latency is
// covered up to 8 x its issue time), w waves per SIMD (256 x w blocks of 256 threads:
// one block = one wave on each SIMD of a CU).  Reported per (instruction, w):
//   cyc_wave   mean over waves of (s_memtime end - start) / (instructions x w)
//              -- shader-clock cycles a SIMD spends per wave-instruction,
//   ns_launch  launch duration (HIP events) / (instructions per SIMD),
//   mhz        the shader clock the two give together with s_memrealtime (100 MHz).
// A class "issues in N cycles" when cyc_wave settles at N once w is large enough to hide
// its latency (w = 4, 6, 8 agree).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

struct Stamp
{
  unsigned long long c0, c1, r0, r1;
};

// Operand shapes.  a* = 64-bit chains, i* = 32-bit chains; `c`, `j` loop-invariant sources.
#define D2(ins) \
  REP16(asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(c)); \
        asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(c)); \
        asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(c)); \
        asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(c));)
#define D3(ins) \
  REP16(asm volatile(ins " %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a1) : "v"(c)); \
        asm volatile(ins " %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a3) : "v"(c)); \
        asm volatile(ins " %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a5) : "v"(c)); \
        asm volatile(ins " %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile(ins " %0, %0, %1, %1" : "+v"(a7) : "v"(c));)
#define D1(ins) \
  REP16(asm volatile(ins " %0, %0" : "+v"(a0)); asm volatile(ins " %0, %0" : "+v"(a1)); \
        asm volatile(ins " %0, %0" : "+v"(a2)); asm volatile(ins " %0, %0" : "+v"(a3)); \
        asm volatile(ins " %0, %0" : "+v"(a4)); asm volatile(ins " %0, %0" : "+v"(a5)); \
        asm volatile(ins " %0, %0" : "+v"(a6)); asm volatile(ins " %0, %0" : "+v"(a7));)
// f64 <- f64, i32 (v_ldexp_f64)
#define DI(ins) \
  REP16(asm volatile(ins " %0, %0, %1" : "+v"(a0) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(a1) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" : "+v"(a2) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(a3) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" : "+v"(a4) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(a5) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" : "+v"(a6) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(a7) : "v"(j));)
#define I2(ins) \
  REP16(asm volatile(ins " %0, %0, %1" : "+v"(i0) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(i1) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" : "+v"(i2) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(i3) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" : "+v"(i4) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(i5) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" : "+v"(i6) : "v"(j)); asm volatile(ins " %0, %0, %1" : "+v"(i7) : "v"(j));)
#define I2S(ins, suffix) \
  REP16(asm volatile(ins " %0, %0, %1" suffix : "+v"(i0) : "v"(j)); asm volatile(ins " %0, %0, %1" suffix : "+v"(i1) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" suffix : "+v"(i2) : "v"(j)); asm volatile(ins " %0, %0, %1" suffix : "+v"(i3) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" suffix : "+v"(i4) : "v"(j)); asm volatile(ins " %0, %0, %1" suffix : "+v"(i5) : "v"(j)); \
        asm volatile(ins " %0, %0, %1" suffix : "+v"(i6) : "v"(j)); asm volatile(ins " %0, %0, %1" suffix : "+v"(i7) : "v"(j));)
#define I3(ins) \
  REP16(asm volatile(ins " %0, %0, %1, %1" : "+v"(i0) : "v"(j)); asm volatile(ins " %0, %0, %1, %1" : "+v"(i1) : "v"(j)); \
        asm volatile(ins " %0, %0, %1, %1" : "+v"(i2) : "v"(j)); asm volatile(ins " %0, %0, %1, %1" : "+v"(i3) : "v"(j)); \
        asm volatile(ins " %0, %0, %1, %1" : "+v"(i4) : "v"(j)); asm volatile(ins " %0, %0, %1, %1" : "+v"(i5) : "v"(j)); \
        asm volatile(ins " %0, %0, %1, %1" : "+v"(i6) : "v"(j)); asm volatile(ins " %0, %0, %1, %1" : "+v"(i7) : "v"(j));)
#define I1(ins, suffix) \
  REP16(asm volatile(ins " %0, %0" suffix : "+v"(i0)); asm volatile(ins " %0, %0" suffix : "+v"(i1)); \
        asm volatile(ins " %0, %0" suffix : "+v"(i2)); asm volatile(ins " %0, %0" suffix : "+v"(i3)); \
        asm volatile(ins " %0, %0" suffix : "+v"(i4)); asm volatile(ins " %0, %0" suffix : "+v"(i5)); \
        asm volatile(ins " %0, %0" suffix : "+v"(i6)); asm volatile(ins " %0, %0" suffix : "+v"(i7));)
// 32-bit <- 64-bit (v_cvt_i32_f64 ...)
#define ID(ins) \
  REP16(asm volatile(ins " %0, %1" : "=v"(i0) : "v"(a0)); asm volatile(ins " %0, %1" : "=v"(i1) : "v"(a1)); \
        asm volatile(ins " %0, %1" : "=v"(i2) : "v"(a2)); asm volatile(ins " %0, %1" : "=v"(i3) : "v"(a3)); \
        asm volatile(ins " %0, %1" : "=v"(i4) : "v"(a4)); asm volatile(ins " %0, %1" : "=v"(i5) : "v"(a5)); \
        asm volatile(ins " %0, %1" : "=v"(i6) : "v"(a6)); asm volatile(ins " %0, %1" : "=v"(i7) : "v"(a7));)
// 64-bit <- 32-bit (v_cvt_f64_i32 ...)
#define DFI(ins) \
  REP16(asm volatile(ins " %0, %1" : "=v"(a0) : "v"(i0)); asm volatile(ins " %0, %1" : "=v"(a1) : "v"(i1)); \
        asm volatile(ins " %0, %1" : "=v"(a2) : "v"(i2)); asm volatile(ins " %0, %1" : "=v"(a3) : "v"(i3)); \
        asm volatile(ins " %0, %1" : "=v"(a4) : "v"(i4)); asm volatile(ins " %0, %1" : "=v"(a5) : "v"(i5)); \
        asm volatile(ins " %0, %1" : "=v"(a6) : "v"(i6)); asm volatile(ins " %0, %1" : "=v"(a7) : "v"(i7));)
// compares (write vcc): 64-bit and 32-bit sources
#define CD(ins) \
  REP16(asm volatile(ins " vcc, %0, %1" : : "v"(a0), "v"(c) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(a1), "v"(c) : "vcc"); \
        asm volatile(ins " vcc, %0, %1" : : "v"(a2), "v"(c) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(a3), "v"(c) : "vcc"); \
        asm volatile(ins " vcc, %0, %1" : : "v"(a4), "v"(c) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(a5), "v"(c) : "vcc"); \
        asm volatile(ins " vcc, %0, %1" : : "v"(a6), "v"(c) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(a7), "v"(c) : "vcc");)
#define CI(ins) \
  REP16(asm volatile(ins " vcc, %0, %1" : : "v"(i0), "v"(j) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(i1), "v"(j) : "vcc"); \
        asm volatile(ins " vcc, %0, %1" : : "v"(i2), "v"(j) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(i3), "v"(j) : "vcc"); \
        asm volatile(ins " vcc, %0, %1" : : "v"(i4), "v"(j) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(i5), "v"(j) : "vcc"); \
        asm volatile(ins " vcc, %0, %1" : : "v"(i6), "v"(j) : "vcc"); asm volatile(ins " vcc, %0, %1" : : "v"(i7), "v"(j) : "vcc");)
// compares into an SGPR pair (VOP3 encoding)
#define CDS(ins) \
  REP16(asm volatile(ins " %0, %1, %2" : "=s"(m0) : "v"(a0), "v"(c)); asm volatile(ins " %0, %1, %2" : "=s"(m1) : "v"(a1), "v"(c)); \
        asm volatile(ins " %0, %1, %2" : "=s"(m0) : "v"(a2), "v"(c)); asm volatile(ins " %0, %1, %2" : "=s"(m1) : "v"(a3), "v"(c)); \
        asm volatile(ins " %0, %1, %2" : "=s"(m0) : "v"(a4), "v"(c)); asm volatile(ins " %0, %1, %2" : "=s"(m1) : "v"(a5), "v"(c)); \
        asm volatile(ins " %0, %1, %2" : "=s"(m0) : "v"(a6), "v"(c)); asm volatile(ins " %0, %1, %2" : "=s"(m1) : "v"(a7), "v"(c));)
// select on vcc / on an SGPR pair
#define SELV \
  REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i0) : "v"(j)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i1) : "v"(j)); \
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i2) : "v"(j)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i3) : "v"(j)); \
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i4) : "v"(j)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i5) : "v"(j)); \
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i6) : "v"(j)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i7) : "v"(j));)
#define SELS \
  REP16(asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i0) : "v"(j), "s"(m0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i1) : "v"(j), "s"(m0)); \
        asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i2) : "v"(j), "s"(m0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i3) : "v"(j), "s"(m0)); \
        asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i4) : "v"(j), "s"(m0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i5) : "v"(j), "s"(m0)); \
        asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i6) : "v"(j), "s"(m0)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i7) : "v"(j), "s"(m0));)
#define SELV3 \
  REP16(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i0) : "v"(i1), "v"(j)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i2) : "v"(i3), "v"(j)); \
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i4) : "v"(i5), "v"(j)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i6) : "v"(i7), "v"(j)); \
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i1) : "v"(i0), "v"(j)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i3) : "v"(i2), "v"(j)); \
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i5) : "v"(i4), "v"(j)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i7) : "v"(i6), "v"(j));)
#define SELE64 \
  REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i0) : "v"(j)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i1) : "v"(j)); \
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i2) : "v"(j)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i3) : "v"(j)); \
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i4) : "v"(j)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i5) : "v"(j)); \
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i6) : "v"(j)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i7) : "v"(j));)
// 32-bit <- 32-bit with a scalar / literal source
#define MOVS(src) \
  REP16(asm volatile("v_mov_b32 %0, " src : "=v"(i0) : "s"(s0)); asm volatile("v_mov_b32 %0, " src : "=v"(i1) : "s"(s0)); \
        asm volatile("v_mov_b32 %0, " src : "=v"(i2) : "s"(s0)); asm volatile("v_mov_b32 %0, " src : "=v"(i3) : "s"(s0)); \
        asm volatile("v_mov_b32 %0, " src : "=v"(i4) : "s"(s0)); asm volatile("v_mov_b32 %0, " src : "=v"(i5) : "s"(s0)); \
        asm volatile("v_mov_b32 %0, " src : "=v"(i6) : "s"(s0)); asm volatile("v_mov_b32 %0, " src : "=v"(i7) : "s"(s0));)
// lane <-> scalar
#define RDL \
  REP16(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(i0)); asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(i1)); \
        asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(i2)); asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(i3)); \
        asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(i4)); asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(i5)); \
        asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(i6)); asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(i7));)
#define WRL \
  REP16(asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i0) : "s"(s0)); asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i1) : "s"(s0)); \
        asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i2) : "s"(s0)); asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i3) : "s"(s0)); \
        asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i4) : "s"(s0)); asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i5) : "s"(s0)); \
        asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i6) : "s"(s0)); asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(i7) : "s"(s0));)

template <int OP>
__global__ void __launch_bounds__(256) k(double * out, Stamp * stamps, int iters, double seed)
{
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
         a7 = a0 + 7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const double c = seed * 0.5;
  const int j = static_cast<int>(seed) + 0x01020304;
  int s0 = __builtin_amdgcn_readfirstlane(static_cast<int>(seed)), s1 = s0;
  unsigned long long m0 = 0x5555555555555555ull + static_cast<unsigned long long>(static_cast<unsigned>(s0)), m1 = m0;
  asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(i0), "v"(j) : "vcc");
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
  {
    if constexpr (OP == 0) { D2("v_add_f64") }
    if constexpr (OP == 1) { D2("v_mul_f64") }
    if constexpr (OP == 2) { D3("v_fma_f64") }
    if constexpr (OP == 3) { D2("v_max_f64") }
    if constexpr (OP == 4) { D1("v_rndne_f64") }
    if constexpr (OP == 5) { D1("v_floor_f64") }
    if constexpr (OP == 6) { DI("v_ldexp_f64") }
    if constexpr (OP == 7) { ID("v_cvt_i32_f64") }
    if constexpr (OP == 8) { ID("v_cvt_u32_f64") }
    if constexpr (OP == 9) { DFI("v_cvt_f64_i32") }
    if constexpr (OP == 10) { DFI("v_cvt_f64_u32") }
    if constexpr (OP == 11) { CD("v_cmp_lt_f64") }
    if constexpr (OP == 12) { CDS("v_cmp_gt_f64") }
    if constexpr (OP == 13) { CD("v_cmp_u_f64") }
    if constexpr (OP == 14) { CI("v_cmp_gt_u16") }
    if constexpr (OP == 15) { CI("v_cmp_le_u32") }
    if constexpr (OP == 16) { SELV }
    if constexpr (OP == 17) { SELS }
    if constexpr (OP == 18) { I1("v_mov_b32", "") }
    if constexpr (OP == 19) { D1("v_mov_b64") }
    if constexpr (OP == 20) { I1("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf") }
    if constexpr (OP == 21) { I1("v_mov_b32_dpp", " row_bcast:31 row_mask:0xf bank_mask:0xf") }
    if constexpr (OP == 22) { RDL }
    if constexpr (OP == 23) { WRL }
    if constexpr (OP == 24) { I3("v_perm_b32") }
    if constexpr (OP == 25) { I3("v_alignbyte_b32") }
    if constexpr (OP == 26) { I2("v_add_u32") }
    if constexpr (OP == 27) { I3("v_add3_u32") }
    if constexpr (OP == 28) { I2("v_and_b32") }
    if constexpr (OP == 29) { I2("v_lshlrev_b32") }
    if constexpr (OP == 30) { I3("v_lshl_add_u32") }
    if constexpr (OP == 31) { I3("v_and_or_b32") }
    if constexpr (OP == 32) { I3("v_mad_u32_u24") }
    if constexpr (OP == 33) { I2("v_mul_u32_u24") }
    if constexpr (OP == 34) { I2("v_mul_lo_u32") }
    if constexpr (OP == 35)
    {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a4) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a5) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a6) : "v"(i0), "v"(j) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a7) : "v"(i0), "v"(j) : "vcc");)
    }
    if constexpr (OP == 36)
    {
      REP16(asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a0) : "v"(c)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a2) : "v"(c)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a3) : "v"(c));
            asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a4) : "v"(c)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a6) : "v"(c)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a7) : "v"(c));)
    }
    if constexpr (OP == 37) { I2S("v_lshrrev_b32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1") }
    if constexpr (OP == 38) { I2("v_min_u32") }
    if constexpr (OP == 39) { I2("v_mul_f32") }
    if constexpr (OP == 40) { I3("v_fma_f32") }
    if constexpr (OP == 41) { D3("v_pk_fma_f32") }
    if constexpr (OP == 42) { D2("v_pk_mul_f32") }
    if constexpr (OP == 43) { D2("v_pk_add_f32") }
    if constexpr (OP == 44) { I1("v_cvt_f32_u32", "") }
    if constexpr (OP == 45) { I1("v_rcp_f32", "") }
    if constexpr (OP == 46) { D1("v_rcp_f64") }
    if constexpr (OP == 47) { I1("v_exp_f32", "") }
    if constexpr (OP == 48) { I1("v_floor_f32", "") }
    if constexpr (OP == 49) { I1("v_cvt_i32_f32", "") }
    if constexpr (OP == 52) { SELV3 }
    if constexpr (OP == 53) { SELE64 }
    if constexpr (OP == 54) { I2("v_sub_u32") }
    if constexpr (OP == 55) { I2("v_or_b32") }
    if constexpr (OP == 56) { I2("v_xor_b32") }
    if constexpr (OP == 57) { I2("v_lshrrev_b32") }
    if constexpr (OP == 58) { I2("v_max_u32") }
    // the particle kernel's screening pass (round 5, second table)
    if constexpr (OP == 80) { I1("v_cvt_flr_i32_f32", "") }
    if constexpr (OP == 81) { I1("v_fract_f32", "") }
    if constexpr (OP == 82) { I2("v_min_f32") }
    if constexpr (OP == 83) { I3("v_alignbit_b32") }
    if constexpr (OP == 84) { I3("v_mad_i32_i24") }
    if constexpr (OP == 85)
    {
      REP16(asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i0) : "v"(j) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i1) : "v"(j) : "vcc");
            asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i2) : "v"(j) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i3) : "v"(j) : "vcc");
            asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i4) : "v"(j) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i5) : "v"(j) : "vcc");
            asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i6) : "v"(j) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i7) : "v"(j) : "vcc");)
    }
    if constexpr (OP == 86) { CI("v_cmp_lt_f32") }
    if constexpr (OP == 87) { I1("v_ffbl_b32", "") }
    if constexpr (OP == 88) { I2("v_bcnt_u32_b32") }
    if constexpr (OP == 89) { I1("v_bfrev_b32", "") }
    if constexpr (OP == 90)
    {
      // compare + add-with-carry pairs, as the screening pass shifts its "near" bits in
      REP16(asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(i0) : "v"(i1), "v"(j) : "vcc");
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(i2) : "v"(i3), "v"(j) : "vcc");
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(i4) : "v"(i5), "v"(j) : "vcc");
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(i6) : "v"(i7), "v"(j) : "vcc");)
    }
    if constexpr (OP == 59) { I2("v_add_f32") }
    if constexpr (OP == 60) { MOVS("%1") }
    if constexpr (OP == 61) { MOVS("0x85ebc8a0") }
    if constexpr (OP == 62) { D2("v_fmac_f64") }
    if constexpr (OP == 63) { D1("v_ceil_f64") }
    if constexpr (OP == 64) { ID("v_frexp_exp_i32_f64") }
    if constexpr (OP == 65) { I3("v_bfe_u32") }
    if constexpr (OP == 66) { I3("v_max3_u32") }
    if constexpr (OP == 67) { I2("v_fmac_f32") }
    if constexpr (OP == 68) { I1("v_mov_b32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    if constexpr (OP == 69) { I2S("v_add_u32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    // one select among seven FP64 instructions: is the e32 (implicit vcc) form's 22 cycles a cost of the
    // instruction wherever it stands, or of a run of them?
    if constexpr (OP == 70)
    {
      REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i3) : "v"(j));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a7) : "v"(c));)
    }
    if constexpr (OP == 71)
    {
      REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a1) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(i3) : "v"(j), "s"(m0));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a5) : "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a7) : "v"(c));)
    }
    // ... and with vcc freshly written by a compare right before it (the usual pair)
    if constexpr (OP == 72)
    {
      REP16(asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(i0) : "v"(a0), "v"(c), "v"(j) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(i1) : "v"(a1), "v"(c), "v"(j) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(i2) : "v"(a2), "v"(c), "v"(j) : "vcc");
            asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(i3) : "v"(a3), "v"(c), "v"(j) : "vcc");)
    }
    if constexpr (OP == 73)
    {
      REP16(asm volatile("v_cmp_lt_f64 %1, %2, %3\n\tv_cndmask_b32 %0, %0, %4, %1" : "+v"(i0), "=&s"(m0) : "v"(a0), "v"(c), "v"(j));
            asm volatile("v_cmp_lt_f64 %1, %2, %3\n\tv_cndmask_b32 %0, %0, %4, %1" : "+v"(i1), "=&s"(m1) : "v"(a1), "v"(c), "v"(j));
            asm volatile("v_cmp_lt_f64 %1, %2, %3\n\tv_cndmask_b32 %0, %0, %4, %1" : "+v"(i2), "=&s"(m0) : "v"(a2), "v"(c), "v"(j));
            asm volatile("v_cmp_lt_f64 %1, %2, %3\n\tv_cndmask_b32 %0, %0, %4, %1" : "+v"(i3), "=&s"(m1) : "v"(a3), "v"(c), "v"(j));)
    }
    // two classes interleaved: does an FP64 instruction dual-issue with / hide a 32-bit one?
    if constexpr (OP == 50)
    {
      REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(j));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(j));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i5) : "v"(j));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(i7) : "v"(j));)
    }
    if constexpr (OP == 51)
    {
      REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile("v_mov_b32 %0, %0" : "+v"(i1));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile("v_mov_b32 %0, %0" : "+v"(i3));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile("v_mov_b32 %0, %0" : "+v"(i5));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile("v_mov_b32 %0, %0" : "+v"(i7));)
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0)
  {
    Stamp s = {c0, c1, r0, r1};
    stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 +
                                               i7 + static_cast<double>(m0 + m1) + s0 + s1;
}

static int g_cus = 256;
static int g_only = -1;   // argv[1]: run this instruction only (counter calibration under rocprofv3)
static int g_waves = 0;   // argv[2]: ... at this many waves per SIMD only

template <int OP>
void run(const char * name, const char * pmc_class)
{
  const int insts_per_iter = 128;
  const int iters = 1000;
  const int max_blocks = g_cus * 8;
  double * out;
  Stamp * d_stamps;
  hipMalloc(&out, static_cast<size_t>(max_blocks) * 256 * 8);
  hipMalloc(&d_stamps, static_cast<size_t>(max_blocks) * 4 * sizeof(Stamp));
  std::vector<Stamp> stamps(static_cast<size_t>(max_blocks) * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if (g_only >= 0 && g_only != OP) return;
  for (int w : {1, 2, 4, 6, 8})
  {
    if (g_waves > 0 && w != g_waves) continue;
    const int blocks = g_cus * w;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, d_stamps, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, d_stamps, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(stamps.data(), d_stamps, static_cast<size_t>(blocks) * 4 * sizeof(Stamp), hipMemcpyDeviceToHost);
    double cyc = 0.0, real = 0.0;
    unsigned long long first = ~0ull, last = 0;
    for (int i = 0; i < blocks * 4; ++i)
    {
      cyc += static_cast<double>(stamps[i].c1 - stamps[i].c0);
      real += static_cast<double>(stamps[i].r1 - stamps[i].r0);
      if (stamps[i].r0 < first) first = stamps[i].r0;
      if (stamps[i].r1 > last) last = stamps[i].r1;
    }
    cyc /= blocks * 4;
    real /= blocks * 4;
    const double inst_wave = static_cast<double>(iters) * insts_per_iter;
    const double inst_simd = inst_wave * w;
    // s_memrealtime ticks at 100 MHz: `real` ticks = real * 10 ns
    const double mhz = real > 0 ? cyc / (real * 10.0e-9) / 1.0e6 : 0.0;
    printf("{\"op\": %d, \"instruction\": \"%s\", \"pmc_class\": \"%s\", \"waves_per_simd\": %d, \"cyc_wave\": %.3f, "
           "\"ns_launch\": %.4f, \"ns_span\": %.4f, \"memtime_mhz\": %.1f}\n",
           OP, name, pmc_class, w, cyc / inst_simd, ms * 1.0e6 / inst_simd,
           static_cast<double>(last - first) * 10.0 / inst_simd, mhz);
    fflush(stdout);
  }
  hipFree(out);
  hipFree(d_stamps);
}

int main(int argc, char ** argv)
{
  if (argc > 1) g_only = atoi(argv[1]);
  if (argc > 2) g_waves = atoi(argv[2]);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  g_cus = prop.multiProcessorCount;
  int wall_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"wall_clock_khz\": %d}\n", prop.name,
         prop.gcnArchName, g_cus, prop.clockRate, wall_khz);
  run<0>("v_add_f64", "FP64");
  run<1>("v_mul_f64", "FP64");
  run<2>("v_fma_f64", "FP64");
  run<3>("v_max_f64", "other");
  run<4>("v_rndne_f64", "other");
  run<5>("v_floor_f64", "other");
  run<6>("v_ldexp_f64", "other");
  run<7>("v_cvt_i32_f64", "CVT");
  run<8>("v_cvt_u32_f64", "CVT");
  run<9>("v_cvt_f64_i32", "CVT");
  run<10>("v_cvt_f64_u32", "CVT");
  run<11>("v_cmp_lt_f64 vcc", "other");
  run<12>("v_cmp_gt_f64 sgpr", "other");
  run<13>("v_cmp_u_f64 vcc", "other");
  run<14>("v_cmp_gt_u16 vcc", "INT32");
  run<15>("v_cmp_le_u32 vcc", "INT32");
  run<16>("v_cndmask_b32 vcc", "other");
  run<17>("v_cndmask_b32 sgpr", "other");
  run<18>("v_mov_b32", "other");
  run<19>("v_mov_b64", "other");
  run<20>("v_mov_b32_dpp row_shr", "other");
  run<21>("v_mov_b32_dpp row_bcast31", "other");
  run<22>("v_readlane_b32", "other");
  run<23>("v_writelane_b32", "other");
  run<24>("v_perm_b32", "other");
  run<25>("v_alignbyte_b32", "other");
  run<26>("v_add_u32", "INT32");
  run<27>("v_add3_u32", "INT32");
  run<28>("v_and_b32", "INT32");
  run<29>("v_lshlrev_b32", "INT32");
  run<30>("v_lshl_add_u32", "INT32");
  run<31>("v_and_or_b32", "INT32");
  run<32>("v_mad_u32_u24", "INT32");
  run<33>("v_mul_u32_u24", "INT32");
  run<34>("v_mul_lo_u32", "INT32");
  run<35>("v_mad_u64_u32", "INT64");
  run<36>("v_lshl_add_u64", "INT64");
  run<37>("v_lshrrev_b32_sdwa", "INT32");
  run<38>("v_min_u32", "INT32");
  run<39>("v_mul_f32", "FP32");
  run<40>("v_fma_f32", "FP32");
  run<41>("v_pk_fma_f32", "FP32");
  run<42>("v_pk_mul_f32", "FP32");
  run<43>("v_pk_add_f32", "FP32");
  run<44>("v_cvt_f32_u32", "CVT");
  run<45>("v_rcp_f32", "TRANS");
  run<46>("v_rcp_f64", "TRANS");
  run<47>("v_exp_f32", "TRANS");
  run<48>("v_floor_f32", "other");
  run<49>("v_cvt_i32_f32", "CVT");
  run<52>("v_cndmask_b32 vcc (dst != src)", "other");
  run<53>("v_cndmask_b32_e64 vcc", "other");
  run<54>("v_sub_u32", "INT32");
  run<55>("v_or_b32", "INT32");
  run<56>("v_xor_b32", "INT32");
  run<57>("v_lshrrev_b32", "INT32");
  run<58>("v_max_u32", "INT32");
  run<80>("v_cvt_flr_i32_f32", "CVT");
  run<81>("v_fract_f32", "other");
  run<82>("v_min_f32", "other");
  run<83>("v_alignbit_b32", "other");
  run<84>("v_mad_i32_i24", "INT32");
  run<85>("v_addc_co_u32 vcc", "INT32");
  run<86>("v_cmp_lt_f32 vcc", "other");
  run<87>("v_ffbl_b32", "other");
  run<88>("v_bcnt_u32_b32", "other");
  run<89>("v_bfrev_b32", "other");
  run<90>("v_cmp_lt_f32 vcc + v_addc_co_u32 vcc pairs (64 per 128)", "mix");
  run<59>("v_add_f32", "FP32");
  run<60>("v_mov_b32 from sgpr", "other");
  run<61>("v_mov_b32 literal", "other");
  run<62>("v_fmac_f64", "FP64");
  run<63>("v_ceil_f64", "other");
  run<64>("v_frexp_exp_i32_f64", "other");
  run<65>("v_bfe_u32", "INT32");
  run<66>("v_max3_u32", "INT32");
  run<67>("v_fmac_f32", "FP32");
  run<68>("v_mov_b32_dpp quad_perm", "other");
  run<69>("v_add_u32_dpp quad_perm", "INT32");
  run<70>("7 x v_fma_f64 + 1 x v_cndmask_b32 vcc", "mix");
  run<71>("7 x v_fma_f64 + 1 x v_cndmask_b32 sgpr", "mix");
  run<72>("v_cmp_lt_f64 vcc + v_cndmask_b32 vcc pairs", "mix");
  run<73>("v_cmp_lt_f64 sgpr + v_cndmask_b32 sgpr pairs", "mix");
  run<50>("v_add_f64 + v_add_u32 alternating", "mix");
  run<51>("v_fma_f64 + v_mov_b32 alternating", "mix");
  return 0;
}
