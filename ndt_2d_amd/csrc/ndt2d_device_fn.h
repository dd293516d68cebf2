// Device-side building blocks shared by the kernels: NDT::getIndex, Cell::score,
// the LDS staging of the grid and the wave reductions.  Included by the .hip
// translation units only.  All double arithmetic keeps the reference's operation
// order; the translation units are compiled with -ffp-contract=off.
#ifndef NDT2D_DEVICE_FN_H_
#define NDT2D_DEVICE_FN_H_

#include <hip/hip_runtime.h>
#include <math.h>

#include "ndt2d_kernels.h"

namespace ndt2d
{

constexpr int kWave = 64;
constexpr int kRecord = 12;  // NDT2D_MATCH_RECORD_DOUBLES
constexpr double kNoIndex = 1.0e308;
// Coordinate given to padding beams (lane index >= n_beams): far left of any
// grid, so they select the sentinel record and contribute exp(-inf) = +0.0.
constexpr double kPadCoord = -1.0e300;

// NDT::getIndex, reference src/ndt_model.cpp:203-218.
//   x < origin_x_ || y < origin_y_            -> outside
//   grid = (unsigned)((x - origin) / cell)    -> truncation toward zero
//   grid >= size                               -> outside
// `fx < size_x` on the un-truncated quotient is equivalent to
// `trunc(fx) < size_x` for fx >= 0 and keeps the conversion in range.
// Returns ncell (the sentinel record) for "outside".
template <bool POW2>
__device__ __forceinline__ uint32_t cell_index(const GridDesc & g, double px, double py)
{
  const double tx = px - g.origin_x;
  const double ty = py - g.origin_y;
  double fx, fy;
  if (POW2)
  {
    // cell_size is a power of two: multiplying by its exact reciprocal gives
    // the correctly rounded quotient, bit-identical to the reference's divide.
    fx = tx * g.inv_cell_size;
    fy = ty * g.inv_cell_size;
  }
  else
  {
    fx = tx / g.cell_size;
    fy = ty / g.cell_size;
  }
  const bool inside = (tx >= 0.0) & (ty >= 0.0) & (fx < static_cast<double>(g.size_x)) &
                      (fy < static_cast<double>(g.size_y));
  const uint32_t gx = static_cast<uint32_t>(fx);
  const uint32_t gy = static_cast<uint32_t>(fy);
  return inside ? gy * g.size_x + gx : g.ncell;
}

// Cell::score, reference src/ndt_model.cpp:105-116, on a packed record with
// h = -0.5 * information:  exponent = ((-0.5 q^T) I) q
//   = (q0*h00 + q1*h01) * q0 + (q0*h01 + q1*h11) * q1   (same roundings).
__device__ __forceinline__ double record_exponent(double mx, double my, double h00,
                                                  double h01, double h11, double px, double py)
{
  const double q0 = px - mx;
  const double q1 = py - my;
  const double r0 = q0 * h00 + q1 * h01;
  const double r1 = q0 * h01 + q1 * h11;
  return r0 * q0 + r1 * q1;
}

__device__ __forceinline__ double record_likelihood(double mx, double my, double h00,
                                                    double h01, double h11, double px,
                                                    double py)
{
  return exp(record_exponent(mx, my, h00, h01, h11, px, py));
}

// exp(x) for the exponents of Cell::score, x below the overflow threshold (x <= 0 for any
// positive semi-definite information matrix; exp_score() below handles the rest: NaN, and the
// positive exponents of an information matrix that rounding made negative definite).
// Same algorithm and polynomial as the device library's exp (n = rint(x log2 e),
// r = x - n ln2 in two pieces, degree-11 minimax polynomial, ldexp), minus its
// range fix-ups: the clamp maps -inf to an exponent whose ldexp underflows to
// +0.0, and v_ldexp_f64 rounds into the denormal range by itself.  NaN is NOT
// propagated (max() drops it): callers route NaN exponents to exp().
//
// The Horner steps are written as three-address v_fma_f64: left to itself the
// compiler prefers the two-address v_fmac_f64 and pays a v_mov_b64 per step to
// copy the polynomial constant into the accumulator (35 instead of 19 VALU
// instructions per exp in this register-starved kernel).
__device__ __forceinline__ double fma3(double a, double b, double c)
{
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// The polynomial constant as a template argument.  NDT2D_EXP_SCALAR_CONSTANTS=1 makes it in a
// scalar register pair right where it is used (two s_mov_b32, `volatile` so that they stay there)
// instead of the 18 VECTOR registers the compiler otherwise keeps the nine constants in for the
// whole kernel: the large search then needs 61 instead of 80 registers -- and is 2.5 % SLOWER
// (0.4720 against 0.4599 ms at cfg-2; eight waves per SIMD instead of six buy nothing either,
// 0.478 ms: the kernel is issue bound).  Measured, off.
#ifndef NDT2D_EXP_SCALAR_CONSTANTS
#define NDT2D_EXP_SCALAR_CONSTANTS 0
#endif
template <uint32_t HI, uint32_t LO>
__device__ __forceinline__ double fma3k(double a, double b)
{
  double d;
#if NDT2D_EXP_SCALAR_CONSTANTS
  uint32_t lo, hi;
  asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"(LO), "n"(HI));
  const double c = __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
#else
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(__hiloint2double(static_cast<int>(HI), static_cast<int>(LO))));
#endif
  return d;
}

__device__ __forceinline__ double exp_of_exponent(double x)
{
  // (one v_max_f64; fmax() would first canonicalise x with a second one)
  asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "s"(-1000.0));
  const double n = rint(x * 1.4426950408889634);             // 0x3ff71547652b82fe
  double r = fma(n, -0.6931471805599453, x);                  // 0xbfe62e42fefa39ef
  r = fma(n, -2.3190468138462996e-17, r);                     // 0xbc7abc9e3b39803f
  // p = c0; p = fma(r, p, c_k) ...: the first step's multiplier is the constant 0x1.ade156a5dcb37p-26
  double p = fma3k<0x3e928af3u, 0xfca7ab0cu>(r, 0x1.ade156a5dcb37p-26);   // + 0x1.28af3fca7ab0cp-22
  p = fma3k<0x3ec71deeu, 0x623fde64u>(r, p);   // + 0x1.71dee623fde64p-19
  p = fma3k<0x3efa0199u, 0x7c89e6b0u>(r, p);   // + 0x1.a01997c89e6b0p-16
  p = fma3k<0x3f2a01a0u, 0x14761f6eu>(r, p);   // + 0x1.a01a014761f6ep-13
  p = fma3k<0x3f56c16cu, 0x1852b7b0u>(r, p);   // + 0x1.6c16c1852b7b0p-10
  p = fma3k<0x3f811111u, 0x11122322u>(r, p);   // + 0x1.1111111122322p-7
  p = fma3k<0x3fa55555u, 0x555502a1u>(r, p);   // + 0x1.55555555502a1p-5
  p = fma3k<0x3fc55555u, 0x55555511u>(r, p);   // + 0x1.5555555555511p-3
  p = fma3k<0x3fe00000u, 0x0000000bu>(r, p);   // + 0x1.000000000000bp-1
  // (the last two constants are the inline operand 1.0: no register holds them)
  asm("v_fma_f64 %0, %1, %2, 1.0" : "=v"(p) : "v"(r), "v"(p));
  asm("v_fma_f64 %0, %1, %2, 1.0" : "=v"(p) : "v"(r), "v"(p));
  return ldexp(p, static_cast<int>(n));
}

// "Some lane of the wave holds p": the lane mask of the compare itself is tested in
// scalar registers.  (__any() first materialises p as 0 / 1 in a vector register
// and compares again: two VALU instructions per test in VALU-bound loops.)
__device__ __forceinline__ bool wave_any(bool p)
{
  return __builtin_amdgcn_ballot_w64(p) != 0ull;
}

// exp() of a Cell::score exponent for the register-tight kernels (lane mapping,
// compacted particle scoring; the wave mapping keeps the library exp, whose
// constants live in SGPRs there): the lean evaluation above, except that a NaN
// exponent (degenerate cell) propagates as in the reference.
//
// ... and an exponent above ln(DBL_MAX) gives +inf, as std::exp does (src/ndt_model.cpp:115).
// That happens: a cell whose >= 5 points are (nearly) identical -- a robot standing still in
// front of a distant wall -- has a covariance of rounding noise, which comes out NEGATIVE on
// both axes for one such cell in eight; Cell::compute's clamp branch (:88-96) then divides by
// det = 0.001 * large^2 > 0 and the information matrix is negative definite with entries of
// 1e19: exponents of +1e17, likelihood +inf, score -inf.  exp_of_exponent() is right up to the
// threshold itself (n = 1024, v_ldexp_f64 rounds p * 2^1024 < DBL_MAX correctly) and garbage
// above it (the reduced argument cancels away).  ONE compare serves both special cases.
constexpr double kExpOverflowAbove = 0x1.62e42fefa39efp+9;   // 709.782712893384: exp(x) = +inf for x > this (glibc, IEEE)

__device__ __forceinline__ double exp_score(double e)
{
  // A NaN goes through as the reference's exp(NaN) has it (quiet NaN); the wave-level test keeps
  // the selects off the common path.  (Round 4: this used to call the library's exp for such a
  // wave -- whose polynomial constants the compiler then kept in 20 vector registers for the
  // whole kernel, on behalf of a path no healthy map ever takes.)
  double r = exp_of_exponent(e);
  if (__builtin_expect(wave_any(!(e <= kExpOverflowAbove)), 0))
  {
    r = (e != e) ? e + e : (e > kExpOverflowAbove ? HUGE_VAL : r);
  }
  return r;
}

// A non-negative term t = exp(e) leaves a running sum s > 0 unchanged,
// RN(s + t) == s, whenever t < 2^-54 * s (less than half an ulp of s), i.e.
// e < ln(s) - 37.43.  negligible_below(s) returns a conservative such bound
// (ln(s) >= (frexp_exp(s) - 1) * ln 2); for s == 0 only the exp() underflow
// range (exp(e) == +0.0 for e < -745.14) qualifies.  Terms only grow the sum, so
// a bound taken earlier stays valid.
__device__ __forceinline__ double negligible_below(double s)
{
  const double from_sum =
    fma(static_cast<double>(__builtin_amdgcn_frexp_exp(s) - 1), 0.6931471805599453, -38.0);
  return (s > 0.0 && from_sum > -746.0) ? from_sum : -746.0;
}

// Exponent of Cell::score for (px, py) against the packed record idx of the
// LDS / HBM grid copy.
template <bool LDS_GRID>
__device__ __forceinline__ double indexed_exponent(const GridDesc & g, const double * lds_cells,
                                                   uint32_t idx, double px, double py)
{
  double2 a, b, c;
  if (LDS_GRID)
  {
    const double2 * rec = reinterpret_cast<const double2 *>(lds_cells + idx * kCellDoubles);
    a = rec[0];
    b = rec[1];
    c = rec[2];
  }
  else
  {
    const double2 * rec =
      reinterpret_cast<const double2 *>(g.cells_global + static_cast<size_t>(idx) * kCellStrideGlobal);
    a = rec[0];
    b = rec[1];
    c = rec[2];
  }
  return record_exponent(a.x, a.y, b.x, b.y, c.x, px, py);
}

template <bool LDS_GRID>
__device__ __forceinline__ double indexed_likelihood(const GridDesc & g, const double * lds_cells,
                                                     uint32_t idx, double px, double py)
{
  return exp(indexed_exponent<LDS_GRID>(g, lds_cells, idx, px, py));
}

// NDT::likelihood(Vector2d), reference src/ndt_model.cpp:162-170.
template <bool LDS_GRID, bool POW2>
__device__ __forceinline__ double point_likelihood(const GridDesc & g, const double * lds_cells,
                                                   double px, double py)
{
  return indexed_likelihood<LDS_GRID>(g, lds_cells, cell_index<POW2>(g, px, py), px, py);
}

__device__ __forceinline__ void stage_grid_to_lds(const GridDesc & g, double * lds_cells)
{
  const uint32_t n2 = (g.ncell + 1) * kCellDoubles / 2;  // kCellDoubles is even
  const double2 * src = reinterpret_cast<const double2 *>(g.cells_lds_image);
  double2 * dst = reinterpret_cast<double2 *>(lds_cells);
  for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x)
  {
    dst[i] = src[i];
  }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1)
  {
    v += __shfl_xor(v, off, kWave);
  }
  return v;
}

// Wave-wide reductions over the DPP lane network (no LDS traffic, fixed tree): the
// result is complete in lane 63.  Steps: the two quad permutes, row_half_mirror,
// row_mirror (a row = 16 lanes), then row_bcast15 into rows 1 and 3 and
// row_bcast31 into rows 2 and 3.
constexpr int kDppQuadXor1 = 0xb1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4e;      // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;

// the other lane's value; lanes of rows outside ROW_MASK get `fallback`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v, double fallback)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fallback), __double2loint(v), CTRL,
                                             ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fallback), __double2hiint(v), CTRL,
                                             ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_to_last_lane(double v)
{
  v += dpp_f64<kDppQuadXor1, 0xf>(v, 0.0);
  v += dpp_f64<kDppQuadXor2, 0xf>(v, 0.0);
  v += dpp_f64<kDppRowHalfMirror, 0xf>(v, 0.0);
  v += dpp_f64<kDppRowMirror, 0xf>(v, 0.0);
  v += dpp_f64<kDppRowBcast15, 0xa>(v, 0.0);
  v += dpp_f64<kDppRowBcast31, 0xc>(v, 0.0);
  return v;
}

// angles::shortest_angular_distance(from, to) = normalize_angle(to - from),
// normalize_angle(a) = fmod(a + pi, 2 pi) -/+ pi  (ROS `angles`, used at
// reference src/particle_filter.cpp:215)
__device__ __forceinline__ double shortest_angular_distance(double from, double to)
{
  const double kPi = 3.14159265358979323846;
  const double r = fmod((to - from) + kPi, 2.0 * kPi);
  return r <= 0.0 ? r + kPi : r - kPi;
}

// Results for a spinning host go to host-coherent memory with system-scope stores
// (write-through, no cache line kept), and the flag follows once they have been
// acknowledged: a __threadfence_system() instead would write back the whole L2 first.
__device__ __forceinline__ void store_host(double * p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// by ONE lane, after the block's store_host() calls and a barrier
__device__ __forceinline__ void raise_host_flag(double * flag_slot, unsigned long long seq)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(flag_slot), seq, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// A kernel whose bounded wait for its own blocks' results gave up (the last block of a launch
// polling the others' `done` words: ndt2d_match_small.hip, score_few_kernel) says so to the host
// instead of trapping -- a trap faults the whole HIP context and with it every matcher of the
// process: the flag leaves as seq | kHostFlagGaveUp, which the host's wait turns into
// NDT2D_ERR_HIP for THIS call only (ndt2d_device.hip wait_host_flag).
// (kHostFlagGaveUp itself: ndt2d_kernels.h, the host reads it too)
// The bound is TIME, read from the chip's 100 MHz wall clock (round 6; until then a count of
// poll trips, nominally two seconds -- but trips are not time: on a time-sliced or oversubscribed
// GPU a healthy launch could have run out of them).  Ten seconds in the product; the test build
// (NDT2D_TEST_HOOKS), whose tests force the poll to run out, waits half a second.  The clock is
// read once per 1,024 trips.
#ifdef NDT2D_TEST_HOOKS
constexpr unsigned long long kDonePollTicks = 50ull * 1000 * 1000;
#else
constexpr unsigned long long kDonePollTicks = 1000ull * 1000 * 1000;
#endif
struct BoundedPoll
{
  uint32_t trips = 0;
  unsigned long long since = 0;
  // one more trip: true when the wait has lasted longer than kDonePollTicks
  __device__ __forceinline__ bool expired()
  {
    if ((++trips & 1023u) != 0u) return false;
    const unsigned long long now = wall_clock64();
    if (since == 0)
    {
      since = now | 1ull;
      return false;
    }
    return now - since > kDonePollTicks;
  }
};

#ifdef NDT2D_TEST_HOOKS
// Test builds only (python -m ndt_2d_amd.build --test-hooks -> libndt2d_hip_hooks.so): the
// producer of record / pose `which - 1` withholds its `done` word (0: nobody does), so that the
// bounded polls can be made to trip (tests/test_gpu_bounded_poll.py).  One copy per translation unit.
static __device__ int g_test_drop_done = 0;
#define NDT2D_TEST_DROPS_DONE(index) (g_test_drop_done != 0 && static_cast<uint32_t>(g_test_drop_done - 1) == static_cast<uint32_t>(index))
#else
#define NDT2D_TEST_DROPS_DONE(index) false
#endif

// (score, index) ordering of the reference's `if (score < best_score)` scan in
// loop order (src/scan_matcher_ndt.cpp:128): lower score wins, ties go to the
// lower flat index.
__device__ __forceinline__ bool better(double s_a, double i_a, double s_b, double i_b)
{
  return (s_a < s_b) | ((s_a == s_b) & (i_a < i_b));
}

// Near-ties.  The kernels' scores differ from the reference's in the last bits (summation
// order within a candidate is the reference's, but exp() is the device's), so two candidates
// whose scores agree to within that rounding may come out in the other order than on the CPU.
// A score is a sum of at most a few thousand non-negative terms, each within ~2 ulp of the
// reference's: its RELATIVE error is below 1e-13, whatever its magnitude (scores of 1e-150 are
// as common as scores of -300 in degenerate maps -- an absolute tolerance would call all of the
// former ties).  Every merge of two (score, index) pairs therefore MARKS its winner when the loser
// was a real candidate within 2^-kNearTieLog2 (relative) of it:
// the mark is index + 0.5 (indices are exact integers in doubles; every consumer truncates, and the
// tie-break between different candidates is unaffected).  A marked winner makes the host collect
// the candidates that close to the best and rescore them with the reference's arithmetic
// (ndt2d_match_near_best, ndt2d_matcher_match_scan); an unmarked one proves that no other candidate
// lies that close: whatever loses to the final winner W within the tolerance (which is taken from
// the larger magnitude, W's) does so in a merge against W or against a candidate between the two,
// which then meets W within the tolerance itself -- and the mark travels with the winner.
// The tolerance is 2^-36 (1.46e-11) so that the test needs no constant in a register:
// |os - s| * 2^36 <= max(|os|, |s|), four instructions (add, max with |.| modifiers, ldexp, compare).
constexpr int kNearTieLog2 = 36;   // NDT2D_NEAR_TIE_REL = 2^-36

__device__ __forceinline__ void merge_best(double os, double oi, double & s, double & i)
{
  double larger;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(larger) : "v"(os), "v"(s));
  const bool near = (ldexp(fabs(os - s), kNearTieLog2) <= larger) & (oi < kNoIndex) & (i < kNoIndex);
  if (better(os, oi, s, i))
  {
    s = os;
    i = oi;
  }
  if (near) i = floor(i) + 0.5;
}

// (score, index) minimum of the wave in lane 63, same network.  A lane outside a
// step's row mask meets its own pair, which changes nothing.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void best_step(double & s, double & i)
{
  // (a lane outside the row mask meets its own pair: |s - s| = 0 would mark it -- the fallback
  // index is "no candidate" instead, which never marks and never wins)
  const double os = dpp_f64<CTRL, ROW_MASK>(s, s);
  const double oi = dpp_f64<CTRL, ROW_MASK>(i, kNoIndex);
  merge_best(os, oi, s, i);
}

__device__ __forceinline__ void wave_best_to_last_lane(double & s, double & i)
{
  best_step<kDppQuadXor1, 0xf>(s, i);
  best_step<kDppQuadXor2, 0xf>(s, i);
  best_step<kDppRowHalfMirror, 0xf>(s, i);
  best_step<kDppRowMirror, 0xf>(s, i);
  best_step<kDppRowBcast15, 0xa>(s, i);
  best_step<kDppRowBcast31, 0xc>(s, i);
}

}  // namespace ndt2d

#endif  // NDT2D_DEVICE_FN_H_
