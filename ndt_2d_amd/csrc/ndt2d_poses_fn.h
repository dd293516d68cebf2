// Shared by the two forms of the batched ScanMatcherNDT::scorePoints kernel (reference
// src/scan_matcher_ndt.cpp:156-178): the fused one (ndt2d_poses_compact.hip) and the
// screen / drain pair (ndt2d_poses_split.hip).
#pragma once

#include "ndt2d_device_fn.h"

namespace ndt2d
{

// The beams are cut into kChunks contiguous chunks.  A lane's score is
//   ((c_0 + c_1) + c_2) + ... + c_7,   c_j = in-order sum of the terms of chunk j,
// whatever number of waves shares the 64 poses of a group and whichever kernel form runs.
constexpr int kChunks = 8;

namespace
{

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The occupancy bitmap starts at LDS offset 0 (checked at kernel entry), so a word's
// byte offset is its LDS address: no per-look-up add of an array base.
__device__ __forceinline__ uint32_t lds_word_at(uint32_t address)
{
  typedef const __attribute__((address_space(3))) uint32_t * lds_word_ptr;
  return *reinterpret_cast<lds_word_ptr>(address);
}

constexpr uint32_t kBeamBits = 26;       // SCREEN queue word = beam index | lane << 26
constexpr uint32_t kScreenPadFloats = 64; // f32 beam array: one screening block (32 beams) of slack

// Bound (in cells) of |u_f32 - u| for the screening coordinate of a pose whose own
// cell coordinate is within `reach` cells of the grid: the four fused operations and
// the rounding of their FP32 inputs each contribute at most 2^-24 of the largest
// magnitude involved; 16x that, and never less than 2^-12 cell.
__host__ __device__ inline float screen_guard(float magnitude)
{
  const float g = magnitude * (16.0f / 16777216.0f);
  return g > (1.0f / 4096.0f) ? g : (1.0f / 4096.0f);
}

// The bitmap bit to read for the FP32 cell (iu, iv): iv * bits_sx + iu, or `bits_outside` (a
// bit that is never set).  ONE range test on the index instead of one per axis: a cell whose
// row is off the grid gives a negative or too large index (signed 24-bit multiply: the screening
// coordinates are bounded by 2^16) and is sent to `bits_outside`; a cell off the grid in u only
// reads a bit of the neighbouring row -- a false candidate now and then, which the exact
// NDT::getIndex of the drain phase answers with the reference's +0.0 (src/ndt_model.cpp:165-169).
// Two instructions (v_mad_i32_i24, v_min_u32) where the per-axis test costs four.
__device__ __forceinline__ uint32_t screen_bit_index(int iu, int iv, uint32_t k_log2, uint32_t bits_sx,
                                                     uint32_t bits_outside)
{
  const int idx = __mul24(iv >> k_log2, static_cast<int>(bits_sx)) + (iu >> k_log2);
  return min(static_cast<uint32_t>(idx), bits_outside);
}

// Inclusive prefix sum of one word per lane over the wave (DPP: a Kogge-Stone scan inside
// each row of 16 lanes, then the rows' totals handed on with row_bcast15 / row_bcast31).
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
  int x = static_cast<int>(v);
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, kDppRowBcast15, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, kDppRowBcast31, 0xc, 0xf, false);
  return static_cast<uint32_t>(x);
}

}  // namespace

}  // namespace ndt2d
