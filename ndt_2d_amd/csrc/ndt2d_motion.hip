// Particle-filter steps either side of `measure`, so that particles stay resident
// in HBM across init -> update -> measure:
//   * MotionModel::sample            reference src/motion_model.cpp:45-83
//   * ParticleFilter::init           reference src/particle_filter.cpp:53-69
//   * the moment sums of ParticleFilter::updateStatistics for given weights
//                                    reference src/particle_filter.cpp:166-200
//
// The reference draws from std::normal_distribution<float> over an mt19937 seeded
// from std::random_device (motion_model.hpp:63-64), so its draws are not
// reproducible.  Here every particle owns a counter-based stream: Philox4x32-10
// keyed by `seed`, counter = (global particle index, step), four 32-bit words ->
// two Box-Muller pairs -> the three standard normals the reference draws per
// particle, in its order.  The stream depends only on (seed, step, index), so a
// shard of the particle set produces the numbers the whole set would.  The same
// device function backs noise_kernel, so a caller (and the parity tests) can read
// the exact normals a fused launch uses.
//
// All three kernels are one particle per lane, 24 B in / 24 B out: HBM-bound,
// coalesced through the stride-3 pose layout's three consecutive doubles.
#include "ndt2d_kernels.h"

#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

constexpr double kPi = 3.14159265358979323846;

// angles::normalize_angle (ROS `angles`): fmod(a + pi, 2 pi), then -/+ pi
__device__ __forceinline__ double normalize_angle(double a)
{
  const double r = fmod(a + kPi, 2.0 * kPi);
  return r <= 0.0 ? r + kPi : r - kPi;
}

struct Normals3
{
  float z0, z1, z2;
};

// Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = {index, step}.
__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t index, uint64_t step,
                                              uint32_t out[4])
{
  uint32_t c0 = static_cast<uint32_t>(index), c1 = static_cast<uint32_t>(index >> 32);
  uint32_t c2 = static_cast<uint32_t>(step), c3 = static_cast<uint32_t>(step >> 32);
  uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r)
  {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// 24 random bits -> (0, 1): every value and its complement are exact floats
__device__ __forceinline__ float unit_open(uint32_t bits)
{
  return (static_cast<float>(bits >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ Normals3 standard_normals(uint64_t seed, uint64_t index, uint64_t step)
{
  uint32_t w[4];
  philox4x32_10(seed, index, step, w);
  const float r0 = sqrtf(-2.0f * logf(unit_open(w[0])));
  const float r1 = sqrtf(-2.0f * logf(unit_open(w[2])));
  const float a0 = 6.28318530717958647692f * unit_open(w[1]);
  const float a1 = 6.28318530717958647692f * unit_open(w[3]);
  Normals3 z;
  z.z0 = r0 * cosf(a0);
  z.z1 = r0 * sinf(a0);
  z.z2 = r1 * cosf(a1);
  return z;
}

__device__ __forceinline__ Normals3 draw(const float * noise, uint64_t i, uint64_t seed,
                                         uint64_t first_index, uint64_t step)
{
  if (noise != nullptr)
  {
    Normals3 z;
    z.z0 = noise[3 * i];
    z.z1 = noise[3 * i + 1];
    z.z2 = noise[3 * i + 2];
    return z;
  }
  return standard_normals(seed, first_index + i, step);
}

__global__ void __launch_bounds__(256) noise_kernel(float * out, uint64_t n, uint64_t seed,
                                                    uint64_t first_index, uint64_t step)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const Normals3 z = standard_normals(seed, first_index + i, step);
    out[3 * i] = z.z0;
    out[3 * i + 1] = z.z1;
    out[3 * i + 2] = z.z2;
  }
}

// MotionModel::sample per-pose body (motion_model.cpp:74-83).  normal_distribution
// <float>(mean, sigma) holds its parameters as float and returns z * sigma + mean.
__global__ void __launch_bounds__(256) motion_kernel(double * poses_xyt, uint64_t n,
                                                     MotionParams p, const float * noise,
                                                     uint64_t seed, uint64_t first_index,
                                                     uint64_t step)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const Normals3 z = draw(noise, i, seed, first_index, step);
    const float r1 = z.z0 * p.sigma_rot1 + p.rot1;
    const float t = z.z1 * p.sigma_trans + p.trans;
    const float r2 = z.z2 * p.sigma_rot2 + p.rot2;
    const double x = poses_xyt[3 * i], y = poses_xyt[3 * i + 1], th = poses_xyt[3 * i + 2];
    const double heading = th + static_cast<double>(r1);
    double s, c;
    sincos(heading, &s, &c);
    poses_xyt[3 * i] = x + static_cast<double>(t) * c;
    poses_xyt[3 * i + 1] = y + static_cast<double>(t) * s;
    poses_xyt[3 * i + 2] = normalize_angle(heading + static_cast<double>(r2));
  }
}

// ParticleFilter::init sampling loop (particle_filter.cpp:60-65)
__global__ void __launch_bounds__(256) init_kernel(double * poses_xyt, uint64_t n, InitParams p,
                                                   const float * noise, uint64_t seed,
                                                   uint64_t first_index, uint64_t step)
{
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const Normals3 z = draw(noise, i, seed, first_index, step);
    const float x = z.z0 * p.sigma_x + p.x;
    const float y = z.z1 * p.sigma_y + p.y;
    const float th = z.z2 * p.sigma_theta + p.theta;
    poses_xyt[3 * i] = static_cast<double>(x);
    poses_xyt[3 * i + 1] = static_cast<double>(y);
    poses_xyt[3 * i + 2] = normalize_angle(static_cast<double>(th));
  }
}

// Per-block partial sums {w, w x, w y, w cos, w sin, w xx, w xy, w yy}
// (particle_filter.cpp:166-200) for given weights (uniform_weight when null).
__global__ void __launch_bounds__(256) moments_kernel(const double * poses_xyt, uint64_t n,
                                                      const double * weights,
                                                      double uniform_weight, double * partials)
{
  __shared__ double sh[4 * 8];
  double st[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) st[k] = 0.0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const double x = poses_xyt[3 * i], y = poses_xyt[3 * i + 1], th = poses_xyt[3 * i + 2];
    const double w = weights != nullptr ? weights[i] : uniform_weight;
    double s, c;
    sincos(th, &s, &c);
    st[0] += w;
    st[1] += w * x;
    st[2] += w * y;
    st[3] += w * c;
    st[4] += w * s;
    st[5] += w * x * x;
    st[6] += w * x * y;
    st[7] += w * y * y;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) st[k] = wave_sum(st[k]);
  if ((threadIdx.x & (kWave - 1)) == 0)
  {
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[(threadIdx.x >> 6) * 8 + k] = st[k];
  }
  __syncthreads();
  if (threadIdx.x < 8)
  {
    const int k = threadIdx.x;
    partials[static_cast<size_t>(blockIdx.x) * 8 + k] =
      ((sh[k] + sh[8 + k]) + sh[16 + k]) + sh[24 + k];
  }
}

__global__ void __launch_bounds__(256) moments_reduce_kernel(const double * partials,
                                                             uint32_t n_blocks, double * stats)
{
  __shared__ double sh[256 * 8];
  const int t = threadIdx.x;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.0;
  for (uint32_t b = t; b < n_blocks; b += 256)
  {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += partials[static_cast<size_t>(b) * 8 + k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[t * 8 + k] = v[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1)
  {
    if (t < s)
    {
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[t * 8 + k] += sh[(t + s) * 8 + k];
    }
    __syncthreads();
  }
  if (t < 8) stats[t] = sh[t];
}

uint32_t stream_blocks(uint64_t n)
{
  const uint64_t need = (n + 255) / 256;
  return static_cast<uint32_t>(need < kMaxStreamBlocks ? need : kMaxStreamBlocks);
}

}  // namespace

hipError_t launch_pf_noise(float * noise_out, uint64_t n, uint64_t seed, uint64_t first_index,
                           uint64_t step, hipStream_t stream)
{
  if (n == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(noise_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, noise_out, n,
                     seed, first_index, step);
  return hipGetLastError();
}

hipError_t launch_pf_motion(double * poses_xyt, uint64_t n, const MotionParams & params,
                            const float * noise, uint64_t seed, uint64_t first_index,
                            uint64_t step, hipStream_t stream)
{
  if (n == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(motion_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, poses_xyt, n,
                     params, noise, seed, first_index, step);
  return hipGetLastError();
}

hipError_t launch_pf_init(double * poses_xyt, uint64_t n, const InitParams & params,
                          const float * noise, uint64_t seed, uint64_t first_index,
                          uint64_t step, hipStream_t stream)
{
  if (n == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(init_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, poses_xyt, n,
                     params, noise, seed, first_index, step);
  return hipGetLastError();
}

hipError_t launch_pose_moments(const double * poses_xyt, uint64_t n, const double * weights,
                               double * workspace, double * stats_out, hipStream_t stream)
{
  if (n == 0) return hipErrorInvalidValue;
  const uint32_t blocks = stream_blocks(n);
  hipLaunchKernelGGL(moments_kernel, dim3(blocks), dim3(256), 0, stream, poses_xyt, n, weights,
                     1.0 / static_cast<double>(n), workspace);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3(1), dim3(256), 0, stream, workspace, blocks,
                     stats_out);
  return hipGetLastError();
}

}  // namespace ndt2d
