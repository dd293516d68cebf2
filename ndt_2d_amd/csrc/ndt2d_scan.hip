// LaserScan -> Scan conversion on the device: the loop of
// NdtMapper::laserCallback that projects every range into the robot frame and
// de-skews it by the odometry motion during the sweep (reference
// src/ndt_mapper.cpp:385-453), followed by matchScan's beam subsampling
// (reference src/scan_matcher_ndt.cpp:95-96,110), so that a scan enters the
// device as its raw float ranges (4 B/beam instead of 16 B/point) and its points
// and beams never visit the host.
//
// A LaserScan is ~10^3 ranges: one 1024-thread block walks it in chunks, keeps
// the reference's visiting order (descending index, index 0 never visited, when
// the laser is mounted inverted, :410) and compacts the kept points with a
// ballot / prefix count per wave and a running offset per chunk.
#include "ndt2d_kernels.h"

#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kScanThreads = 1024;
constexpr int kScanWaves = kScanThreads / kWave;

__device__ __forceinline__ double radius_bound(double x, double y)
{
  // an upper bound of |p| that is robust to the last ulps of hypot; NaN -> +inf
  const double r = hypot(x, y) * (1.0 + 1e-12);
  return isnan(r) ? HUGE_VAL : r;
}

// info_out = {number of points kept, max |point| (upper bound)}
__global__ void __launch_bounds__(kScanThreads) convert_scan_kernel(const float * ranges,
                                                                    uint32_t n_ranges,
                                                                    ScanDesc d,
                                                                    double * points_xy,
                                                                    double * info_out)
{
  __shared__ uint32_t sh_count[kScanWaves];
  __shared__ double sh_rmax[kScanWaves];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  // :391-394 trans_per_meas = translation / ranges.size()
  const double n_as_double = static_cast<double>(n_ranges);
  const double per_x = d.motion_x / n_as_double;
  const double per_y = d.motion_y / n_as_double;
  const double per_th = d.motion_theta / n_as_double;

  // inverted: positions j = 0 .. n-2 visit i = n-1 .. 1 (:410); else i = j (:433)
  const uint32_t n_visit = d.inverted ? (n_ranges > 0 ? n_ranges - 1 : 0) : n_ranges;
  uint32_t base = 0;
  double rmax = 0.0;
  for (uint32_t j0 = 0; j0 < n_visit; j0 += kScanThreads)
  {
    const uint32_t j = j0 + threadIdx.x;
    bool keep = false;
    double out_x = 0.0, out_y = 0.0;
    if (j < n_visit)
    {
      const uint32_t i = d.inverted ? (n_ranges - 1 - j) : j;
      const float range = ranges[i];
      // :413,436
      keep = !(isnan(range) || static_cast<double>(range) > d.range_max);
      if (keep)
      {
        // :415,438 float arithmetic (size_t * float -> float), then widened
        float a = d.angle_min + static_cast<float>(i) * d.angle_increment;
        if (d.inverted) a = -a;
        const double angle = static_cast<double>(a);
        double sa, ca;
        sincos(angle, &sa, &ca);
        const double r = static_cast<double>(range);
        const double lx = ca * r;
        const double ly = sa * r;
        // :419-420,442-443 laser frame -> robot frame
        const double px = d.cos_lt * lx - d.sin_lt * ly + d.laser_x;
        const double py = d.sin_lt * lx + d.cos_lt * ly + d.laser_y;
        // :422-426,445-448 motion of the robot while the sweep was taken
        const double di = static_cast<double>(i);
        double tt, tx, ty;
        if (d.inverted)
        {
          tt = d.motion_theta - (per_th * di);
          tx = d.motion_x - (per_x * di);
          ty = d.motion_y - (per_y * di);
        }
        else
        {
          tt = per_th * di;
          tx = per_x * di;
          ty = per_y * di;
        }
        double st, ct;
        sincos(tt, &st, &ct);
        out_x = ct * px - st * py + tx;
        out_y = st * px + ct * py + ty;
      }
    }
    // order-preserving compaction: rank inside the wave, then across waves
    const uint64_t mask = __ballot(keep);
    const uint32_t in_wave = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) sh_count[wave] = __popcll(mask);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kScanWaves; ++w)
    {
      const uint32_t c = sh_count[w];
      if (w < wave) before += c;
      total += c;
    }
    if (keep)
    {
      const uint32_t at = base + before + in_wave;
      points_xy[2 * static_cast<size_t>(at)] = out_x;
      points_xy[2 * static_cast<size_t>(at) + 1] = out_y;
      rmax = fmax(rmax, radius_bound(out_x, out_y));
    }
    base += total;
    __syncthreads();
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, off, kWave));
  if (lane == 0) sh_rmax[wave] = rmax;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double m = 0.0;
    for (int w = 0; w < kScanWaves; ++w) m = fmax(m, sh_rmax[w]);
    info_out[0] = static_cast<double>(base);
    info_out[1] = m;
  }
}

// matchScan / scorePoints subsampling (scan_matcher_ndt.cpp:95-96,110,164-165,170):
// use = min(max_beams, n), scan_step = double(n) / use, beam k = point[size_t(k * step)].
// n comes from the conversion's info record; info_out = {n, use, max |beam|}.
__global__ void __launch_bounds__(kScanThreads) subsample_kernel(const double * points_xy,
                                                                 const double * scan_info,
                                                                 uint32_t max_beams,
                                                                 double * beams_xy,
                                                                 double * info_out)
{
  __shared__ double sh_rmax[kScanWaves];
  const uint32_t n = static_cast<uint32_t>(scan_info[0]);
  const uint32_t use = max_beams < n ? max_beams : n;
  double rmax = 0.0;
  if (use > 0)
  {
    const double scan_step = static_cast<double>(n) / static_cast<double>(use);
    for (uint32_t k = threadIdx.x; k < use; k += kScanThreads)
    {
      const size_t idx = static_cast<size_t>(static_cast<double>(k) * scan_step);
      const double x = points_xy[2 * idx], y = points_xy[2 * idx + 1];
      beams_xy[2 * static_cast<size_t>(k)] = x;
      beams_xy[2 * static_cast<size_t>(k) + 1] = y;
      rmax = fmax(rmax, radius_bound(x, y));
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, off, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0) sh_rmax[threadIdx.x >> 6] = rmax;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double m = 0.0;
    for (int w = 0; w < kScanWaves; ++w) m = fmax(m, sh_rmax[w]);
    info_out[0] = static_cast<double>(n);
    info_out[1] = static_cast<double>(use);
    info_out[2] = m;
  }
}

}  // namespace

hipError_t launch_convert_scan(const float * ranges, uint32_t n_ranges, const ScanDesc & desc,
                               double * points_xy, double * info_out, hipStream_t stream)
{
  hipLaunchKernelGGL(convert_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, ranges, n_ranges,
                     desc, points_xy, info_out);
  return hipGetLastError();
}

hipError_t launch_subsample(const double * points_xy, const double * scan_info,
                            uint32_t max_beams, double * beams_xy, double * info_out,
                            hipStream_t stream)
{
  hipLaunchKernelGGL(subsample_kernel, dim3(1), dim3(kScanThreads), 0, stream, points_xy,
                     scan_info, max_beams, beams_xy, info_out);
  return hipGetLastError();
}

}  // namespace ndt2d
