// OccupancyGrid rendering on the device: the map the node
// publishes, ray-traced from every scan of the graph (reference
// src/occupancy_grid.cpp:47-185).  Integer work: one thread per beam walks the
// reference's simplified Bresenham line (:93-131) and counts with atomics; integer
// adds commute, so the counts -- and the published map -- are bit-identical to the
// sequential loop whatever the order.
//
//   bounds_kernel       updateBounds (:154-178): min / max of the scans' points in the
//                       map frame, per block, then one block folds the partials
//   rays_kernel         getMsg's ray loop (:73-131); `empty` and `hit` of a cell share
//                       one 64-bit counter (low / high word), one atomic per visit
//   finalize_kernel     (:134-150) -1 / 0 / 100 per cell
//
// HBM-bound on the counter traffic: a visit is one 8-byte L2 atomic; cells near a
// scan pose are shared by up to 720 beams of that scan.
#include "ndt2d_kernels.h"

#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

struct ScanRec
{
  double x, y, c, s;  // pose translation, cos / sin of pose theta (host libm, :78-79,163-164)
};

__device__ __forceinline__ uint32_t scan_of_point(const uint32_t * offsets, uint32_t n_scans,
                                                  uint32_t i)
{
  // last k with offsets[k] <= i
  uint32_t lo = 0, hi = n_scans;
  while (hi - lo > 1)
  {
    const uint32_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// partials[block] = {min_x, max_x, min_y, max_y} over the points [first_point, n_points)
__global__ void __launch_bounds__(256) bounds_kernel(const OccupancyArgs a, uint32_t first_point,
                                                     double * partials)
{
  __shared__ double sh[4][4];
  double mn_x = HUGE_VAL, mx_x = -HUGE_VAL, mn_y = HUGE_VAL, mx_y = -HUGE_VAL;
  for (uint32_t i = first_point + blockIdx.x * 256 + threadIdx.x; i < a.n_points;
       i += gridDim.x * 256)
  {
    const ScanRec sc = reinterpret_cast<const ScanRec *>(a.scans)[scan_of_point(a.offsets, a.n_scans, i)];
    const double2 p = reinterpret_cast<const double2 *>(a.points_xy)[i];
    // Point p(x, y); p.x += point.x * cos_th - point.y * sin_th (:171-173)
    const double px = sc.x + (p.x * sc.c - p.y * sc.s);
    const double py = sc.y + (p.x * sc.s + p.y * sc.c);
    mn_x = fmin(mn_x, px);
    mx_x = fmax(mx_x, px);
    mn_y = fmin(mn_y, py);
    mx_y = fmax(mx_y, py);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1)
  {
    mn_x = fmin(mn_x, __shfl_xor(mn_x, off, kWave));
    mx_x = fmax(mx_x, __shfl_xor(mx_x, off, kWave));
    mn_y = fmin(mn_y, __shfl_xor(mn_y, off, kWave));
    mx_y = fmax(mx_y, __shfl_xor(mx_y, off, kWave));
  }
  if ((threadIdx.x & (kWave - 1)) == 0)
  {
    const int w = threadIdx.x >> 6;
    sh[w][0] = mn_x;
    sh[w][1] = mx_x;
    sh[w][2] = mn_y;
    sh[w][3] = mx_y;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double * out = partials + static_cast<size_t>(blockIdx.x) * 4;
    out[0] = fmin(fmin(sh[0][0], sh[1][0]), fmin(sh[2][0], sh[3][0]));
    out[1] = fmax(fmax(sh[0][1], sh[1][1]), fmax(sh[2][1], sh[3][1]));
    out[2] = fmin(fmin(sh[0][2], sh[1][2]), fmin(sh[2][2], sh[3][2]));
    out[3] = fmax(fmax(sh[0][3], sh[1][3]), fmax(sh[2][3], sh[3][3]));
  }
}

__global__ void __launch_bounds__(256) bounds_reduce_kernel(const double * partials,
                                                            uint32_t n_blocks, double * out)
{
  __shared__ double sh[256][4];
  double v[4] = {HUGE_VAL, -HUGE_VAL, HUGE_VAL, -HUGE_VAL};
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256)
  {
    const double * p = partials + static_cast<size_t>(b) * 4;
    v[0] = fmin(v[0], p[0]);
    v[1] = fmax(v[1], p[1]);
    v[2] = fmin(v[2], p[2]);
    v[3] = fmax(v[3], p[3]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = v[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1)
  {
    if (static_cast<int>(threadIdx.x) < s)
    {
      sh[threadIdx.x][0] = fmin(sh[threadIdx.x][0], sh[threadIdx.x + s][0]);
      sh[threadIdx.x][1] = fmax(sh[threadIdx.x][1], sh[threadIdx.x + s][1]);
      sh[threadIdx.x][2] = fmin(sh[threadIdx.x][2], sh[threadIdx.x + s][2]);
      sh[threadIdx.x][3] = fmax(sh[threadIdx.x][3], sh[threadIdx.x + s][3]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) out[threadIdx.x] = sh[0][threadIdx.x];
}

// One beam per thread.  counts[cell] = hit << 32 | empty.
__global__ void __launch_bounds__(256) rays_kernel(const OccupancyArgs a,
                                                   unsigned long long * counts)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_points) return;
  const ScanRec sc = reinterpret_cast<const ScanRec *>(a.scans)[scan_of_point(a.offsets, a.n_scans, i)];
  const double2 p = reinterpret_cast<const double2 *>(a.points_xy)[i];
  // :82-83 (double -> int truncates toward zero)
  const int start_x = static_cast<int>((sc.x - a.origin_x) / a.resolution);
  const int start_y = static_cast<int>((sc.y - a.origin_y) / a.resolution);
  // :87-91
  const double point_x = p.x * sc.c - p.y * sc.s + sc.x;
  const double point_y = p.x * sc.s + p.y * sc.c + sc.y;
  const int end_x = static_cast<int>((point_x - a.origin_x) / a.resolution);
  const int end_y = static_cast<int>((point_y - a.origin_y) / a.resolution);
  // :93-98
  const int dx = abs(end_x - start_x);
  const int sx = (start_x < end_x) ? 1 : -1;
  const int dy = -abs(end_y - start_y);
  const int sy = (start_y < end_y) ? 1 : -1;
  int error = dx + dy;
  int x = start_x, y = start_y;
  const unsigned long long kHit = 1ull << 32, kEmpty = 1ull;
  while (true)
  {
    // a cell outside the grid is skipped (the reference would write out of bounds)
    const bool inside = x >= 0 && y >= 0 && static_cast<uint32_t>(x) < a.width &&
                        static_cast<uint32_t>(y) < a.height;
    unsigned long long * cell =
      counts + (inside ? static_cast<size_t>(x) + static_cast<size_t>(y) * a.width : 0);
    if (x == end_x && y == end_y)
    {
      if (inside) atomicAdd(cell, kHit);
      break;
    }
    unsigned long long add = kEmpty;
    bool done = false;
    if (2 * error >= dy)
    {
      if (x == end_x)
      {
        add += kHit;  // ++empty and ++hit of the same cell (:111,116)
        done = true;
      }
      else
      {
        error = error + dy;
        x += sx;
      }
    }
    if (!done && 2 * error <= dx)
    {
      if (y == end_y)
      {
        add += kHit;
        done = true;
      }
      else
      {
        error = error + dx;
        y += sy;
      }
    }
    if (inside) atomicAdd(cell, add);
    if (done) break;
  }
}

__global__ void __launch_bounds__(256) finalize_kernel(const unsigned long long * counts,
                                                       size_t n_cells, double occ_thresh,
                                                       signed char * data)
{
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_cells) return;
  const unsigned long long c = counts[i];
  const int hit = static_cast<int>(c >> 32), empty = static_cast<int>(c & 0xffffffffull);
  // :136-149
  const double touches = hit + empty;
  signed char v = -1;
  if (touches > 0.5) v = (static_cast<double>(hit) / touches > occ_thresh) ? 100 : 0;
  data[i] = v;
}

}  // namespace

hipError_t launch_occupancy_bounds(const OccupancyArgs & args, uint32_t first_point,
                                   double * workspace, double * bounds_out, hipStream_t stream)
{
  const uint32_t n = args.n_points - first_point;
  uint32_t blocks = (n + 255) / 256;
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(bounds_kernel, dim3(blocks), dim3(256), 0, stream, args, first_point, workspace);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bounds_reduce_kernel, dim3(1), dim3(256), 0, stream, workspace, blocks,
                     bounds_out);
  return hipGetLastError();
}

hipError_t launch_occupancy_render(const OccupancyArgs & args, double occ_thresh,
                                   unsigned long long * counts, signed char * data,
                                   hipStream_t stream)
{
  const size_t n_cells = static_cast<size_t>(args.width) * args.height;
  if (n_cells == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(counts, 0, n_cells * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  if (args.n_points > 0)
  {
    hipLaunchKernelGGL(rays_kernel, dim3((args.n_points + 255) / 256), dim3(256), 0, stream, args,
                       counts);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(static_cast<uint32_t>((n_cells + 255) / 256)), dim3(256), 0,
                     stream, counts, n_cells, occ_thresh, data);
  return hipGetLastError();
}

}  // namespace ndt2d
