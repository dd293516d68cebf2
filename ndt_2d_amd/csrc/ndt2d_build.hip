// NDT build on the device: ScanMatcherNDT::addScans' NDT::addScan + NDT::compute
// (reference src/scan_matcher_ndt.cpp:67-73, src/ndt_model.cpp:50-103,132-160).
//
// Cell::addPoint is an order-dependent recurrence (incremental mean and second
// moment, src/ndt_model.cpp:52-57) whose later `correlation - mean^2`
// (src/ndt_model.cpp:78) cancels catastrophically, so bit parity needs every cell to see its
// points in the reference's order (scan by scan, point by point).  Pipeline:
//
//   1. points_kernel   one thread per point: transform by its scan's pose
//                      (:139-141, cos/sin from the host libm as in :135-136),
//                      NDT::getIndex (:203-218) -> sort key = cell (ncell = outside)
//   2. stable radix sort of (cell, point index)  [rocprim::radix_sort_pairs, onesweep]: points
//                      of one cell stay in their original order
//   3. segments_kernel first / one-past-last sorted position of every cell
//   4. cell_sums_kernel + cells_kernel   a lane per (cell, quantity) walks the cell's points in order: addPoint,
//                      then Cell::compute (:65-103), and writes the cell in all the
//                      layouts the scorers read (cells6, LDS image, 64-byte gather
//                      copy)
//      (the occupancy bitmap comes from the waves' ballots), then cell_bytes_kernel: the
//      per-cell occupancy-map bytes the small-lattice search copies
//
// Everything is IEEE double with the reference's operation order (file compiled
// with -ffp-contract=off; '/' and sqrt are correctly rounded).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "ndt2d_eigen2.h"
#include "ndt2d_lane_fn.h"

namespace ndt2d
{

namespace
{

struct ScanRec
{
  double x, y, c, s;  // pose translation, cos / sin of pose theta
};

__global__ void __launch_bounds__(256) points_kernel(const BuildArgs a)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_points) return;
  // which scan does point i belong to: last k with offsets[k] <= i
  uint32_t lo = 0, hi = a.n_scans;
  while (hi - lo > 1)
  {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.offsets[mid] <= i) lo = mid; else hi = mid;
  }
  const ScanRec sc = reinterpret_cast<const ScanRec *>(a.scans)[lo];
  const double2 p = reinterpret_cast<const double2 *>(a.points_xy)[i];
  // p(0) = pose.x; p(0) += point.x * cos_th - point.y * sin_th (:139-141)
  const double wx = sc.x + (p.x * sc.c - p.y * sc.s);
  const double wy = sc.y + (p.x * sc.s + p.y * sc.c);
  a.world_xy[2 * static_cast<size_t>(i)] = wx;
  a.world_xy[2 * static_cast<size_t>(i) + 1] = wy;
  a.keys_in[i] = a.grid.pow2 ? cell_index<true>(a.grid, wx, wy) : cell_index<false>(a.grid, wx, wy);
  a.vals_in[i] = i;
}

__global__ void __launch_bounds__(256) segments_kernel(const uint32_t * keys, uint32_t n,
                                                       uint32_t * seg_begin, uint32_t * seg_end)
{
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t k = keys[j];
  if (j == 0 || keys[j - 1] != k) seg_begin[k] = j;
  if (j == n - 1 || keys[j + 1] != k) seg_end[k] = j + 1;
}

// One packed record in both strides, and -- from the wave's ballot -- the two words of
// the occupancy bitmap its 64 cells fill (bit i = cell i can score; bits >= ncell are 0).
// Called by every thread of the wave, also those past the last record (cell > ncell).
__device__ __forceinline__ void write_scorer_record(uint32_t ncell, uint32_t cell,
                                                    const double (&rec)[kCellDoubles],
                                                    double * cells_lds_image, double * cells_global,
                                                    uint32_t * occ_bits)
{
  if (cell <= ncell)
  {
    double * l = cells_lds_image + static_cast<size_t>(cell) * kCellDoubles;
    double * gl = cells_global + static_cast<size_t>(cell) * kCellStrideGlobal;
#pragma unroll
    for (int k = 0; k < kCellDoubles; ++k)
    {
      l[k] = rec[k];
      gl[k] = rec[k];
    }
    gl[6] = 0.0;
    gl[7] = 0.0;
  }
  const uint64_t mask = __builtin_amdgcn_ballot_w64(cell < ncell && rec[5] != 0.0);
  const uint32_t n_words = (ncell + 1 + 31) / 32;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t word = (cell >> 5) & ~1u;     // first word of this wave's 64 cells
  if (lane == 0 && word < n_words) occ_bits[word] = static_cast<uint32_t>(mask);
  if (lane == 1 && word + 1 < n_words) occ_bits[word + 1] = static_cast<uint32_t>(mask >> 32);
}

// Cell::addPoint in the reference's point order (src/ndt_model.cpp:50-63): five recurrences per
// cell -- the mean's two components and the upper triangle of the second moment, each
// v = (v * n + t) / (n + 1) with its own t -- that share nothing but n.  The recurrence is a
// chain (a wall cell seen from hundreds of scans holds hundreds of points) and with a lane per
// CELL that lane issues all five, ~135 FP64 instructions per point, less than one wave to a SIMD.
// So: a lane per (cell, quantity), eight lanes to a cell of which five work -- the same
// arithmetic on every quantity, a fifth of the instructions on the longest chain.  The sums
// are left in cells6[cell] = {mean_x, mean_y, cxx, cxy, cyy, n} for cells_kernel below.
__global__ void __launch_bounds__(256) cell_sums_kernel(const BuildArgs a, const uint32_t * sorted_vals,
                                                        const uint32_t * seg_begin,
                                                        const uint32_t * seg_end)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t cell = t >> 3, q = t & 7u;
  if (cell >= a.grid.ncell || q >= 5u) return;
  const uint32_t b = seg_begin[cell], e = seg_end[cell];
  const double2 * xy = reinterpret_cast<const double2 *>(a.world_xy);
  double n = 0.0, v = 0.0;
  auto add_point = [&](double2 p) {
    const double term = q == 0u ? p.x : q == 1u ? p.y : q == 2u ? p.x * p.x : q == 3u ? p.x * p.y : p.y * p.y;
    v = (v * n + term) / (n + 1);
    n += 1;
  };
  // (the two gathers per point -- its index, then the point -- eight points ahead of the chain)
  constexpr uint32_t kBatch = 8;
  uint32_t j = b;
  for (; j + kBatch <= e; j += kBatch)
  {
    uint32_t idx[kBatch];
    double2 p[kBatch];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) idx[u] = sorted_vals[j + u];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) p[u] = xy[idx[u]];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) add_point(p[u]);
  }
  for (; j < e; ++j) add_point(xy[sorted_vals[j]]);
  double * c6 = a.cells6 + static_cast<size_t>(cell) * 6;
  c6[q] = v;
  if (q == 0u) c6[5] = n;
}

// Cell::compute (src/ndt_model.cpp:65-103), a lane per cell.  OWN_SUMS: the lane also walks its
// cell's points itself (all five recurrences: the form for sparse grids -- cfg-5's 641,601 cells
// hold 1.7 points each on average, and eight lanes per cell would be five million threads with
// nothing to do); otherwise from the sums cell_sums_kernel left in cells6.
template <bool OWN_SUMS>
__global__ void __launch_bounds__(256) cells_kernel(const BuildArgs a, const uint32_t * sorted_vals,
                                                    const uint32_t * seg_begin,
                                                    const uint32_t * seg_end)
{
  const uint32_t cell = blockIdx.x * 256 + threadIdx.x;
  // (threads past the last record stay: the wave's ballot writes its occupancy words)

  double n = 0.0, mean_x = 0.0, mean_y = 0.0, cxx = 0.0, cxy = 0.0, cyy = 0.0;
  double ixx = 0.0, ixy = 0.0, iyy = 0.0;
  if (cell < a.grid.ncell)
  {
    if (OWN_SUMS)
    {
      const uint32_t b = seg_begin[cell], e = seg_end[cell];
      const double2 * xy = reinterpret_cast<const double2 *>(a.world_xy);
      auto add_point = [&](double2 p) {
        const double n1 = n + 1;
        mean_x = (mean_x * n + p.x) / n1;
        mean_y = (mean_y * n + p.y) / n1;
        cxx = (cxx * n + p.x * p.x) / n1;
        cxy = (cxy * n + p.x * p.y) / n1;
        cyy = (cyy * n + p.y * p.y) / n1;
        n += 1;
      };
      constexpr uint32_t kBatch = 8;
      uint32_t j = b;
      for (; j + kBatch <= e; j += kBatch)
      {
        uint32_t idx[kBatch];
        double2 p[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) idx[u] = sorted_vals[j + u];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) p[u] = xy[idx[u]];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) add_point(p[u]);
      }
      for (; j < e; ++j) add_point(xy[sorted_vals[j]]);
    }
    else
    {
      const double * s6 = a.cells6 + static_cast<size_t>(cell) * 6;
      mean_x = s6[0];
      mean_y = s6[1];
      cxx = s6[2];
      cxy = s6[3];
      cyy = s6[4];
      n = s6[5];
    }
    // Cell::compute (src/ndt_model.cpp:65-103)
    if (!(n < 3))
    {
      const double scale = n / (n - 1);
      const double vxx = (cxx - (mean_x * mean_x)) * scale;
      const double vxy = (cxy - (mean_x * mean_y)) * scale;
      const double vyy = (cyy - (mean_y * mean_y)) * scale;
      double small = 1.0, large = 1.0;
      if (!clamp_test_surely_false(vxx, vxy, vyy))   // (else: the branch of :99 whatever their last bits)
      {
        covariance_eigenvalues(a.eigen_form, vxx, vxy, vyy, &small, &large);   // (:84-85, ndt2d_eigen2.h)
        if (small > large)
        {
          const double t = small;
          small = large;
          large = t;
        }
      }
      if (small < 0.001 * large)
      {
        const double determinant = (0.001 * large) * large;
        ixx = vyy / determinant;
        ixy = -vxy / determinant;
        iyy = vxx / determinant;
      }
      else
      {
        const double det = vxx * vyy - vxy * vxy;
        const double invdet = 1.0 / det;
        ixx = vyy * invdet;
        ixy = -vxy * invdet;
        iyy = vxx * invdet;
      }
    }
    double * c6 = a.cells6 + static_cast<size_t>(cell) * 6;
    c6[0] = mean_x;
    c6[1] = mean_y;
    c6[2] = ixx;
    c6[3] = ixy;
    c6[4] = iyy;
    c6[5] = n;
  }

  // scorer layouts: h = -0.5 * information; sentinel for cells that cannot score
  // (n < 5, src/ndt_model.cpp:107) and for record ncell ("outside")
  double rec[kCellDoubles] = {1.0e300, 0.0, -1.0, 0.0, -1.0, 0.0};
  if (cell < a.grid.ncell && !(n < 5.0))
  {
    rec[0] = mean_x;
    rec[1] = mean_y;
    rec[2] = -0.5 * ixx;
    rec[3] = -0.5 * ixy;
    rec[4] = -0.5 * iyy;
    rec[5] = 1.0;
  }
  write_scorer_record(a.grid.ncell, cell, rec, a.cells_lds_image, a.cells_global, a.occ_bits);
}

__global__ void __launch_bounds__(256) cell_bytes_kernel(const GridDesc g, uint8_t * bytes)
{
  cell_byte_row<false>(g, nullptr, (blockIdx.x * 256 + threadIdx.x) >> 4, threadIdx.x & 15u, bytes);
}

// Everything a host-installed grid needs, in ONE launch (the mapper installs a grid per
// scan): the first pack_blocks blocks write the packed records and the occupancy bitmap,
// the others the per-cell map bytes -- straight from the cells6 records, so that they do
// not have to wait for the first.
__global__ void __launch_bounds__(256) pack_grid_kernel(const GridDesc g, const double * cells6,
                                                        uint32_t pack_blocks, double * cells_lds_image,
                                                        double * cells_global, uint32_t * occ_bits,
                                                        uint8_t * bytes)
{
  if (blockIdx.x >= pack_blocks)
  {
    cell_byte_row<true>(g, cells6, ((blockIdx.x - pack_blocks) * 256 + threadIdx.x) >> 4,
                        threadIdx.x & 15u, bytes);
    return;
  }
  const uint32_t cell = blockIdx.x * 256 + threadIdx.x;
  // h = -0.5 * information (exact); sentinel for cells that cannot score (n < 5,
  // src/ndt_model.cpp:107) and for record ncell ("outside")
  double rec[kCellDoubles] = {1.0e300, 0.0, -1.0, 0.0, -1.0, 0.0};
  if (cell < g.ncell)
  {
    const double * c = cells6 + static_cast<size_t>(cell) * 6;
    if (!(c[5] < 5.0))
    {
      rec[0] = c[0];
      rec[1] = c[1];
      rec[2] = -0.5 * c[2];
      rec[3] = -0.5 * c[3];
      rec[4] = -0.5 * c[4];
      rec[5] = 1.0;
    }
  }
  write_scorer_record(g.ncell, cell, rec, cells_lds_image, cells_global, occ_bits);
}

// ---- a grid given as the LIST of its cells that hold points (ndt2d_set_grid_sparse) ----
//
// The mapper installs a new local NDT for every scan (reference src/ndt_mapper.cpp:508-509),
// and with a real lidar that grid is tens of thousands of cells (scan poses +- range_max,
// src/scan_matcher_ndt.cpp:52-66: 245 x 245 at 30 m / 0.25 m) of which the scans' points touch
// one or two thousand.  Dense, every install moved and packed all of them (2.9 MB over PCIe);
// here the host sends the touched cells only and two small kernels produce the layouts the
// scorers read: install (every listed cell its packed record, every other cell the "cannot
// score" sentinel) and bytes (the map bytes of the cells around the listed ones).

// fill + scatter in ONE launch that reads the host's staged image in place (over PCIe, each
// byte once) and leaves its device copy behind: no copy command in front of it -- the copy
// and the dependency behind it were a third of what the mapper's per-scan install kept the
// stream busy for.  Nothing here is ordered against anything else in the launch: the host's
// occupancy words say which cells the list will write, and the fill leaves those alone.
//   listed cells, 256 per step: their records through LDS with 16-byte loads (a lane reading
//   its own 48 bytes would fetch every line three times), packed into both strides;
//   all other cells, 1024 per step: the sentinel, rank n_occ; the occupancy words copied;
//   the compacted records copied; the map bytes zeroed (grid_bytes_sparse_kernel follows).
__global__ void __launch_bounds__(256) grid_install_kernel(const GridDesc g, const SparseImage im,
                                                           double * cells_lds_image, double * cells_global,
                                                           uint32_t * occ_bits, uint8_t * bytes,
                                                           uint16_t * ranks, uint32_t n_occ)
{
  __shared__ __align__(16) double sh_rec[256 * 6];
  __shared__ uint32_t sh_words[32];
  const uint32_t tid = threadIdx.x;
  const uint32_t * src_idx = reinterpret_cast<const uint32_t *>(im.src + im.off_idx);
  const uint16_t * src_rk = reinterpret_cast<const uint16_t *>(im.src + im.off_rk);
  const uint32_t * src_occ = reinterpret_cast<const uint32_t *>(im.src + im.off_occ);
  uint32_t * dst_idx = im.dst != nullptr ? reinterpret_cast<uint32_t *>(im.dst + im.off_idx) : nullptr;
  uint16_t * dst_rk = im.dst != nullptr ? reinterpret_cast<uint16_t *>(im.dst + im.off_rk) : nullptr;

  // Reads of the staged image cross PCIe (a couple of microseconds each way): everything a
  // step needs is requested before anything is waited for -- the records' pieces, the cell
  // index and rank, and, ahead of the first step, the block's first occupancy word.
  const uint32_t n_words = (g.ncell + 1 + 31) / 32;
  const uint32_t fill_chunks = (n_words + 31) / 32;
  // (unconditional loads from clamped addresses: behind a branch the compiler waits for the
  // value where the branches join, i.e. before the next request goes out)
  const uint32_t word_index = blockIdx.x * 32 + (tid & 31u);
  const uint32_t first_word = src_occ[word_index < n_words ? word_index : n_words - 1];
  const uint32_t stride = gridDim.x * 256;
  const size_t first_piece = blockIdx.x * 256 + tid;     // of the compacted records' copy (the last step)
  const size_t n_pieces = im.n_compact / 2;
  const double2 first_compact =
    reinterpret_cast<const double2 *>(im.src + im.off_compact)[first_piece < n_pieces ? first_piece : 0];

  const uint32_t list_chunks = (im.n + 255) / 256;
  const size_t d_end = static_cast<size_t>(im.n) * 6;
  for (uint32_t chunk = blockIdx.x; chunk < list_chunks; chunk += gridDim.x)
  {
    const size_t d0 = static_cast<size_t>(chunk) * (256 * 6);
    const uint32_t k = chunk * 256 + tid;
    double2 v[3];
#pragma unroll
    for (uint32_t p = 0; p < 3; ++p)
    {
      const size_t d = d0 + 2 * static_cast<size_t>(p * 256 + tid);
      v[p] = d < d_end ? *reinterpret_cast<const double2 *>(im.src + d) : double2{0.0, 0.0};
    }
    const uint32_t cell = k < im.n ? src_idx[k] : 0u;
    const uint16_t rk = k < im.n ? src_rk[k] : static_cast<uint16_t>(0);
#pragma unroll
    for (uint32_t p = 0; p < 3; ++p)
    {
      const uint32_t piece = p * 256 + tid;
      const size_t d = d0 + 2 * static_cast<size_t>(piece);
      *reinterpret_cast<double2 *>(sh_rec + 2 * piece) = v[p];
      if (im.dst != nullptr && d < d_end) *reinterpret_cast<double2 *>(im.dst + d) = v[p];
    }
    __syncthreads();
    if (k < im.n)
    {
      if (im.dst != nullptr)
      {
        dst_idx[k] = cell;
        dst_rk[k] = rk;
      }
      const double * c = sh_rec + tid * 6;
      // (a listed cell that cannot score -- n < 5, src/ndt_model.cpp:107 -- has no occupancy
      // bit: the fill below writes its sentinel)
      if (cell < g.ncell && !(c[5] < 5.0))
      {
        const double2 r0 = {c[0], c[1]}, r1 = {-0.5 * c[2], -0.5 * c[3]}, r2 = {-0.5 * c[4], 1.0};
        double2 * l = reinterpret_cast<double2 *>(cells_lds_image + static_cast<size_t>(cell) * kCellDoubles);
        double2 * gl = reinterpret_cast<double2 *>(cells_global + static_cast<size_t>(cell) * kCellStrideGlobal);
        l[0] = r0;
        l[1] = r1;
        l[2] = r2;
        gl[0] = r0;
        gl[1] = r1;
        gl[2] = r2;
        if (ranks != nullptr) ranks[cell] = rk;
      }
    }
    __syncthreads();
  }

  for (uint32_t chunk = blockIdx.x; chunk < fill_chunks; chunk += gridDim.x)
  {
    if (tid < 32)
    {
      const uint32_t w = chunk * 32 + tid;
      const uint32_t v = w >= n_words ? 0u : (chunk == blockIdx.x ? first_word : src_occ[w]);
      sh_words[tid] = v;
      if (w < n_words) occ_bits[w] = v;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
    {
      const uint32_t local = j * 256 + tid;
      const uint32_t cell = chunk * 1024 + local;
      if (cell <= g.ncell && !((sh_words[local >> 5] >> (local & 31u)) & 1u))
      {
        double2 * l = reinterpret_cast<double2 *>(cells_lds_image + static_cast<size_t>(cell) * kCellDoubles);
        double2 * gl = reinterpret_cast<double2 *>(cells_global + static_cast<size_t>(cell) * kCellStrideGlobal);
        const double2 a = {1.0e300, 0.0}, b = {-1.0, 0.0}, z = {0.0, 0.0};
        l[0] = a;
        l[1] = b;
        l[2] = b;
        gl[0] = a;
        gl[1] = b;
        gl[2] = b;
        gl[3] = z;
        if (ranks != nullptr) ranks[cell] = static_cast<uint16_t>(n_occ);
      }
    }
    __syncthreads();
  }

  if (im.dst != nullptr)
  {
    // (the launch has a block per 256 pieces, so the loop behind the first piece is idle)
    if (first_piece < n_pieces) reinterpret_cast<double2 *>(im.dst + im.off_compact)[first_piece] = first_compact;
    for (size_t p = first_piece + stride; p < n_pieces; p += stride)
    {
      reinterpret_cast<double2 *>(im.dst + im.off_compact)[p] =
        reinterpret_cast<const double2 *>(im.src + im.off_compact)[p];
    }
  }
  // (whole words; the byte array is allocated in doubles)
  const uint32_t n_bytes = (g.size_x + 2) * (g.size_y + 2);
  uint32_t * b32 = reinterpret_cast<uint32_t *>(bytes);
  for (uint32_t i = blockIdx.x * 256 + tid; i < (n_bytes + 3) / 4; i += stride) b32[i] = 0u;
}

// The map bytes around the listed cells (sparse_byte_rows, ndt2d_lane_fn.h) as a launch of its own.
__global__ void __launch_bounds__(256) grid_bytes_sparse_kernel(const GridDesc g, const SparseBytesJob job)
{
  sparse_byte_rows(g, job, blockIdx.x * 256 + threadIdx.x);
}

// Dense raw records of a sparse grid (ndt2d_get_grid): zeroed by the caller, the listed cells here.
__global__ void __launch_bounds__(256) grid_scatter_raw_kernel(const uint32_t * cell_index, const double * cells6,
                                                               uint32_t n, uint32_t ncell, double * dense6)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n || cell_index[k] >= ncell) return;
#pragma unroll
  for (int j = 0; j < 6; ++j) dense6[static_cast<size_t>(cell_index[k]) * 6 + j] = cells6[static_cast<size_t>(k) * 6 + j];
}

int key_bits(uint32_t ncell)
{
  int b = 1;
  while (b < 32 && (1ull << b) <= ncell) ++b;
  return b;
}

}  // namespace

// The compacted form of a device-built grid (what the host install of a small grid prepares on
// the CPU, ndt2d_device.hip ndt2d_set_grid): the records of the cells that can score, in cell
// order, followed by the sentinel's, and the cell -> record table (uint16; cells that cannot score
// and entry ncell, "outside", point at the sentinel).  The searches keep both in LDS -- 15 KB
// instead of 81 at cfg-2, two blocks to a CU -- and were 8-20 % slower on a device-built grid
// (a loop closure's addScans of many scans) for want of them.  One block: the occupancy bitmap's
// words (bit = the cell can score, written by cells_kernel's ballots) are counted and scanned in
// LDS, a cell's rank is its word's prefix plus the set bits below it.  ncell < 65,535.
constexpr int kCompactThreads = 1024;
__global__ void __launch_bounds__(kCompactThreads) compact_grid_kernel(uint32_t ncell, const double * cells_lds_image,
                                                                       const uint32_t * occ_bits, double * records,
                                                                       uint16_t * ranks, uint32_t * n_occ_out)
{
  __shared__ uint32_t prefix[2048 + 1];
  __shared__ uint32_t wave_total[kCompactThreads / 64];
  const uint32_t n_words = (ncell + 31) / 32;           // words that hold cells (<= 2048)
  const uint32_t t = threadIdx.x;
  // two consecutive words per thread: their counts, scanned over the block
  uint32_t w0 = 0, w1 = 0;
  if (2 * t < n_words) w0 = occ_bits[2 * t];
  if (2 * t + 1 < n_words) w1 = occ_bits[2 * t + 1];
  // (bits at and beyond ncell are zero: write_scorer_record's ballot tests cell < ncell)
  const uint32_t c0 = static_cast<uint32_t>(__builtin_popcount(w0)), c1 = static_cast<uint32_t>(__builtin_popcount(w1));
  uint32_t incl = c0 + c1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const uint32_t up = __shfl_up(incl, off, 64);
    if ((t & 63u) >= static_cast<uint32_t>(off)) incl += up;
  }
  if ((t & 63u) == 63u) wave_total[t >> 6] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < (t >> 6); ++w) base += wave_total[w];
  const uint32_t excl = base + incl - (c0 + c1);
  prefix[2 * t] = excl;
  prefix[2 * t + 1] = excl + c0;
  if (t == kCompactThreads - 1) prefix[2048] = base + incl;
  __syncthreads();
  const uint32_t n_occ = prefix[2048];
  for (uint32_t cell = t; cell < ncell; cell += kCompactThreads)
  {
    const uint32_t word = occ_bits[cell >> 5], bit = cell & 31u;
    const bool occ = ((word >> bit) & 1u) != 0u;
    const uint32_t rank = occ ? prefix[cell >> 5] + static_cast<uint32_t>(__builtin_popcount(word & ((1u << bit) - 1u))) : n_occ;
    ranks[cell] = static_cast<uint16_t>(rank);
    if (occ)
    {
      const double * src = cells_lds_image + static_cast<size_t>(cell) * kCellDoubles;
      double * dst = records + static_cast<size_t>(rank) * kCellDoubles;
#pragma unroll
      for (int k = 0; k < kCellDoubles; ++k) dst[k] = src[k];
    }
  }
  if (t == 0)
  {
    ranks[ncell] = static_cast<uint16_t>(n_occ);
    const double * src = cells_lds_image + static_cast<size_t>(ncell) * kCellDoubles;   // the sentinel record
    double * dst = records + static_cast<size_t>(n_occ) * kCellDoubles;
#pragma unroll
    for (int k = 0; k < kCellDoubles; ++k) dst[k] = src[k];
    *n_occ_out = n_occ;
  }
}

// The library's radix sort hands anything below a million keys to its merge sort -- 21 launches
// of ~6 us each for the 378,000 points of cfg-3's map, a third of the whole device build.  With the
// limit lowered the one-sweep radix sort takes them: a histogram, its scan and one pass per 8 key
// bits (the keys are cell indices: 16 bits at cfg-3, 20 at cfg-5).  Both are stable.
using BuildSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 16384>;

size_t build_sort_temp_bytes(uint32_t n_points, uint32_t ncell)
{
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<BuildSortConfig>(nullptr, bytes, static_cast<const uint32_t *>(nullptr),
                                                   static_cast<uint32_t *>(nullptr),
                                                   static_cast<const uint32_t *>(nullptr),
                                                   static_cast<uint32_t *>(nullptr), n_points, 0u, key_bits(ncell));
  return bytes;
}

hipError_t launch_build_grid(const BuildArgs & a, hipStream_t stream)
{
  hipError_t e;
  const uint32_t ncell = a.grid.ncell;
  e = hipMemsetAsync(a.seg_begin, 0, static_cast<size_t>(ncell + 1) * 2 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  if (a.n_points > 0)
  {
    const uint32_t pb = (a.n_points + 255) / 256;
    hipLaunchKernelGGL(points_kernel, dim3(pb), dim3(256), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t temp = a.sort_temp_bytes;
    e = rocprim::radix_sort_pairs<BuildSortConfig>(a.sort_temp, temp, a.keys_in, a.keys_out, a.vals_in,
                                                   a.vals_out, a.n_points, 0u, key_bits(ncell), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(segments_kernel, dim3(pb), dim3(256), 0, stream, a.keys_out, a.n_points,
                       a.seg_begin, a.seg_begin + (ncell + 1));
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  // a lane per (cell, quantity) where cells hold points by the dozen (local and loop-closure maps,
  // cfg-3: 9 per cell on average, hundreds in a wall's cells); a lane per cell on sparse grids
  if (static_cast<uint64_t>(a.n_points) >= 4ull * ncell)
  {
    hipLaunchKernelGGL(cell_sums_kernel, dim3((ncell * 8 + 255) / 256), dim3(256), 0, stream, a,
                       a.vals_out, a.seg_begin, a.seg_begin + (ncell + 1));
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cells_kernel<false>, dim3((ncell + 1 + 255) / 256), dim3(256), 0, stream, a,
                       a.vals_out, a.seg_begin, a.seg_begin + (ncell + 1));
  }
  else
  {
    hipLaunchKernelGGL(cells_kernel<true>, dim3((ncell + 1 + 255) / 256), dim3(256), 0, stream, a,
                       a.vals_out, a.seg_begin, a.seg_begin + (ncell + 1));
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_grid_tail(a.grid, a.cells_lds_image, a.occ_bits, a.cell_bytes, stream);
}

// occupancy bitmap + per-cell map bytes from the packed records
hipError_t launch_grid_tail(const GridDesc & geometry, const double * cells_lds_image,
                            uint32_t * occ_bits, uint8_t * cell_bytes, hipStream_t stream)
{
  GridDesc g = geometry;
  g.cells_lds_image = cells_lds_image;
  g.occ_bits = occ_bits;
  const uint32_t n_bytes = (g.size_x + 2) * (g.size_y + 2);
  hipLaunchKernelGGL(cell_bytes_kernel, dim3((n_bytes + 15) / 16), dim3(256), 0, stream, g, cell_bytes);
  return hipGetLastError();
}

hipError_t launch_pack_grid(const GridDesc & geometry, const double * cells6,
                            double * cells_lds_image, double * cells_global, uint32_t * occ_bits,
                            uint8_t * cell_bytes, hipStream_t stream)
{
  const uint32_t pack_blocks = (geometry.ncell + 1 + 255) / 256;
  const uint32_t n_bytes = (geometry.size_x + 2) * (geometry.size_y + 2);
  const uint32_t byte_blocks = (n_bytes + 15) / 16;
  hipLaunchKernelGGL(pack_grid_kernel, dim3(pack_blocks + byte_blocks), dim3(256), 0, stream, geometry,
                     cells6, pack_blocks, cells_lds_image, cells_global, occ_bits, cell_bytes);
  return hipGetLastError();
}

hipError_t launch_grid_install(const GridDesc & geometry, const SparseImage & image, double * cells_lds_image,
                               double * cells_global, uint32_t * occ_bits, uint8_t * cell_bytes,
                               uint16_t * ranks, uint32_t n_occ, hipStream_t stream)
{
  const uint32_t n_words = (geometry.ncell + 1 + 31) / 32;
  const uint32_t fill_chunks = (n_words + 31) / 32;
  const uint32_t list_chunks = (image.n + 255) / 256;
  uint32_t blocks = fill_chunks < 2048 ? fill_chunks : 2048;
  if (list_chunks > blocks) blocks = list_chunks < 4096 ? list_chunks : 4096;
  // (every thread's first piece of the compacted records is requested up front: a block per 256 pieces)
  const uint32_t copy_blocks = image.dst != nullptr ? static_cast<uint32_t>((image.n_compact / 2 + 255) / 256) : 0u;
  if (copy_blocks > blocks) blocks = copy_blocks < 4096 ? copy_blocks : 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(grid_install_kernel, dim3(blocks), dim3(256), 0, stream, geometry, image, cells_lds_image,
                     cells_global, occ_bits, cell_bytes, ranks, n_occ);
  return hipGetLastError();
}

hipError_t launch_sparse_bytes(const GridDesc & geometry, const SparseBytesJob & job, hipStream_t stream)
{
  if (job.n == 0) return hipSuccess;
  hipLaunchKernelGGL(grid_bytes_sparse_kernel, dim3(sparse_bytes_blocks(job)), dim3(256), 0, stream, geometry, job);
  return hipGetLastError();
}

hipError_t launch_grid_sparse_to_dense(const uint32_t * cell_index, const double * cells6, uint32_t n,
                                       uint32_t ncell, double * dense6, hipStream_t stream)
{
  hipError_t e = hipMemsetAsync(dense6, 0, static_cast<size_t>(ncell) * 6 * sizeof(double), stream);
  if (e != hipSuccess || n == 0) return e;
  hipLaunchKernelGGL(grid_scatter_raw_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cell_index, cells6,
                     n, ncell, dense6);
  return hipGetLastError();
}

hipError_t launch_compact_grid(uint32_t ncell, const double * cells_lds_image, const uint32_t * occ_bits,
                               double * records, uint16_t * ranks, uint32_t * n_occ_out, hipStream_t stream)
{
  if (ncell >= 65535u) return hipErrorInvalidValue;
  hipLaunchKernelGGL(compact_grid_kernel, dim3(1), dim3(kCompactThreads), 0, stream, ncell, cells_lds_image,
                     occ_bits, records, ranks, n_occ_out);
  return hipGetLastError();
}

}  // namespace ndt2d
