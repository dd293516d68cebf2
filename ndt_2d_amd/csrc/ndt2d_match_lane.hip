// matchScan search, lane-per-candidate mapping (gfx950 / MI355X).
//
// Reference: ScanMatcherNDT::matchScan's loops, src/scan_matcher_ndt.cpp:103-143.
//
// A wavefront owns an 8 x 8 patch of (dx, dy) translations of one theta: lane l
// is the candidate (ix0 + l/8, iy0 + l%8).  The wave walks the beams in the
// reference's order, so every lane accumulates its candidate's likelihood sum
// exactly as NDT::likelihood(points_inner) does (src/ndt_model.cpp:178-187) --
// no cross-lane reduction, no reordering.  The rotated beam (points_outer,
// :108-115) is wave-uniform; a small pre-kernel tabulates it per (theta, beam)
// and the search kernel reads it with scalar loads.
//
// The 64 points of a patch for one beam lie within 0.16 m of each other, so
// they fall into one to four NDT cells, and four cells out of five hold no
// distribution (n < 5: likelihood exactly 0, src/ndt_model.cpp:107).  The kernel
// is VALU-issue bound, so each beam first takes a 9-instruction look-up:
//
//   fixed-point cell coordinate  f = int(k + d)      k: per beam (scalar),
//                                                    d: per lane (register)
//   map byte                     m = lds_map[(fy >> ..) | (fx >> F)]
//
// lds_map is the occupancy of the grid, padded so that no clamping is needed:
// bit 0 = the cell holds a distribution, bit 1 = the cell or one of its eight
// neighbours does.  If no lane of the wave sees bit 1 the beam contributes
// exactly +0.0 to all 64 sums and is skipped -- even a coordinate that the
// fixed-point arithmetic puts on the wrong side of a cell boundary lands in a
// neighbour, which bit 1 covers.  Otherwise the wave checks which lanes are
// within 4 fixed-point units of a boundary ("near"); if any lane is occupied or
// near it runs the exact reference arithmetic (points_inner :121-125,
// NDT::getIndex src/ndt_model.cpp:203-218 for near lanes, Cell::score :105-116).
// Every skipped term is an exact zero, so the sums are bit-identical to the
// unskipped evaluation.
#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kLaneThreads = 1024;
constexpr int kLaneWaves = kLaneThreads / kWave;
constexpr int kPatch = 8;          // patch is kPatch x kPatch candidates = one wave
constexpr int kNearUnits = 4;      // fixed-point guard band around cell boundaries
constexpr int kUnroll = 4;

struct LaneGeom
{
  int32_t pad;        // border cells on every side of the grid in the map
  int32_t map_w_log2; // map row stride = 2^map_w_log2 >= size_x + 2 * pad
  int32_t map_h;      // size_y + 2 * pad
  int32_t fbits;      // fixed-point fraction bits
  double scale;       // 2^fbits
  double k_min, k_max_x, k_max_y;  // clamp of the per-beam fixed-point coordinate
};

// points_outer for the slab (reference :106-115) plus the fixed-point map
// coordinate of each rotated beam:
//   outer[t][b] = {ox, oy, kx, ky},  k = ((o - origin) * inv_cell + pad) * 2^fbits
// clamped so that k + d stays inside the map for every lane offset d; a clamped
// beam is further outside the grid than any offset can bring back, so it stays
// in the (empty) border.
__global__ void __launch_bounds__(256) outer_table_kernel(const MatchArgs a, double4 * outer,
                                                          const LaneGeom geo)
{
  const uint64_t n = static_cast<uint64_t>(a.th_end - a.th_begin) * a.n_beams;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const uint32_t t = static_cast<uint32_t>(i / a.n_beams);
    const uint32_t b = static_cast<uint32_t>(i - static_cast<uint64_t>(t) * a.n_beams);
    const uint32_t ith = a.th_begin + t;
    const double ct = a.cos_th[ith];
    const double st = a.sin_th[ith];
    const double2 p = reinterpret_cast<const double2 *>(a.beams_xy)[b];
    double4 o;
    o.x = p.x * ct - p.y * st + a.pose_x;
    o.y = p.x * st + p.y * ct + a.pose_y;
    const double kx = ((o.x - a.grid.origin_x) * a.grid.inv_cell_size + geo.pad) * geo.scale;
    const double ky = ((o.y - a.grid.origin_y) * a.grid.inv_cell_size + geo.pad) * geo.scale;
    // !(k >= min) also catches NaN
    o.z = !(kx >= geo.k_min) ? geo.k_min : (kx > geo.k_max_x ? geo.k_max_x : kx);
    o.w = !(ky >= geo.k_min) ? geo.k_min : (ky > geo.k_max_y ? geo.k_max_y : ky);
    outer[i] = o;
  }
}

struct LaneCtx
{
  const double * lds_cells;
  const uint8_t * lds_map;
  int32_t fbits;
  int32_t row_shift;   // fbits - map_w_log2
  int32_t row_mask;    // ~(2^map_w_log2 - 1)
  int32_t frac_mask;   // 2^fbits - 1
  int32_t pad;
  int32_t size_x;
};

// U consecutive beams of one patch.
template <int U, bool POW2>
__device__ __forceinline__ void lane_beams(const GridDesc & g, const LaneCtx & c,
                                           const double4 * __restrict__ row, double dx, double dy,
                                           double dxs, double dys, double & sum)
{
  double4 o[U];
  int32_t fx[U], fy[U];
  uint32_t m[U];
  uint32_t any_bits = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) o[u] = row[u];
#pragma unroll
  for (int u = 0; u < U; ++u)
  {
    fx[u] = __double2int_rz(o[u].z + dxs);
    fy[u] = __double2int_rz(o[u].w + dys);
    const int32_t idx = ((fy[u] >> c.row_shift) & c.row_mask) | (fx[u] >> c.fbits);
    m[u] = c.lds_map[idx];
    any_bits |= m[u];
  }
  if (__any(any_bits & 2u))
  {
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (__any(m[u] & 2u))
      {
        const bool near =
          (static_cast<uint32_t>((fx[u] + kNearUnits) & c.frac_mask) < 2u * kNearUnits) |
          (static_cast<uint32_t>((fy[u] + kNearUnits) & c.frac_mask) < 2u * kNearUnits);
        const bool occ = (m[u] & 1u) != 0;
        if (__any(occ | near))
        {
          // points_inner (:121-125) and Cell::score, exact
          const double px = o[u].x + dx;
          const double py = o[u].y + dy;
          uint32_t idx;
          if (__any(near))
          {
            idx = cell_index<POW2>(g, px, py);
          }
          else
          {
            // interior of a cell: the look-up cell is the reference's cell
            const int32_t cx = (fx[u] >> c.fbits) - c.pad;
            const int32_t cy = (fy[u] >> c.fbits) - c.pad;
            idx = occ ? static_cast<uint32_t>(cy * c.size_x + cx) : g.ncell;
          }
          sum += indexed_likelihood<true>(g, c.lds_cells, idx, px, py);
        }
      }
    }
  }
}

template <bool POW2>
__global__ void __launch_bounds__(kLaneThreads) match_lane_kernel(
  const MatchArgs a, const double4 * __restrict__ outer, const LaneGeom geo)
{
  extern __shared__ __align__(16) double lds[];
  const GridDesc & g = a.grid;
  double * lds_cells = lds;
  uint8_t * lds_map = reinterpret_cast<uint8_t *>(lds + static_cast<size_t>(g.ncell + 1) * kCellDoubles);

  // LDS image: packed cell records + padded occupancy map
  stage_grid_to_lds(g, lds_cells);
  {
    const int32_t w = 1 << geo.map_w_log2;
    const int32_t sx = static_cast<int32_t>(g.size_x), sy = static_cast<int32_t>(g.size_y);
    for (int32_t i = threadIdx.x; i < w * geo.map_h; i += kLaneThreads)
    {
      const int32_t cx = (i & (w - 1)) - geo.pad, cy = (i >> geo.map_w_log2) - geo.pad;
      uint32_t self = 0, around = 0;
      for (int32_t ny = cy - 1; ny <= cy + 1; ++ny)
      {
        for (int32_t nx = cx - 1; nx <= cx + 1; ++nx)
        {
          if (nx >= 0 && nx < sx && ny >= 0 && ny < sy)
          {
            const bool o =
              g.cells_lds_image[static_cast<size_t>(ny * sx + nx) * kCellDoubles + 5] != 0.0;
            around |= o ? 1u : 0u;
            if (nx == cx && ny == cy) self = o ? 1u : 0u;
          }
        }
      }
      lds_map[i] = static_cast<uint8_t>(self | (around << 1));
    }
  }
  __syncthreads();

  LaneCtx c;
  c.lds_cells = lds_cells;
  c.lds_map = lds_map;
  c.fbits = geo.fbits;
  c.row_shift = geo.fbits - geo.map_w_log2;
  c.row_mask = ~((1 << geo.map_w_log2) - 1);
  c.frac_mask = (1 << geo.fbits) - 1;
  c.pad = geo.pad;
  c.size_x = static_cast<int32_t>(g.size_x);

  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lx = lane >> 3, ly = lane & 7;
  const uint32_t n_lin = a.n_lin;
  const uint32_t patches_1d = (n_lin + kPatch - 1) / kPatch;
  const uint32_t patches = patches_1d * patches_1d;
  const uint32_t n_items = (a.th_end - a.th_begin) * patches;
  const uint32_t n_workers = gridDim.x * kLaneWaves;
  const uint32_t worker = wave * gridDim.x + blockIdx.x;
  const uint64_t per_theta = static_cast<uint64_t>(n_lin) * n_lin;
  const double inv_scaled = g.inv_cell_size * geo.scale;

  double best_s = 0.0;       // `double best_score = 0;` (:83)
  double best_i = kNoIndex;
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;

  for (uint32_t item = worker; item < n_items; item += n_workers)
  {
    const uint32_t t = item / patches;
    const uint32_t p = item - t * patches;
    const uint32_t pxi = p / patches_1d;
    const uint32_t pyi = p - pxi * patches_1d;
    const uint32_t ix = pxi * kPatch + lx;
    const uint32_t iy = pyi * kPatch + ly;
    const bool valid = (ix < n_lin) & (iy < n_lin);
    // lanes beyond the lattice edge shadow the edge candidate and are dropped below
    const double dx = a.dlin[min(ix, n_lin - 1)];
    const double dy = a.dlin[min(iy, n_lin - 1)];
    const double dxs = dx * inv_scaled;
    const double dys = dy * inv_scaled;
    const double4 * __restrict__ row = outer + static_cast<size_t>(t) * a.n_beams;

    double sum = 0.0;
    uint32_t b = 0;
    for (; b + kUnroll <= a.n_beams; b += kUnroll)
    {
      lane_beams<kUnroll, POW2>(g, c, row + b, dx, dy, dxs, dys, sum);
    }
    for (; b < a.n_beams; ++b)
    {
      lane_beams<1, POW2>(g, c, row + b, dx, dy, dxs, dys, sum);
    }

    if (valid)
    {
      const double score = -sum;  // (:127)
      const uint64_t local = static_cast<uint64_t>(t) * per_theta + static_cast<uint64_t>(ix) * n_lin + iy;
      const double flat = static_cast<double>(static_cast<uint64_t>(a.th_begin) * per_theta + local);
      if (score < best_s)
      {
        best_s = score;
        best_i = flat;
      }
      // k += x x^T score, u += x score, s += score (:137-140)
      const double dt = a.dth[a.th_begin + t];
      acc[0] += (dx * dx) * score;
      acc[1] += (dx * dy) * score;
      acc[2] += (dx * dt) * score;
      acc[3] += (dy * dy) * score;
      acc[4] += (dy * dt) * score;
      acc[5] += (dt * dt) * score;
      acc[6] += dx * score;
      acc[7] += dy * score;
      acc[8] += dt * score;
      acc[9] += score;
      if (a.scores != nullptr) a.scores[local] = score;
    }
  }

  // wave-level reduction of the per-lane records
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1)
  {
    const double os = __shfl_xor(best_s, off, kWave);
    const double oi = __shfl_xor(best_i, off, kWave);
    if (better(os, oi, best_s, best_i))
    {
      best_s = os;
      best_i = oi;
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = wave_sum(acc[k]);

  if (lane == 0)
  {
    double * out = a.partials + static_cast<size_t>(worker) * kRecord;
    out[0] = best_s;
    out[1] = best_i;
#pragma unroll
    for (int k = 0; k < 10; ++k) out[2 + k] = acc[k];
  }
}

// Map geometry for a search; false if the fixed-point range cannot hold it.
bool lane_geometry(const MatchArgs & args, LaneGeom * geo, size_t * map_bytes)
{
  const double lin_cells = args.dlin_absmax * args.grid.inv_cell_size;
  if (!(lin_cells >= 0.0) || lin_cells > 4096.0) return false;
  const int32_t pad = static_cast<int32_t>(2.0 * lin_cells) + 3;
  const uint64_t need_w = static_cast<uint64_t>(args.grid.size_x) + 2 * pad;
  const uint64_t need_h = static_cast<uint64_t>(args.grid.size_y) + 2 * pad;
  int w_log2 = 0;
  while ((1ull << w_log2) < need_w) ++w_log2;
  // coordinates up to max(need_w, need_h) cells must fit a positive int32
  int c_log2 = w_log2;
  while ((1ull << c_log2) < need_h) ++c_log2;
  int fbits = 30 - c_log2;
  if (fbits > 24) fbits = 24;
  if (fbits < 8 || fbits < w_log2) return false;
  geo->pad = pad;
  geo->map_w_log2 = w_log2;
  geo->map_h = static_cast<int32_t>(need_h);
  geo->fbits = fbits;
  geo->scale = static_cast<double>(1u << fbits);
  // lanes add |d| <= lin_cells * scale; keep one cell of margin on both sides
  const double reach = (lin_cells + 1.0) * geo->scale;
  geo->k_min = reach;
  geo->k_max_x = static_cast<double>(need_w - 1) * geo->scale - reach;
  geo->k_max_y = static_cast<double>(need_h - 1) * geo->scale - reach;
  *map_bytes = (static_cast<size_t>(1u << w_log2) * need_h + 15) & ~size_t(15);
  return true;
}

}  // namespace

size_t match_lane_outer_doubles(const MatchArgs & args)
{
  return static_cast<size_t>(args.th_end - args.th_begin) * args.n_beams * 4;
}

bool match_lane_supported(const MatchArgs & args, size_t lds_per_block)
{
  LaneGeom geo;
  size_t map_bytes = 0;
  if (!lane_geometry(args, &geo, &map_bytes)) return false;
  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t items = static_cast<uint64_t>(args.th_end - args.th_begin) * p1 * p1;
  return grid_bytes + map_bytes <= lds_per_block && items < (1ull << 32);
}

hipError_t launch_match_lane(const MatchArgs & args_in, double * outer, double * workspace,
                             uint32_t max_workers, int cus, hipStream_t stream,
                             uint32_t * n_workers_out)
{
  MatchArgs args = args_in;
  args.partials = workspace;
  LaneGeom geo;
  size_t map_bytes = 0;
  if (!lane_geometry(args, &geo, &map_bytes)) return hipErrorInvalidValue;

  const uint64_t n_outer = static_cast<uint64_t>(args.th_end - args.th_begin) * args.n_beams;
  uint32_t oblocks = static_cast<uint32_t>((n_outer + 255) / 256);
  if (oblocks > 4096) oblocks = 4096;
  hipLaunchKernelGGL(outer_table_kernel, dim3(oblocks), dim3(256), 0, stream, args,
                     reinterpret_cast<double4 *>(outer), geo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;

  const uint32_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t n_items = static_cast<uint64_t>(args.th_end - args.th_begin) * p1 * p1;
  uint32_t blocks = static_cast<uint32_t>((n_items + kLaneWaves - 1) / kLaneWaves);
  uint32_t max_blocks = static_cast<uint32_t>(cus);
  if (max_blocks * kLaneWaves > max_workers) max_blocks = max_workers / kLaneWaves;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;

  const size_t lds_bytes =
    static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double) + map_bytes;
  auto launch = [&](auto kernel) -> hipError_t {
    if (lds_bytes > 48 * 1024)
    {
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(lds_bytes));
      if (e2 != hipSuccess) return e2;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kLaneThreads), lds_bytes, stream, args,
                       reinterpret_cast<const double4 *>(outer), geo);
    return hipGetLastError();
  };
  e = args.grid.pow2 ? launch(match_lane_kernel<true>) : launch(match_lane_kernel<false>);
  if (n_workers_out != nullptr) *n_workers_out = blocks * kLaneWaves;
  return e;
}

}  // namespace ndt2d
