// matchScan search for SMALL lattices, lane-per-candidate with the beams split across
// the waves of a block (gfx950 / MI355X).
//
// Reference: ScanMatcherNDT::matchScan's loops, src/scan_matcher_ndt.cpp:103-143.
//
// The node's own searches are small: the plugin's defaults (src/scan_matcher_ndt.cpp:
// 37-44) give 21 x 21 x 80 = 35,280 candidates x 100 beams, i.e. 80 x 3 x 3 = 720
// patches of 8 x 8 translations.  The large-lattice kernel (ndt2d_match_lane.hip) gives
// a wave one patch and lets it walk all the beams: with fewer patches than the chip has
// SIMDs that is one wave per SIMD at best, every look-up -> exact-evaluation chain runs
// un-overlapped, and a table pre-kernel precedes it.  Here a block owns up to P patches
// of ONE theta step and its waves split the beams:
//
//   block   rotates the scan once for its theta (points_outer, :106-115) into LDS rows
//           {ox, oy, K} and copies its window of the per-cell map bytes prepared when
//           the grid was installed (ndt2d_build.hip, cell_bytes_kernel) -- no pre-kernel;
//   wave w  = (patch slot w / C, beam chunk w % C): lane = candidate of the patch, the
//           wave adds the likelihoods of its chunk of beams in beam order, with the same
//           two-instruction map look-up / bit-exact skipping / exact reference arithmetic
//           as the large kernel (ndt2d_lane_fn.h);
//   combine the C partial sums of a candidate are added in chunk order,
//           ((p_0 + p_1) + p_2) + ..., by the slot's first wave, which then keeps the
//           reference's strict-< best (:128-134) and the covariance accumulators
//           (:137-140) and leaves one 12-double record per (theta, patch) item.
//
// A candidate's sum is therefore a fixed-order sum of in-order chunk sums: deterministic,
// independent of timing, and within a few ulps of the reference's single running sum.
// Records are reduced by match_reduce_kernel exactly as for the other mappings.
#include "ndt2d_lane_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kSmallMaxWaves = 16;
constexpr uint32_t kSmallMaxBeams = 2048;     // 64 KB of LDS rows
constexpr uint32_t kRowDoubles = 4;           // {ox, oy, K, -}: two 16-byte LDS reads

struct SmallPlan
{
  uint32_t patches_per_block;   // P: patch slots of a block (all of one theta step)
  uint32_t chunks;              // C: beam chunks = waves per patch slot
  uint32_t chunk_beams;         // beams per chunk (a multiple of kUnroll)
  uint32_t blocks_per_theta;    // ceil(patches / P)
  uint32_t need_w;              // map columns in use: window + 2 * pad
};

template <bool POW2>
__global__ void __launch_bounds__(kSmallMaxWaves * kWave) match_small_kernel(const MatchArgs a,
                                                                             const LaneGeom geo,
                                                                             const SmallPlan plan)
{
  // LDS: [map, at offset 0 so that the packed cell bytes are the address][rows][partials]
  extern __shared__ __align__(16) double lds[];
  const GridDesc & g = a.grid;
  if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
  uint8_t * lds_map = reinterpret_cast<uint8_t *>(lds);
  const uint32_t map_bytes = static_cast<uint32_t>(geo.map_h) * kMapStride;
  double * rows = lds + map_bytes / sizeof(double);
  double * partials = rows + static_cast<size_t>(a.n_beams) * kRowDoubles;

  const uint32_t n_threads = blockDim.x;
  const uint32_t t_local = blockIdx.x / plan.blocks_per_theta;
  const uint32_t first_patch = (blockIdx.x - t_local * plan.blocks_per_theta) * plan.patches_per_block;
  const uint32_t ith = a.th_begin + t_local * a.th_stride;

  // map window: one byte per grid cell, copied from the grid's extended byte image;
  // cells further than one cell outside the grid cannot be reached by anything: 0
  {
    const int32_t ew = static_cast<int32_t>(g.size_x) + 2, eh = static_cast<int32_t>(g.size_y) + 2;
    const uint32_t n = plan.need_w * static_cast<uint32_t>(geo.map_h);
    for (uint32_t i = threadIdx.x; i < n; i += n_threads)
    {
      const uint32_t my = i / plan.need_w, mx = i - my * plan.need_w;
      const int32_t ex = static_cast<int32_t>(mx) - geo.pad + geo.win_x0 + 1;
      const int32_t ey = static_cast<int32_t>(my) - geo.pad + geo.win_y0 + 1;
      uint8_t v = 0;
      if (ex >= 0 && ex < ew && ey >= 0 && ey < eh) v = g.cell_bytes[ey * ew + ex];
      lds_map[my * kMapStride + mx] = v;
    }
  }
  // points_outer (:106-115) and the packed fixed-point map coordinate of each beam
  {
    const double ct = a.cos_th[ith];
    const double st = a.sin_th[ith];
    for (uint32_t b = threadIdx.x; b < a.n_beams; b += n_threads)
    {
      const double2 p = reinterpret_cast<const double2 *>(a.beams_xy)[b];
      const double ox = p.x * ct - p.y * st + a.pose_x;
      const double oy = p.x * st + p.y * ct + a.pose_y;
      double kx = ((ox - g.origin_x) * g.inv_cell_size + (geo.pad - geo.win_x0)) * geo.unit_scale;
      double ky = ((oy - g.origin_y) * g.inv_cell_size + (geo.pad - geo.win_y0)) * geo.unit_scale;
      // !(k >= min) also catches NaN
      kx = !(kx >= geo.k_min) ? geo.k_min : (kx > geo.k_max_x ? geo.k_max_x : kx);
      ky = !(ky >= geo.k_min) ? geo.k_min : (ky > geo.k_max_y ? geo.k_max_y : ky);
      double4 o;
      o.x = ox;
      o.y = oy;
      o.z = kTwo52 + (rint(ky) * kTwo24 + rint(kx));
      o.w = 0.0;
      reinterpret_cast<double4 *>(rows)[b] = o;
    }
  }
  __syncthreads();

  LaneCtx c;
  c.lds_cells_address = 0;   // records are gathered from the 64-byte-stride HBM copy
  c.sub_log2 = 0;
  c.idx_bias = static_cast<uint32_t>(geo.pad - geo.win_y0) * g.size_x +
               static_cast<uint32_t>(geo.pad - geo.win_x0);
  c.size_x = g.size_x;

  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wave / plan.chunks;
  const uint32_t chunk = wave - slot * plan.chunks;
  const uint32_t n_lin = a.n_lin;
  const uint32_t patches_1d = (n_lin + kPatch - 1) / kPatch;
  const uint32_t patches = patches_1d * patches_1d;
  const uint32_t patch = first_patch + slot;
  const bool active = slot < plan.patches_per_block && patch < patches;

  const uint32_t lx = lane >> 3, ly = lane & 7;
  const uint32_t pxi = active ? patch / patches_1d : 0;
  const uint32_t pyi = active ? patch - pxi * patches_1d : 0;
  const uint32_t ix = pxi * kPatch + lx;
  const uint32_t iy = pyi * kPatch + ly;
  const bool valid = active & (ix < n_lin) & (iy < n_lin);
  // lanes beyond the lattice edge shadow the edge candidate and are dropped below
  const double dx = a.dlin[min(ix, n_lin - 1)];
  const double dy = a.dlin[min(iy, n_lin - 1)];

  if (active)
  {
    const double inv_scaled = g.inv_cell_size * geo.unit_scale;
    const double dxy = rint(dy * inv_scaled) * kTwo24 + rint(dx * inv_scaled);
    const uint32_t b0 = min(chunk * plan.chunk_beams, a.n_beams);
    const uint32_t b1 = min(b0 + plan.chunk_beams, a.n_beams);
    const double4 * row = reinterpret_cast<const double4 *>(rows);
    double sum = 0.0;
    SkipState skip = skip_state(0.0, geo.no_skip);
    uint32_t b = b0;
    for (; b + kUnroll <= b1; b += kUnroll)
    {
      double4 o[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) o[u] = row[b + u];
      lane_beams<kUnroll, POW2, false, false>(g, c, o, dx, dy, dxy, sum, skip, geo.no_skip);
    }
    for (; b < b1; ++b)
    {
      const double4 one[1] = {row[b]};
      lane_beams<1, POW2, false, false>(g, c, one, dx, dy, dxy, sum, skip, geo.no_skip);
    }
    partials[wave * kWave + lane] = sum;
  }
  __syncthreads();

  if (active && chunk == 0)
  {
    // ((p_0 + p_1) + p_2) + ... in chunk order
    const double * mine = partials + static_cast<size_t>(slot) * plan.chunks * kWave + lane;
    double sum = mine[0];
    for (uint32_t j = 1; j < plan.chunks; ++j) sum += mine[j * kWave];

    double best_s = 0.0;       // `double best_score = 0;` (:83)
    double best_i = kNoIndex;
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
    if (valid)
    {
      const double score = -sum;  // (:127)
      const uint64_t per_theta = static_cast<uint64_t>(n_lin) * n_lin;
      const uint64_t in_theta = static_cast<uint64_t>(ix) * n_lin + iy;
      if (score < 0.0)
      {
        best_s = score;
        best_i = static_cast<double>(static_cast<uint64_t>(ith) * per_theta + in_theta);
      }
      // k += x x^T score, u += x score, s += score (:137-140)
      const double dt = a.dth[ith];
      acc[0] = (dx * dx) * score;
      acc[1] = (dx * dy) * score;
      acc[2] = (dx * dt) * score;
      acc[3] = (dy * dy) * score;
      acc[4] = (dy * dt) * score;
      acc[5] = (dt * dt) * score;
      acc[6] = dx * score;
      acc[7] = dy * score;
      acc[8] = dt * score;
      acc[9] = score;
      if (a.scores != nullptr) a.scores[static_cast<uint64_t>(t_local) * per_theta + in_theta] = score;
    }
    wave_best_to_last_lane(best_s, best_i);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);
    if (lane == kWave - 1)
    {
      double * out = a.partials + (static_cast<size_t>(t_local) * patches + patch) * kRecord;
      out[0] = best_s;
      out[1] = best_i;
#pragma unroll
      for (int k = 0; k < 10; ++k) out[2 + k] = acc[k];
    }
  }
}

size_t small_lds_bytes(const MatchArgs & args, const LaneGeom & geo, uint32_t waves)
{
  return static_cast<size_t>(geo.map_h) * kMapStride +
         static_cast<size_t>(args.n_beams) * kRowDoubles * sizeof(double) +
         static_cast<size_t>(waves) * kWave * sizeof(double);
}

// How a lattice is cut into blocks: enough waves to fill the chip's wave slots once
// (4 per SIMD at this kernel's register budget), every wave with whole look-up groups.
// The cut follows the WHOLE lattice (n_th), not the theta steps of this launch: a
// candidate's chunks, hence the bits of its score, are the same whichever rank of a
// sharded search evaluates it.
SmallPlan small_plan(const MatchArgs & args, const LaneGeom & geo, int cus)
{
  const uint32_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint32_t patches = p1 * p1;
  const uint64_t items = static_cast<uint64_t>(args.n_th) * patches;
  const uint32_t groups = (args.n_beams + kUnroll - 1) / kUnroll;
  const uint64_t wave_slots = static_cast<uint64_t>(cus) * 16;
  uint64_t c = items > 0 ? wave_slots / items : 1;
  if (c < 1) c = 1;
  if (c > static_cast<uint64_t>(kSmallMaxWaves)) c = kSmallMaxWaves;
  if (c > groups) c = groups;
  const uint32_t chunk_groups = (groups + static_cast<uint32_t>(c) - 1) / static_cast<uint32_t>(c);
  SmallPlan plan;
  plan.chunks = (groups + chunk_groups - 1) / chunk_groups;
  plan.chunk_beams = chunk_groups * kUnroll;
  uint32_t p = kSmallMaxWaves / plan.chunks;
  if (p > patches) p = patches;
  if (p < 1) p = 1;
  plan.patches_per_block = p;
  plan.blocks_per_theta = (patches + p - 1) / p;
  plan.need_w = static_cast<uint32_t>(geo.win_w + 2 * geo.pad);
  return plan;
}

}  // namespace

bool match_small_supported(const MatchArgs & args, size_t lds_per_block)
{
  LaneGeom geo;
  size_t map_bytes = 0;
  if (args.grid.cell_bytes == nullptr || args.n_beams == 0 || args.n_beams > kSmallMaxBeams) return false;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes, true)) return false;
  if (geo.sub_log2 != 0) return false;
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t items = static_cast<uint64_t>(args.th_end - args.th_begin) * p1 * p1;
  return items <= kSmallMaxItems && args.grid.size_x < (1u << 24) && args.grid.ncell < (1u << 24) &&
         small_lds_bytes(args, geo, kSmallMaxWaves) <= lds_per_block;
}

hipError_t launch_match_small(const MatchArgs & args_in, double * workspace, int cus,
                              size_t lds_per_block, bool no_skip, hipStream_t stream,
                              uint32_t * n_records_out)
{
  MatchArgs args = args_in;
  args.partials = workspace;
  LaneGeom geo;
  size_t map_bytes = 0;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes, true) || geo.sub_log2 != 0)
  {
    return hipErrorInvalidValue;
  }
  geo.no_skip = no_skip ? 1 : 0;
  const SmallPlan plan = small_plan(args, geo, cus);
  const uint32_t waves = plan.patches_per_block * plan.chunks;
  const uint32_t blocks = (args.th_end - args.th_begin) * plan.blocks_per_theta;
  const size_t lds_bytes = small_lds_bytes(args, geo, waves);
  auto launch = [&](auto kernel) -> hipError_t {
    if (lds_bytes > 48 * 1024)
    {
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(lds_bytes));
      if (e2 != hipSuccess) return e2;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves * kWave), lds_bytes, stream, args, geo, plan);
    return hipGetLastError();
  };
  const hipError_t e = args.grid.pow2 ? launch(match_small_kernel<true>) : launch(match_small_kernel<false>);
  const uint32_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  if (n_records_out != nullptr) *n_records_out = (args.th_end - args.th_begin) * p1 * p1;
  return e;
}

}  // namespace ndt2d
