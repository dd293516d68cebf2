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
// is VALU-issue bound, so each beam first takes a two-instruction look-up into
// an occupancy map of the grid held in LDS:
//
//   map coordinates are 8.16 fixed point (cell . fraction), x and y packed into
//   ONE double:  K = 2^52 + ky * 2^24 + kx  per beam (wave-uniform, SGPR),
//                D =        dy * 2^24 + dx  per lane (register).
//   K + D is exact integer arithmetic in the f64 mantissa (v_add_f64); the low
//   48 mantissa bits of the sum are {fy, fx}, so the map address
//   (cell_y << 8) | cell_x is two bytes of the sum picked by one v_perm_b32.
//
// Before any per-lane work the wave asks, ONE BEAM PER LANE, whether the patch as a
// whole can reach a distribution with that beam: the 64 candidates' coordinates lie
// within a few sub-cells of the patch's first lane, and if the 4 x 4 box of map bytes
// there holds level 0 only, every lane would skip the beam (patch_can_score, four LDS
// reads for 64 beams).  Groups of eight consecutive beams none of which passes are
// stepped over by a scalar test -- four beams out of five at cfg-2.
//
// The map is padded by more cells than any offset reaches and beams are
// pre-clamped, so no per-lane range check exists.  The map is kept at up to
// 4 x 4 sub-cells per cell.  Map byte: bit 0 = the cell holds a distribution;
// bits 2..7 = a level: 0 if no distribution can be hit from this sub-cell (its
// box, slightly widened because the fixed-point coordinate is rounded, touches no
// occupied cell), else an upper bound l - 63 of Cell::score's exponent over that
// box (63: no claim).  Each lane carries the level below which a term cannot
// change its sum (RN(s + exp(e)) == s); if every lane's byte is below its level
// the beam is skipped.  Otherwise lanes within 4 fixed-point units of a boundary
// are "near" (K carries a bias of 4 units on both fractions, so the test is a 16-bit
// compare per axis); if any live lane is occupied or near, the wave runs the exact
// reference arithmetic (points_inner :121-125, NDT::getIndex
// src/ndt_model.cpp:203-218 for near lanes, Cell::score :105-116).  Every
// skipped term is one that leaves the sum unchanged, so the sums are
// bit-identical to the unskipped evaluation (variant "lane-noskip" is that
// evaluation; the tests compare the two bitwise).
#include <cmath>
#include <cstdlib>

#include <type_traits>

#include "ndt2d_lane_fn.h"

namespace ndt2d
{

namespace
{

constexpr uint32_t kDynamicItemsFromBeams = 256;
constexpr int kLaneThreads = 1024;       // large searches: one block per CU
constexpr int kLaneThreadsSmall = 256;   // small searches: spread the few work items over more CUs
#ifndef NDT2D_LANE_THREADS_COMPACT
#define NDT2D_LANE_THREADS_COMPACT 768
#endif
constexpr int kLaneThreadsCompact = NDT2D_LANE_THREADS_COMPACT; // compacted records: two blocks per CU
#define NDT2D_LANE_COMPACT_WAVES_PER_EU (NDT2D_LANE_THREADS_COMPACT / 128)   // 768 threads: six waves per SIMD

// points_outer for the slab (reference :106-115) plus the packed fixed-point
// map coordinate of each rotated beam:
//   outer[t][b] = {ox, oy, K, 0},   K = 2^52 + ky * 2^24 + kx (+ kNearBias),
//   k = rint(((o - origin) * inv_cell - window_origin + pad) * 2^16) clamped to [k_min, k_max]
// so that k + d stays inside the map for every lane offset d.  A clamped beam
// is further outside the grid than any offset can bring back; it stays in the
// empty border.
//
// The same launch builds the occupancy map of the window (one byte per map
// cell, see the file header) once, in HBM; every search block then copies the
// finished image into its LDS.
__global__ void __launch_bounds__(256) outer_table_kernel(const MatchArgs a, double4 * outer,
                                                          uint8_t * map_out, const LaneGeom geo)
{
  {
    const GridDesc & g = a.grid;
    const uint32_t n_map = static_cast<uint32_t>(kMapStride) * geo.map_h;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_map; i += gridDim.x * 256)
    {
      // map sub-cell (mx, my): its box in the world, widened by kBoxMargin of a
      // sub-cell so that it holds every point whose rounded fixed-point coordinate
      // lands in this sub-cell, and the grid cells that widened box overlaps (its
      // own, plus the neighbours across a cell boundary it touches)
      const int32_t mx = static_cast<int32_t>(i & (kMapStride - 1));
      const int32_t my = static_cast<int32_t>(i >> 8);
      if (geo.block_log2 > 0)
      {
        // a map cell per block of grid cells
        const int32_t cells = 1 << geo.block_log2;
        map_out[i] = block_byte(g, mx * cells - geo.pad + geo.win_x0, my * cells - geo.pad + geo.win_y0, cells);
        continue;
      }
      const int32_t sub = 1 << geo.sub_log2;
      const int32_t cx = (mx >> geo.sub_log2) - geo.pad + geo.win_x0;
      const int32_t cy = (my >> geo.sub_log2) - geo.pad + geo.win_y0;
      map_out[i] = sub_cell_byte(g, cx, cy, mx & (sub - 1), my & (sub - 1), geo.sub_log2);
    }
  }

  // the work-item counter of the search kernel that follows on this stream
  if (blockIdx.x == 0 && threadIdx.x < kItemShards) a.next_item[threadIdx.x * kItemShardStride] = 0u;
  const uint64_t n = static_cast<uint64_t>(a.th_end - a.th_begin) * a.n_beams;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const uint32_t t = static_cast<uint32_t>(i / a.n_beams);
    const uint32_t b = static_cast<uint32_t>(i - static_cast<uint64_t>(t) * a.n_beams);
    const uint32_t ith = a.th_begin + t * a.th_stride;
    const double ct = a.cos_th[ith];
    const double st = a.sin_th[ith];
    const double2 p = reinterpret_cast<const double2 *>(a.beams_xy)[b];
    double4 o;
    o.x = p.x * ct - p.y * st + a.pose_x;
    o.y = p.x * st + p.y * ct + a.pose_y;
    double kx = ((o.x - a.grid.origin_x) * a.grid.inv_cell_size + (geo.pad - geo.win_x0)) * geo.unit_scale;
    double ky = ((o.y - a.grid.origin_y) * a.grid.inv_cell_size + (geo.pad - geo.win_y0)) * geo.unit_scale;
    // !(k >= min) also catches NaN
    kx = !(kx >= geo.k_min) ? geo.k_min : (kx > geo.k_max_x ? geo.k_max_x : kx);
    ky = !(ky >= geo.k_min) ? geo.k_min : (ky > geo.k_max_y ? geo.k_max_y : ky);
    o.z = kTwo52 + (rint(ky) * kTwo24 + rint(kx)) + kNearBias;
    o.w = 0.0;
    outer[i] = o;
  }
}

// Work item -> (theta step of the launch, patch).  Theta steps are visited from the middle
// of the range outwards: the steps around the scan's own heading are the expensive ones
// when the guess is any good.
__device__ __forceinline__ void item_place(uint32_t item, uint32_t patches, uint32_t patches_1d,
                                           uint32_t th_mid, uint32_t & t, uint32_t & pxi, uint32_t & pyi)
{
  const uint32_t rank = item / patches;
  const uint32_t p = item - rank * patches;
  t = (rank & 1u) ? th_mid + (rank + 1u) / 2u : th_mid - rank / 2u;
  pxi = p / patches_1d;
  pyi = p - pxi * patches_1d;
}

// The lane's packed fixed-point offset (see the file header).
__device__ __forceinline__ double packed_offset(double dx, double dy, double inv_scaled)
{
  return rint(dy * inv_scaled) * kTwo24 + rint(dx * inv_scaled);
}

// COMPACT (with LDS_RECORDS): instead of the whole grid's records the block keeps the
// compacted records of the cells that can score and the cell -> record table
// (GridDesc::compact_records / cell_rank): at cfg-2 15 KB instead of 81 KB, so that two
// blocks share a CU (match_lane_compact_kernel below).
// kNoIndex made where it is needed: held in a register pair across the beam loop it was the
// one value the 80-VGPR kernels still spilled (the compiler would not rematerialise it).
__device__ __forceinline__ double no_index_here()
{
  static_assert(kNoIndex == 1.0e308, "the literal below is 1.0e308");
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, 0x85ebc8a0\n\tv_mov_b32 %1, 0x7fe1ccf3" : "=v"(lo), "=v"(hi));
  return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
}

template <int THREADS, bool POW2, bool LDS_RECORDS, bool DYNAMIC_ITEMS, bool COMPACT, bool PARTS = false>
__device__ __forceinline__ void match_lane_body(
  const MatchArgs & a, const double4 * __restrict__ outer, const uint8_t * __restrict__ map_image,
  const LaneGeom & geo)
{
  // LDS image: padded occupancy map of the window (at offset 0, so the packed
  // cell bytes are the LDS address) followed, if they fit (LDS_RECORDS), by the
  // packed cell records of the whole grid; otherwise records are gathered from
  // the 64-byte-stride HBM copy (a patch touches 1-4 lines).
  extern __shared__ __align__(16) double lds[];
  const GridDesc & g = a.grid;
#if defined(NDT2D_LANE_HIST) || defined(NDT2D_LANE_PATHS)
  if (blockIdx.x == 0 && threadIdx.x == 0) g_lane_hist = a.scores;   // (set before any block uses it: see the experiment)
#endif
  uint8_t * lds_map = reinterpret_cast<uint8_t *>(lds);
  // lds_byte_at() addresses the map absolutely: it must start at LDS offset 0 -- no static
  // __shared__ in this kernel (the launcher checks: prepare_absolute_lds_kernel)
  const uint32_t map_bytes = static_cast<uint32_t>(geo.map_h) * kMapStride;
  const uint32_t rank_bytes = COMPACT ? compact_rank_bytes(g.ncell) : 0u;
  double * lds_cells = lds + (map_bytes + rank_bytes) / sizeof(double);

  if (COMPACT)
  {
    const uint4 * src_rank = reinterpret_cast<const uint4 *>(g.cell_rank);
    uint4 * dst_rank = reinterpret_cast<uint4 *>(lds_map + map_bytes + kRankLead);
    for (uint32_t i = threadIdx.x; i < (rank_bytes - kRankLead) / 16; i += THREADS) dst_rank[i] = src_rank[i];
    if (threadIdx.x == 0)
    {
      reinterpret_cast<uint16_t *>(lds_map + map_bytes + kRankLead)[-1] = static_cast<uint16_t>(g.n_occ);
    }
    const uint32_t n2 = (g.n_occ + 1) * kCellDoubles / 2;
    const double2 * src = reinterpret_cast<const double2 *>(g.compact_records);
    double2 * dst = reinterpret_cast<double2 *>(lds_cells);
    for (uint32_t i = threadIdx.x; i < n2; i += THREADS) dst[i] = src[i];
  }
  else if (LDS_RECORDS)
  {
    stage_grid_to_lds(g, lds_cells);
  }
  {
    // the map image was built by the pre-kernel; kMapStride * map_h is a multiple of 16
    const uint32_t n16 = static_cast<uint32_t>(kMapStride) * geo.map_h / 16;
    const uint4 * src = reinterpret_cast<const uint4 *>(map_image);
    uint4 * dst = reinterpret_cast<uint4 *>(lds_map);
    for (uint32_t i = threadIdx.x; i < n16; i += THREADS) dst[i] = src[i];
  }
  __syncthreads();

#ifdef NDT2D_LANE_TRACE
  // experiments/lane_wave_trace.py: when a wave starts / ends its items and how many it ran
  const unsigned long long trace_t0 = wall_clock64();
  unsigned long long trace_last = trace_t0, trace_max = 0, trace_last_start = trace_t0;
  uint32_t trace_items = 0, trace_prev_item = 0, trace_flagged = 0;
#endif
  LaneCtx c;
  c.rank_address = map_bytes + (COMPACT ? kRankLead : 0u);
  c.lds_cells_address = map_bytes + rank_bytes;
  c.sub_log2 = static_cast<uint32_t>(geo.sub_log2);
  c.exact_index = geo.block_log2 > 0 ? 1u : 0u;
  c.idx_bias = static_cast<uint32_t>(geo.pad - geo.win_y0) * g.size_x +
               static_cast<uint32_t>(geo.pad - geo.win_x0);
  c.size_x = g.size_x;

  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_lin = a.n_lin;
  const uint32_t patches_1d = (n_lin + kPatch - 1) / kPatch;
  const uint32_t patches = patches_1d * patches_1d;
  // PARTS: a work item is (theta, patch, part of the beams); item / beam_parts is the
  // (theta, patch) item of the unsplit search
  const uint32_t n_items = (a.th_end - a.th_begin) * patches * (PARTS ? a.beam_parts : 1u);
  constexpr uint32_t kLaneWaves = THREADS / kWave;
  const uint32_t n_workers = gridDim.x * kLaneWaves;
  const uint32_t worker = wave * gridDim.x + blockIdx.x;
  const uint64_t per_theta = static_cast<uint64_t>(n_lin) * n_lin;
  const uint32_t th_mid = (a.th_end - a.th_begin - 1u) / 2u;
  const uint32_t home_shard = blockIdx.x % kItemShards;
  const double inv_scaled = geo.inv_scaled;

  // One record per ITEM (not per wave): a wave takes its first item by its index and
  // every further one from an atomic counter, so which wave ran an item leaves no
  // trace in the records or in anything reduced from them.
  double best_s = 0.0;       // `double best_score = 0;` (:83)
  double best_i = kNoIndex;
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  for (uint32_t item = worker; item < n_items;)
  {
    uint32_t t, pxi, pyi;
#ifdef NDT2D_LANE_TRACE
    {
      const unsigned long long now = wall_clock64();
      if (trace_items > 0 && now - trace_last > trace_max) trace_max = now - trace_last;
      if (trace_items > 0 && a.scores != nullptr && lane == 0)
      {
        a.scores[5 * 8192 + 150000 + trace_prev_item] = static_cast<double>(now - trace_last);   // item duration
        a.scores[5 * 8192 + 300000 + trace_prev_item] = static_cast<double>(trace_flagged);      // beams the pre-test flagged
      }
      trace_flagged = 0;
      trace_prev_item = item;
      trace_last = now;
      trace_last_start = now;
      ++trace_items;
      if (a.scores != nullptr && lane == 0) a.scores[5 * 8192 + item] = static_cast<double>(now);   // item start
    }
#endif
    const uint32_t whole_item = PARTS ? item / a.beam_parts : item;
    const uint32_t part = PARTS ? item - whole_item * a.beam_parts : 0u;
    item_place(whole_item, patches, patches_1d, th_mid, t, pxi, pyi);
    // (lane -> patch position per item, two instructions, rather than two registers held for good)
    uint32_t lane_here = lane;
    asm volatile("" : "+v"(lane_here));
    const uint32_t ix = pxi * kPatch + (lane_here >> 3);
    const uint32_t iy = pyi * kPatch + (lane_here & 7);
    // lanes beyond the lattice edge shadow the edge candidate and are dropped below
    const double dx = a.dlin[min(ix, n_lin - 1)];
    const double dy = a.dlin[min(iy, n_lin - 1)];
    const double dxy = packed_offset(dx, dy, inv_scaled);
    const double4 * __restrict__ row = outer + static_cast<size_t>(t) * a.n_beams;

    double sum = 0.0;
    SkipState skip = skip_state(0.0, geo.no_skip);
    NDT2D_PATH(200, 1);
    // Beams go by in chunks of 64.  First the wave asks, one beam per lane, whether the
    // patch as a whole can reach a distribution with that beam (patch_can_score: four LDS
    // reads for 64 beams); four beams out of five cannot, and the groups of eight
    // consecutive beams none of which can are passed over without a vector instruction.
    // The others take the per-lane look-up and, where it says so, the exact path -- in
    // beam order, so a lane's sum is built exactly as before.
    const double dxy_corner = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dxy)),
                                               __builtin_amdgcn_readfirstlane(__double2loint(dxy)));
    const uint32_t beams_begin = PARTS ? part * a.part_beams : 0u;
    const uint32_t beams_end = PARTS ? min(beams_begin + a.part_beams, a.n_beams) : a.n_beams;
    for (uint32_t b0 = beams_begin; b0 < beams_end; b0 += kWave)
    {
      uint64_t can_score = ~0ull;
      if (geo.box_span >= 0)
      {
        const double k = row[min(b0 + lane, a.n_beams - 1u)].z;
        can_score = __builtin_amdgcn_ballot_w64(patch_can_score(k + dxy_corner, geo.box_span));
        NDT2D_PATH(201, 1);
      }
#ifdef NDT2D_LANE_TRACE
      trace_flagged += static_cast<uint32_t>(__popcll(can_score));
#endif
      const uint32_t chunk_end = min(b0 + static_cast<uint32_t>(kWave), beams_end);
      uint32_t b = b0;
      for (; b + kUnroll <= chunk_end; b += kUnroll, can_score >>= kUnroll)
      {
        if ((can_score & ((1ull << kUnroll) - 1ull)) == 0ull) continue;
        double4 o[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) o[u] = row[b + u];
        // A wave with a flagged group in hand is on the launch's critical path (an item of
        // 720 exact evaluations takes a wave 0.2 ms even alone); the waves stepping over
        // unflagged groups or reducing an item's records are not: issue priority while the
        // group is evaluated (cfg-2: - 3 %).
        __builtin_amdgcn_s_setprio(3);
        lane_beams<kUnroll, POW2, LDS_RECORDS, true, COMPACT>(g, c, o, dx, dy, dxy, sum, skip, geo.no_skip);
        __builtin_amdgcn_s_setprio(0);
      }
      for (; b < chunk_end; ++b, can_score >>= 1)
      {
        if ((can_score & 1ull) == 0ull) continue;
        const double4 one[1] = {row[b]};
        lane_beams<1, POW2, LDS_RECORDS, true, COMPACT>(g, c, one, dx, dy, dxy, sum, skip, geo.no_skip);
      }
    }

    if (PARTS)
    {
      // the part's 64 partial sums, one coalesced store; match_lane_combine_kernel adds a
      // candidate's parts in order and does everything that follows from its score
      a.part_sums[static_cast<size_t>(item) * kWave + lane] = sum;
      uint32_t next = n_items;
      if (lane == kWave - 1)
      {
        for (uint32_t tried = 0; tried < kItemShards && next >= n_items; ++tried)
        {
          const uint32_t shard = (home_shard + tried) % kItemShards;
          const uint32_t k = atomicAdd(a.next_item + shard * kItemShardStride, 1u);
          const uint64_t candidate = static_cast<uint64_t>(n_workers) + static_cast<uint64_t>(k) * kItemShards + shard;
          if (candidate < n_items) next = static_cast<uint32_t>(candidate);
        }
      }
      item = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(next), kWave - 1));
      continue;
    }
    // (the candidate's lattice indices are made again here, from the lane number, rather than
    // held in registers across the beam loop: the kernel runs at 80 VGPRs and used to spill them)
    uint32_t lane_again = lane;
    asm volatile("" : "+v"(lane_again));
    const uint32_t ix_e = pxi * kPatch + (lane_again >> 3);
    const uint32_t iy_e = pyi * kPatch + (lane_again & 7);
    if ((ix_e < n_lin) & (iy_e < n_lin))
    {
      const uint32_t ix = ix_e, iy = iy_e;
      const double score = -sum;  // (:127)
      const uint64_t local = static_cast<uint64_t>(t) * per_theta + static_cast<uint64_t>(ix) * n_lin + iy;
      const double flat = static_cast<double>(
        static_cast<uint64_t>(a.th_begin + t * a.th_stride) * per_theta + static_cast<uint64_t>(ix) * n_lin + iy);
      if (DYNAMIC_ITEMS)
      {
        // (the lane's one candidate of this item: best_s is still 0)
        if (score < best_s)
        {
          best_s = score;
          best_i = flat;
        }
      }
      else if (score < 0.0)
      {
        merge_best(score, flat, best_s, best_i);   // the lane's running best over its items; marks near-ties
      }
      // k += x x^T score, u += x score, s += score (:137-140)
      const double dt = a.dth[a.th_begin + t * a.th_stride];
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
#if !defined(NDT2D_LANE_HIST) && !defined(NDT2D_LANE_TRACE) && !defined(NDT2D_LANE_PATHS)
      if (a.scores != nullptr) a.scores[local] = score;
#endif
    }

    if (!DYNAMIC_ITEMS)
    {
      item += n_workers;
      continue;
    }
    // wave-level reduction of the item's per-lane records: complete in lane 63, which
    // writes the item's record and fetches the wave's next item
    // (an item none of whose candidates reached a distribution: every lane holds the
    // record {0, no index, +0.0 x 10} already -- x * -0.0 added to +0.0 is +0.0 -- and so
    // would the reduction: skip it.  cfg-2 - 0.7 %, cfg-4 - 1.3 %)
    if (wave_any(sum != 0.0))
    {
      NDT2D_PATH(209, 1);
      wave_best_to_last_lane(best_s, best_i);
#pragma unroll
      for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);
    }

    uint32_t next = n_items;
    if (lane == kWave - 1)
    {
      double * out = a.partials + static_cast<size_t>(item) * kRecord;
      out[0] = best_s;
      out[1] = best_i;
#pragma unroll
      for (int k = 0; k < 10; ++k) out[2 + k] = acc[k];
      // Items beyond the first n_workers are dealt to kItemShards counters by
      // index; a wave draws from the shard of its block (blocks b, b + 8, ... are
      // observed to share an XCD) and, once that is empty, from the others.
      for (uint32_t tried = 0; tried < kItemShards && next >= n_items; ++tried)
      {
        const uint32_t shard = (home_shard + tried) % kItemShards;
        const uint32_t k = atomicAdd(a.next_item + shard * kItemShardStride, 1u);
        const uint64_t candidate = static_cast<uint64_t>(n_workers) + static_cast<uint64_t>(k) * kItemShards + shard;
        if (candidate < n_items) next = static_cast<uint32_t>(candidate);
      }
    }
    item = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(next), kWave - 1));
    best_s = 0.0;
    best_i = no_index_here();
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  }
#ifdef NDT2D_LANE_TRACE
  if (a.scores != nullptr && lane == 0)
  {
    const unsigned long long now = wall_clock64();
    if (trace_items > 0 && now - trace_last > trace_max) trace_max = now - trace_last;
    if (trace_items > 0) a.scores[5 * 8192 + 150000 + trace_prev_item] = static_cast<double>(now - trace_last);
    if (trace_items > 0) a.scores[5 * 8192 + 300000 + trace_prev_item] = static_cast<double>(trace_flagged);
    a.scores[5 * worker] = static_cast<double>(trace_t0);
    a.scores[5 * worker + 1] = static_cast<double>(now);
    a.scores[5 * worker + 2] = static_cast<double>(trace_items);
    a.scores[5 * worker + 3] = static_cast<double>(trace_max);
    a.scores[5 * worker + 4] = static_cast<double>(trace_last_start);
  }
#endif
  if (!DYNAMIC_ITEMS)
  {
    wave_best_to_last_lane(best_s, best_i);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);
    if (lane == kWave - 1)
    {
      double * out = a.partials + static_cast<size_t>(worker) * kRecord;
      out[0] = best_s;
      out[1] = best_i;
#pragma unroll
      for (int k = 0; k < 10; ++k) out[2 + k] = acc[k];
    }
  }
}

template <int THREADS, bool POW2, bool LDS_RECORDS, bool DYNAMIC_ITEMS>
__global__ void __launch_bounds__(THREADS) match_lane_kernel(
  const MatchArgs a, const double4 * __restrict__ outer, const uint8_t * __restrict__ map_image,
  const LaneGeom geo)
{
  match_lane_body<THREADS, POW2, LDS_RECORDS, DYNAMIC_ITEMS, false>(a, outer, map_image, geo);
}

// The compacted-records form: 768-thread blocks held to 80 VGPRs, two per CU = six waves
// per SIMD (a block's waves must spread evenly over the four SIMDs: 640-thread blocks
// would put six waves of two blocks on one SIMD, which its register file cannot hold).
__global__ void __launch_bounds__(kLaneThreadsCompact) __attribute__((amdgpu_waves_per_eu(NDT2D_LANE_COMPACT_WAVES_PER_EU)))
match_lane_compact_kernel(const MatchArgs a, const double4 * __restrict__ outer,
                          const uint8_t * __restrict__ map_image, const LaneGeom geo)
{
  match_lane_body<kLaneThreadsCompact, true, true, true, true>(a, outer, map_image, geo);
}

// Maps too large for LDS (records gathered from the 64-byte-stride HBM copy through L2):
// the LDS image is the occupancy map alone, so two 768-thread blocks fit a CU as well --
// six waves per SIMD to cover the gathers' latency instead of four.
template <bool POW2>
__global__ void __launch_bounds__(kLaneThreadsCompact) __attribute__((amdgpu_waves_per_eu(NDT2D_LANE_COMPACT_WAVES_PER_EU)))
match_lane_gather6_kernel(const MatchArgs a, const double4 * __restrict__ outer,
                          const uint8_t * __restrict__ map_image, const LaneGeom geo)
{
  match_lane_body<kLaneThreadsCompact, POW2, false, true, false>(a, outer, map_image, geo);
}

// The beam-part forms of the two six-wave kernels (mid-size lattices, MatchArgs::beam_parts).
__global__ void __launch_bounds__(kLaneThreadsCompact) __attribute__((amdgpu_waves_per_eu(NDT2D_LANE_COMPACT_WAVES_PER_EU)))
match_lane_compact_parts_kernel(const MatchArgs a, const double4 * __restrict__ outer,
                                const uint8_t * __restrict__ map_image, const LaneGeom geo)
{
  match_lane_body<kLaneThreadsCompact, true, true, true, true, true>(a, outer, map_image, geo);
}

template <bool POW2>
__global__ void __launch_bounds__(kLaneThreadsCompact) __attribute__((amdgpu_waves_per_eu(NDT2D_LANE_COMPACT_WAVES_PER_EU)))
match_lane_gather6_parts_kernel(const MatchArgs a, const double4 * __restrict__ outer,
                                const uint8_t * __restrict__ map_image, const LaneGeom geo)
{
  match_lane_body<kLaneThreadsCompact, POW2, false, true, false, true>(a, outer, map_image, geo);
}

// Second kernel of the beam-part form: a wave per (theta, patch) item, lane = candidate.
// The candidate's score is the in-order sum of its parts' in-order sums,
// ((p_0 + p_1) + p_2) + p_3; then the reference's strict-< best (:128-134) and the
// covariance accumulators (:137-140) exactly as the unsplit search's epilogue has them,
// one 12-double record per item.
__global__ void __launch_bounds__(256) match_lane_combine_kernel(const MatchArgs a)
{
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t n_lin = a.n_lin;
  const uint32_t patches_1d = (n_lin + kPatch - 1) / kPatch;
  const uint32_t patches = patches_1d * patches_1d;
  const uint32_t n_items = (a.th_end - a.th_begin) * patches;
  const uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const uint32_t th_mid = (a.th_end - a.th_begin - 1u) / 2u;
  uint32_t t, pxi, pyi;
  item_place(item, patches, patches_1d, th_mid, t, pxi, pyi);
  const uint32_t ix = pxi * kPatch + (lane >> 3);
  const uint32_t iy = pyi * kPatch + (lane & 7);
  const bool valid = (ix < n_lin) & (iy < n_lin);
  const double * p = a.part_sums + static_cast<size_t>(item) * a.beam_parts * kWave + lane;
  double sum = p[0];
  for (uint32_t j = 1; j < a.beam_parts; ++j) sum += p[static_cast<size_t>(j) * kWave];
  double best_s = 0.0;       // `double best_score = 0;` (:83)
  double best_i = kNoIndex;
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  if (valid)
  {
    const double dx = a.dlin[ix];
    const double dy = a.dlin[iy];
    const double score = -sum;  // (:127)
    const uint64_t per_theta = static_cast<uint64_t>(n_lin) * n_lin;
    const uint64_t in_theta = static_cast<uint64_t>(ix) * n_lin + iy;
    const uint32_t ith = a.th_begin + t * a.th_stride;
    if (score < 0.0)
    {
      best_s = score;
      best_i = static_cast<double>(static_cast<uint64_t>(ith) * per_theta + in_theta);
    }
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
#ifndef NDT2D_LANE_TRACE
    if (a.scores != nullptr) a.scores[static_cast<uint64_t>(t) * per_theta + in_theta] = score;
#endif
  }
  wave_best_to_last_lane(best_s, best_i);
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);
  if (lane == kWave - 1)
  {
    double * out = a.partials + static_cast<size_t>(item) * kRecord;
    out[0] = best_s;
    out[1] = best_i;
#pragma unroll
    for (int k = 0; k < 10; ++k) out[2 + k] = acc[k];
  }
}

bool pow2_grid(const MatchArgs & args) { return args.grid.pow2 != 0; }

// Beam parts of a lattice (see MatchArgs::beam_parts): by the WHOLE lattice's work items, so
// that every shard of a search cuts a candidate's beams alike.
uint32_t lane_beam_parts(const MatchArgs & args, uint32_t * part_beams)
{
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t whole = static_cast<uint64_t>(args.n_th) * p1 * p1;
  uint32_t parts = whole < kPartsFourBelow ? 4u : (whole < kPartsTwoBelow ? 2u : 1u);
  if (const char * env = std::getenv("NDT2D_LANE_PARTS"))   // A/B knob (1: never cut)
  {
    const int v = std::atoi(env);
    if (v == 1 || v == 2 || v == 4 || v == 8) parts = static_cast<uint32_t>(v);
  }
  const uint32_t chunks = (args.n_beams + kWave - 1) / kWave;
  if (args.n_beams < kDynamicItemsFromBeams || parts > chunks) parts = 1;
  const uint32_t per_part = (chunks + parts - 1) / parts;
  *part_beams = per_part * kWave;
  return (chunks + per_part - 1) / per_part;
}

bool lane_records_in_lds(const MatchArgs & args, size_t map_bytes, size_t lds_per_block)
{
  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  return grid_bytes + map_bytes <= lds_per_block;
}

}  // namespace

bool lane_slabs(const MatchArgs & args, uint32_t * slab_th, uint32_t * n_slabs)
{
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t per_theta = p1 * p1;
  const uint64_t th_total = args.th_end - args.th_begin;
  if (per_theta == 0 || th_total == 0 || per_theta > kMaxLaneItems) return false;
  uint64_t slab_items = kLaneSlabItems;
  if (const char * env = std::getenv("NDT2D_LANE_SLAB_ITEMS"))   // tests: slabs on small lattices
  {
    const long long v = std::atoll(env);
    if (v > 0 && static_cast<uint64_t>(v) <= kMaxLaneItems) slab_items = static_cast<uint64_t>(v);
  }
  uint64_t th = slab_items / per_theta;
  if (th < 1) th = 1;
  const uint64_t at_least = (th_total + kMaxLaneSlabs - 1) / kMaxLaneSlabs;
  if (th < at_least) th = at_least;
  if (th > th_total) th = th_total;
  if (th * per_theta > kMaxLaneItems) return false;
  *slab_th = static_cast<uint32_t>(th);
  *n_slabs = static_cast<uint32_t>((th_total + th - 1) / th);
  return true;
}

size_t match_lane_outer_doubles(const MatchArgs & args)
{
  // rotated-beam table of one slab + the occupancy-map image (at most 256 x 256 bytes)
  uint32_t slab_th = args.th_end - args.th_begin, n_slabs = 1;
  (void)lane_slabs(args, &slab_th, &n_slabs);
  uint32_t part_beams = 0;
  const uint32_t parts = lane_beam_parts(args, &part_beams);
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const size_t part_sums = parts > 1 ? static_cast<size_t>(slab_th) * p1 * p1 * parts * kWave : 0;
  return static_cast<size_t>(slab_th) * args.n_beams * 4 +
         static_cast<size_t>(kMapStride) * kMaxMapCells / sizeof(double) + part_sums;
}

bool match_lane_supported(const MatchArgs & args, size_t lds_per_block)
{
  LaneGeom geo;
  size_t map_bytes = 0;
  if (args.grid.occ_bits == nullptr || !lane_geometry(args, lds_per_block, &geo, &map_bytes)) return false;
  uint32_t slab_th = 0, n_slabs = 0;
  // (24-bit multiplies index the grid rows and the LDS records)
  return map_bytes <= lds_per_block && lane_slabs(args, &slab_th, &n_slabs) &&
         args.grid.size_x < (1u << 24) && args.grid.ncell < (1u << 24);
}

hipError_t launch_match_lane(const MatchArgs & args_in, double * outer, double * workspace,
                             uint32_t max_workers, int cus, size_t lds_per_block, bool no_skip,
                             hipStream_t stream, hipEvent_t ev_after_pre_kernel,
                             uint32_t * n_workers_out, int * records_mode_out, uint32_t * parts_out)
{
  MatchArgs args = args_in;
  args.partials = workspace;
  LaneGeom geo;
  size_t map_bytes = 0;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes)) return hipErrorInvalidValue;
  geo.no_skip = no_skip ? 1 : 0;
  {
    // the control mode evaluates every term; NDT2D_LANE_PRETEST=0 is the A/B switch
    static const bool pretest = [] {
      const char * v = std::getenv("NDT2D_LANE_PRETEST");
      return v == nullptr || v[0] != '0';
    }();
    if (no_skip || !pretest) geo.box_span = -1;
  }
  // Items of few beams are too short to repay a reduction, a record and an atomic
  // each (100 beams: static assignment is ~15 % faster; 720 beams: dynamic is 17 %
  // faster, experiments/small_search_sweep.py).
  const bool dynamic_items = args.n_beams >= kDynamicItemsFromBeams;
  const uint32_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t n_items = static_cast<uint64_t>(args.th_end - args.th_begin) * p1 * p1;

  const uint64_t n_outer = static_cast<uint64_t>(args.th_end - args.th_begin) * args.n_beams;
  uint32_t oblocks = static_cast<uint32_t>((n_outer + 255) / 256);
  if (oblocks > 4096) oblocks = 4096;
  if (oblocks < 64) oblocks = 64;  // the map build wants a few thousand threads as well
  uint8_t * map_image = reinterpret_cast<uint8_t *>(outer + n_outer * 4);
  // mid-size lattices: a candidate's beams in parts (decided below, once the kernel form is known)
  uint32_t part_beams = 0;
  uint32_t beam_parts = dynamic_items ? lane_beam_parts(args, &part_beams) : 1u;
  args.part_sums = outer + n_outer * 4 + static_cast<size_t>(kMapStride) * kMaxMapCells / sizeof(double);

  // (NDT2D_LANE_RECORDS=global: gather the records from HBM although they would fit LDS --
  // what a map too large for LDS costs, measured on a workload that has both forms)
  const char * knob_rec = std::getenv("NDT2D_LANE_RECORDS");
  const bool lds_records = lane_records_in_lds(args, map_bytes, lds_per_block) &&
                           !(knob_rec != nullptr && knob_rec[0] == 'g');
  size_t lds_bytes =
    map_bytes +
    (lds_records ? static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double) : 0);
  // Compacted records: the image shrinks enough for two blocks per CU (A/B knob:
  // NDT2D_LANE_COMPACT=0 keeps the whole-grid image and one 1024-thread block).
  const size_t compact_bytes =
    compact_rank_bytes(args.grid.ncell) +
    (static_cast<size_t>(args.grid.n_occ) + 1) * kCellDoubles * sizeof(double);
  const char * knob = std::getenv("NDT2D_LANE_COMPACT");
  const bool compact_possible = dynamic_items && pow2_grid(args) && lds_records && args.grid.n_occ > 0 &&
                                args.grid.compact_records != nullptr &&
                                2 * (map_bytes + compact_bytes) <= lds_per_block &&
                                !(knob != nullptr && knob[0] == '0');
  // records gathered from HBM: the same block geometry when two maps fit a CU
  // (NDT2D_LANE_GATHER6=0: the one-block form, for A/B runs)
  const char * knob6 = std::getenv("NDT2D_LANE_GATHER6");
  const bool gather6_possible = dynamic_items && !lds_records && 2 * map_bytes <= lds_per_block &&
                                !(knob6 != nullptr && knob6[0] == '0');
  // The beam-part form exists for the two six-wave kernels.  Whether a candidate's beams are
  // cut decides the bits of its score, so it must not depend on THIS launch's share of the
  // lattice: everything above follows from the grid, the scan and the whole lattice (the map
  // geometry included, lane_geometry), and a cut search takes a six-wave kernel however
  // few items the launch holds.
  if (!(compact_possible || gather6_possible)) beam_parts = 1;
  args.beam_parts = beam_parts;
  args.part_beams = part_beams;
  const uint64_t n_sub_items = n_items * beam_parts;

  // Few work items (the plugin's default lattice is 720 of them): 256-thread blocks
  // put them on four times as many CUs -- as long as every wave still gets at most
  // one item; the LDS image allows one block per CU, so beyond that the larger block
  // is what puts more waves on a CU.
  const bool small = beam_parts == 1 && n_items <= static_cast<uint64_t>(cus) * (kLaneThreadsSmall / kWave);
  const uint32_t waves_per_block = (small ? kLaneThreadsSmall : kLaneThreads) / kWave;
  uint32_t blocks = static_cast<uint32_t>((n_items + waves_per_block - 1) / waves_per_block);
  uint32_t max_blocks = static_cast<uint32_t>(cus);
  if (max_blocks * waves_per_block > max_workers) max_blocks = max_workers / waves_per_block;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  const bool compact = !small && compact_possible;
  const bool gather6 = !compact && !small && gather6_possible;
  // 0: records gathered from HBM, 1: the whole grid's records in LDS, 2: compacted records in LDS
  // (+4: the map is one byte per block of grid cells)
  if (records_mode_out != nullptr)
  {
    *records_mode_out = (compact ? 2 : (lds_records ? 1 : 0)) | (geo.block_log2 > 0 ? 4 : 0);
  }
  if (compact || gather6)
  {
    lds_bytes = compact ? map_bytes + compact_bytes : map_bytes;
    const uint32_t wpb = kLaneThreadsCompact / kWave;
    blocks = static_cast<uint32_t>((n_sub_items + wpb - 1) / wpb);
    uint32_t cap = 2 * static_cast<uint32_t>(cus);
    if (cap * wpb > max_workers) cap = max_workers / wpb;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
  }
  // the table pre-kernel (it also clears the item counters of the search that follows)
  hipLaunchKernelGGL(outer_table_kernel, dim3(oblocks), dim3(256), 0, stream, args,
                     reinterpret_cast<double4 *>(outer), map_image, geo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ev_after_pre_kernel != nullptr)
  {
    e = hipEventRecord(ev_after_pre_kernel, stream);
    if (e != hipSuccess) return e;
  }
  auto launch = [&](auto kernel, int threads) -> hipError_t {
    // (no static LDS in front of the map: prepare_absolute_lds_kernel, ndt2d_kernels.h)
    const hipError_t e2 = prepare_absolute_lds_kernel(reinterpret_cast<const void *>(kernel), lds_bytes);
    if (e2 != hipSuccess) return e2;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, args,
                       reinterpret_cast<const double4 *>(outer),
                       static_cast<const uint8_t *>(map_image), geo);
    return hipGetLastError();
  };
  const bool pow2 = args.grid.pow2 != 0;
  auto pick = [&](auto threads_tag) -> hipError_t {
    constexpr int T = decltype(threads_tag)::value;
    auto with_items = [&](auto dyn_tag) -> hipError_t {
      constexpr bool D = decltype(dyn_tag)::value;
      return lds_records ? (pow2 ? launch(match_lane_kernel<T, true, true, D>, T)
                                 : launch(match_lane_kernel<T, false, true, D>, T))
                         : (pow2 ? launch(match_lane_kernel<T, true, false, D>, T)
                                 : launch(match_lane_kernel<T, false, false, D>, T));
    };
    return dynamic_items ? with_items(std::true_type{}) : with_items(std::false_type{});
  };
  if (beam_parts > 1)
  {
    e = compact ? launch(match_lane_compact_parts_kernel, kLaneThreadsCompact)
                : (pow2 ? launch(match_lane_gather6_parts_kernel<true>, kLaneThreadsCompact)
                        : launch(match_lane_gather6_parts_kernel<false>, kLaneThreadsCompact));
    if (e == hipSuccess)
    {
      hipLaunchKernelGGL(match_lane_combine_kernel, dim3(static_cast<uint32_t>((n_items + 3) / 4)), dim3(256),
                         0, stream, args);
      e = hipGetLastError();
    }
  }
  else if (compact)
  {
    e = launch(match_lane_compact_kernel, kLaneThreadsCompact);
  }
  else if (gather6)
  {
    e = pow2 ? launch(match_lane_gather6_kernel<true>, kLaneThreadsCompact)
             : launch(match_lane_gather6_kernel<false>, kLaneThreadsCompact);
  }
  else
  {
    e = small ? pick(std::integral_constant<int, kLaneThreadsSmall>{})
              : pick(std::integral_constant<int, kLaneThreads>{});
  }
  if (parts_out != nullptr) *parts_out = beam_parts;
  if (n_workers_out != nullptr)
  {
    *n_workers_out = dynamic_items ? static_cast<uint32_t>(n_items) : blocks * waves_per_block;
  }
  return e;
}

}  // namespace ndt2d
