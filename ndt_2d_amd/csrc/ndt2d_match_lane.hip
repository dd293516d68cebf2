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
// The map is padded by more cells than any offset reaches and beams are
// pre-clamped, so no per-lane range check exists.  The map is kept at up to
// 4 x 4 sub-cells per cell.  Map byte: bit 0 = the cell holds a distribution;
// bits 2..7 = a level: 0 if no distribution can be hit from this sub-cell (its
// box, slightly widened because the fixed-point coordinate is rounded, touches no
// occupied cell), else an upper bound l - 63 of Cell::score's exponent over that
// box (63: no claim).  Each lane carries the level below which a term cannot
// change its sum (RN(s + exp(e)) == s); if every lane's byte is below its level
// the beam is skipped.  Otherwise lanes within 4 fixed-point units of a boundary
// are "near"; if any live lane is occupied or near, the wave runs the exact
// reference arithmetic (points_inner :121-125, NDT::getIndex
// src/ndt_model.cpp:203-218 for near lanes, Cell::score :105-116).  Every
// skipped term is one that leaves the sum unchanged, so the sums are
// bit-identical to the unskipped evaluation (variant "lane-noskip" is that
// evaluation; the tests compare the two bitwise).
#include <cmath>

#include <type_traits>

#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

constexpr uint32_t kDynamicItemsFromBeams = 256;
constexpr int kLaneThreads = 1024;       // large searches: one block per CU
constexpr int kLaneThreadsSmall = 256;   // small searches: spread the few work items over more CUs
constexpr int kPatch = 8;            // patch is kPatch x kPatch candidates = one wave
constexpr int kUnroll = 8;  // beams per look-up group
// map coordinates are 8.16 fixed point
constexpr double kFracScale = 65536.0;
constexpr int kMapStride = 256;      // map row stride in bytes = 2^8 cells
constexpr int kMaxMapCells = 256;    // cell coordinate is one byte
constexpr uint32_t kNearUnits = 4;   // guard band around cell boundaries, in 2^-16 cells
// widening of a map sub-cell's box (fraction of a sub-cell) when its exponent bound
// is taken: 16 x the rounding of the fixed-point coordinate that selects it
constexpr double kBoxMargin = 1.0 / 1024.0;
constexpr double kTwo24 = 16777216.0;
constexpr double kTwo52 = 4503599627370496.0;
constexpr double kNearBias = kNearUnits * (kTwo24 + 1.0);  // +kNearUnits on both packed fractions
static_assert(kNearUnits == 4, "the near test masks bits 3..15 of the biased fractions");

struct LaneGeom
{
  int32_t pad;     // border cells on every side of the window in the map
  // The map is kept at 2^sub_log2 sub-cells per cell (as fine as the one-byte
  // coordinate and LDS allow): the finer the sub-cell, the tighter its bound on the
  // exponent, so fewer beams next to walls take the exact path.
  int32_t sub_log2;
  double unit_scale;  // fixed-point units per cell = 2^16 << sub_log2
  int32_t map_h;   // ((win_h + 2 * pad) << sub_log2) rows of kMapStride bytes
  // Window of grid cells the map covers: every point this search can produce
  // (scan pose +- (longest beam + largest offset)) lies inside it or outside the
  // grid.  For small grids it is the whole grid.
  int32_t win_x0, win_y0, win_w, win_h;
  double k_min, k_max_x, k_max_y;  // clamp of the per-beam fixed-point coordinate
  int32_t no_skip;  // control mode: every beam takes the exact path
};

// Upper bound of Cell::score's exponent e(p) = q^T h q, q = p - mean (h = -0.5 *
// information, packed record rec) over the box [x0, x1] x [y0, y1].  For a
// negative definite h the form is concave: its maximum over the box is 0 at the
// mean if the box holds it, else it lies on one of the four edges, where the
// restriction is a concave parabola whose clamped vertex is found in closed form.
// Anything else (NaN / degenerate information) returns +inf: no claim.  The slack
// covers the rounding of this evaluation and of the reference's own.
__device__ __forceinline__ double exponent_upper_bound(const double * rec, double x0, double x1,
                                                       double y0, double y1)
{
  const double mx = rec[0], my = rec[1], h00 = rec[2], h01 = rec[3], h11 = rec[4];
  if (!(h00 < 0.0 && h11 < 0.0 && h00 * h11 - h01 * h01 > 0.0)) return HUGE_VAL;
  const double a0 = x0 - mx, a1 = x1 - mx, b0 = y0 - my, b1 = y1 - my;
  double best;
  if (a0 <= 0.0 && a1 >= 0.0 && b0 <= 0.0 && b1 >= 0.0)
  {
    best = 0.0;
  }
  else
  {
    best = -HUGE_VAL;
    const double qa[2] = {a0, a1}, qb[2] = {b0, b1};
#pragma unroll
    for (int k = 0; k < 2; ++k)
    {
      // edge x = const: maximise over q1 in [b0, b1]
      double q0 = qa[k];
      double q1 = fmin(fmax(-h01 * q0 / h11, b0), b1);
      best = fmax(best, h00 * q0 * q0 + 2.0 * h01 * q0 * q1 + h11 * q1 * q1);
      // edge y = const: maximise over q0 in [a0, a1]
      q1 = qb[k];
      q0 = fmin(fmax(-h01 * q1 / h00, a0), a1);
      best = fmax(best, h00 * q0 * q0 + 2.0 * h01 * q0 * q1 + h11 * q1 * q1);
    }
  }
  const double am = fmax(fabs(a0), fabs(a1)), bm = fmax(fabs(b0), fabs(b1));
  const double magnitude = fabs(h00) * am * am + 2.0 * fabs(h01) * am * bm + fabs(h11) * bm * bm;
  return best + (1e-9 * magnitude + 1e-6);
}

// points_outer for the slab (reference :106-115) plus the packed fixed-point
// map coordinate of each rotated beam:
//   outer[t][b] = {ox, oy, K, 0},   K = 2^52 + ky * 2^24 + kx,
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
    const int32_t sx = static_cast<int32_t>(g.size_x), sy = static_cast<int32_t>(g.size_y);
    const uint32_t n_map = static_cast<uint32_t>(kMapStride) * geo.map_h;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_map; i += gridDim.x * 256)
    {
      // map sub-cell (mx, my): its box in the world, widened by kBoxMargin of a
      // sub-cell so that it holds every point whose rounded fixed-point coordinate
      // lands in this sub-cell, and the grid cells that widened box overlaps (its
      // own, plus the neighbours across a cell boundary it touches)
      const int32_t mx = static_cast<int32_t>(i & (kMapStride - 1));
      const int32_t my = static_cast<int32_t>(i >> 8);
      const int32_t sub = 1 << geo.sub_log2;
      const double sub_size = g.cell_size / static_cast<double>(sub);
      const int32_t cx = (mx >> geo.sub_log2) - geo.pad + geo.win_x0;
      const int32_t cy = (my >> geo.sub_log2) - geo.pad + geo.win_y0;
      const int32_t lx = mx & (sub - 1), ly = my & (sub - 1);
      const double x0 = g.origin_x + (static_cast<double>(cx) * sub + lx - kBoxMargin) * sub_size;
      const double y0 = g.origin_y + (static_cast<double>(cy) * sub + ly - kBoxMargin) * sub_size;
      const double x1 = x0 + (1.0 + 2.0 * kBoxMargin) * sub_size;
      const double y1 = y0 + (1.0 + 2.0 * kBoxMargin) * sub_size;
      uint32_t self = 0;
      double bound = -HUGE_VAL;
      for (int32_t b = (ly == 0 ? -1 : 0); b <= (ly == sub - 1 ? 1 : 0); ++b)
      {
        for (int32_t c = (lx == 0 ? -1 : 0); c <= (lx == sub - 1 ? 1 : 0); ++c)
        {
          const int32_t nx = cx + c, ny = cy + b;
          if (nx >= 0 && nx < sx && ny >= 0 && ny < sy)
          {
            const uint32_t cell = static_cast<uint32_t>(ny * sx + nx);
            const uint32_t o = (g.occ_bits[cell >> 5] >> (cell & 31u)) & 1u;
            if (b == 0 && c == 0) self = o;
            if (o != 0)
            {
              const double e = exponent_upper_bound(g.cells_lds_image + static_cast<size_t>(cell) * kCellDoubles,
                                                    x0, x1, y0, y1);
              bound = !(e <= bound) ? e : bound;   // NaN-propagating max
            }
          }
        }
      }
      // level 0: no distribution can be hit from here; level l >= 1: every exponent
      // reachable from this sub-cell is <= l - 63 (level 63: no claim)
      uint32_t level = 0;
      if (bound > -HUGE_VAL || bound != bound)
      {
        level = !(bound <= 0.0) ? 63u : (bound < -62.0 ? 1u : static_cast<uint32_t>(63 + static_cast<int32_t>(ceil(bound))));
      }
      map_out[i] = static_cast<uint8_t>(self | (level != 0 ? 2u : 0u) | (level << 2));
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
    o.z = kTwo52 + (rint(ky) * kTwo24 + rint(kx));
    o.w = 0.0;
    outer[i] = o;
  }
}

struct LaneCtx
{
  uint32_t lds_cells_address;  // LDS byte address of the packed records (behind the map)
  uint32_t sub_log2;           // map sub-cell -> map cell: shift right
  // map cell (col, row) -> grid cell index: row * size_x + col - idx_bias,
  // idx_bias = (pad - win_y0) * size_x + (pad - win_x0), modulo 2^32
  uint32_t idx_bias;
  uint32_t size_x;
};

// U consecutive beams of one patch; o[] holds their table rows, dxy the lane's
// packed fixed-point offset.
// The map sits at LDS address 0 (the kernel's only LDS is its dynamic block, whose
// first bytes are the map; checked at kernel entry), so the packed cell bytes ARE
// the address: reading through an absolute LDS pointer saves the per-beam
// v_add_u32 of the array base that indexing a __shared__ array costs.
__device__ __forceinline__ uint32_t lds_byte_at(uint32_t address)
{
  typedef const __attribute__((address_space(3))) uint8_t * lds_byte_ptr;
  return *reinterpret_cast<lds_byte_ptr>(address);
}

// Cell::score's exponent against packed record idx of the LDS copy, addressed with
// 32-bit LDS arithmetic (one v_mad_u32_u24; idx < 2^24 for any grid that fits LDS).
__device__ __forceinline__ double lds_record_exponent(uint32_t cells_address, uint32_t idx,
                                                      double px, double py)
{
  typedef const __attribute__((address_space(3), aligned(16))) double * lds_double_ptr;
  const lds_double_ptr rec = reinterpret_cast<lds_double_ptr>(
    __umul24(idx, static_cast<uint32_t>(kCellDoubles * sizeof(double))) + cells_address);
  return record_exponent(rec[0], rec[1], rec[2], rec[3], rec[4], px, py);
}

// Per-lane skip state.  Terms whose exponent is below skip_below cannot change
// the lane's sum (bit-exact skip); skip_level is the same threshold on the map's
// scale, pre-shifted to the byte's layout: a map byte below it promises an
// exponent below skip_below.  Both only ever rise with the sum, so they are
// refreshed after a group of beams that added something, not per beam.
struct SkipState
{
  double skip_below;
  uint32_t skip_level;
};

__device__ __forceinline__ SkipState skip_state(double sum, int32_t no_skip)
{
  SkipState s;
  if (no_skip)
  {
    s.skip_below = -HUGE_VAL;
    s.skip_level = 0;
    return s;
  }
  s.skip_below = negligible_below(sum);
  // level l promises e <= l - 63; l <= ceil(T) + 62 then gives e <= ceil(T) - 1 < T
  const int32_t l = static_cast<int32_t>(ceil(s.skip_below)) + 63;
  s.skip_level = static_cast<uint32_t>(min(max(l, 1), 63)) << 2;
  return s;
}

template <int U, bool POW2, bool LDS_RECORDS>
__device__ __forceinline__ void lane_beams(const GridDesc & g, const LaneCtx & c,
                                           const double4 (&o)[U], double dx, double dy,
                                           double dxy, double & sum, SkipState & skip,
                                           int32_t no_skip)
{
  const double skip_below = skip.skip_below;
  const uint32_t skip_level = skip.skip_level;
  uint32_t lo[U], hi[U], m[U];
  uint32_t top = 0;
#pragma unroll
  for (int u = 0; u < U; ++u)
  {
    const double s = o[u].z + dxy;  // exact: integers below 2^53
    lo[u] = static_cast<uint32_t>(__double2loint(s));
    hi[u] = static_cast<uint32_t>(__double2hiint(s));
    // byte 0 <- lo.byte2 (cell x), byte 1 <- hi.byte1 (cell y), bytes 2,3 <- 0
    m[u] = lds_byte_at(__builtin_amdgcn_perm(hi[u], lo[u], 0x0c0c0502u));
    top = max(top, m[u]);
  }
  if (wave_any(top >= skip_level))
  {
    bool added = false;
    // have the beams' end points on their way before the first exact evaluation needs
    // them (left to itself the compiler loads each pair inside its own branch)
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : : "s"(o[u].x), "s"(o[u].y));
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      // Lanes below their level are negligible whatever cell they are in: they do
      // not keep the wave on this path.  (If the wave stays for another lane, their
      // term is still evaluated exactly -- and, being negligible, changes nothing.)
      // The wave-level tests combine the compares' lane masks as 64-bit integers in
      // scalar registers; a ballot of a combined bool would round-trip through a
      // vector register.
      const uint64_t live_mask = __builtin_amdgcn_ballot_w64(m[u] >= skip_level);
      if (live_mask != 0ull)
      {
        // within kNearUnits of a unit boundary on either axis: one more exact f64 add
        // biases both 16-bit fractions at once, then (frac + 4) mod 2^16 < 8 is
        // "bits 3..15 clear"; the y fraction straddles the two words (bytes 3, 4)
        const double sn = __hiloint2double(static_cast<int>(hi[u]), static_cast<int>(lo[u])) + kNearBias;
        const uint32_t nlo = static_cast<uint32_t>(__double2loint(sn));
        const uint32_t nhi = static_cast<uint32_t>(__double2hiint(sn));
        const uint32_t tx = nlo & 0xfff8u;
        const uint32_t ty = __builtin_amdgcn_perm(nhi, nlo, 0x0c0c0403u) & 0xfff8u;
        const bool occ = (m[u] & 1u) != 0;
        const uint64_t near_mask =
          (__builtin_amdgcn_ballot_w64(min(tx, ty) == 0u) | (no_skip != 0 ? ~0ull : 0ull)) & live_mask;
        const uint64_t occ_mask = __builtin_amdgcn_ballot_w64(occ) & live_mask;
        if ((occ_mask | near_mask) != 0ull)
        {
          // points_inner (:121-125) and Cell::score, exact
          const double px = o[u].x + dx;
          const double py = o[u].y + dy;
          uint32_t idx;
          if (near_mask != 0ull)
          {
            idx = cell_index<POW2>(g, px, py);
          }
          else
          {
            // interior of a cell: the look-up cell is the reference's cell,
            // row * size_x + column with the window's offset folded into idx_bias
            const uint32_t col = ((lo[u] >> 16) & 0xffu) >> c.sub_log2;
            const uint32_t row = ((hi[u] >> 8) & 0xffu) >> c.sub_log2;
            idx = occ ? __umul24(row, c.size_x) + (col - c.idx_bias) : g.ncell;
          }
          const double e = LDS_RECORDS ? lds_record_exponent(c.lds_cells_address, idx, px, py)
                                       : indexed_exponent<false>(g, nullptr, idx, px, py);
          // !(e < bound) also keeps NaN exponents (degenerate cells) on the exact path
          if (wave_any(!(e < skip_below)))
          {
            sum += exp_score(e);
            added = true;
          }
        }
      }
    }
    if (added) skip = skip_state(sum, no_skip);
  }
}

template <int THREADS, bool POW2, bool LDS_RECORDS, bool DYNAMIC_ITEMS>
__global__ void __launch_bounds__(THREADS) match_lane_kernel(
  const MatchArgs a, const double4 * __restrict__ outer, const uint8_t * __restrict__ map_image,
  const LaneGeom geo)
{
  // LDS image: padded occupancy map of the window (at offset 0, so the packed
  // cell bytes are the LDS address) followed, if they fit (LDS_RECORDS), by the
  // packed cell records of the whole grid; otherwise records are gathered from
  // the 64-byte-stride HBM copy (a patch touches 1-4 lines).
  extern __shared__ __align__(16) double lds[];
  const GridDesc & g = a.grid;
  uint8_t * lds_map = reinterpret_cast<uint8_t *>(lds);
  // lds_byte_at() addresses the map absolutely: it must start at LDS offset 0
  if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
  double * lds_cells = lds + (static_cast<size_t>(geo.map_h) * kMapStride) / sizeof(double);

  if (LDS_RECORDS) stage_grid_to_lds(g, lds_cells);
  {
    // the map image was built by the pre-kernel; kMapStride * map_h is a multiple of 16
    const uint32_t n16 = static_cast<uint32_t>(kMapStride) * geo.map_h / 16;
    const uint4 * src = reinterpret_cast<const uint4 *>(map_image);
    uint4 * dst = reinterpret_cast<uint4 *>(lds_map);
    for (uint32_t i = threadIdx.x; i < n16; i += THREADS) dst[i] = src[i];
  }
  __syncthreads();

  LaneCtx c;
  c.lds_cells_address = static_cast<uint32_t>(geo.map_h) * kMapStride;
  c.sub_log2 = static_cast<uint32_t>(geo.sub_log2);
  c.idx_bias = static_cast<uint32_t>(geo.pad - geo.win_y0) * g.size_x +
               static_cast<uint32_t>(geo.pad - geo.win_x0);
  c.size_x = g.size_x;

  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lx = lane >> 3, ly = lane & 7;
  const uint32_t n_lin = a.n_lin;
  const uint32_t patches_1d = (n_lin + kPatch - 1) / kPatch;
  const uint32_t patches = patches_1d * patches_1d;
  const uint32_t n_items = (a.th_end - a.th_begin) * patches;
  constexpr uint32_t kLaneWaves = THREADS / kWave;
  const uint32_t n_workers = gridDim.x * kLaneWaves;
  const uint32_t worker = wave * gridDim.x + blockIdx.x;
  const uint64_t per_theta = static_cast<uint64_t>(n_lin) * n_lin;
  const uint32_t th_mid = (a.th_end - a.th_begin - 1u) / 2u;
  const uint32_t home_shard = blockIdx.x % kItemShards;
  const double inv_scaled = g.inv_cell_size * geo.unit_scale;

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

    // theta steps are visited from the middle of the range outwards: the steps around
    // the scan's own heading are the expensive ones when the guess is any good, and
    // the last items a launch hands out should be cheap ones
    const uint32_t rank = item / patches;
    const uint32_t p = item - rank * patches;
    const uint32_t t = (rank & 1u) ? th_mid + (rank + 1u) / 2u : th_mid - rank / 2u;
    const uint32_t pxi = p / patches_1d;
    const uint32_t pyi = p - pxi * patches_1d;
    const uint32_t ix = pxi * kPatch + lx;
    const uint32_t iy = pyi * kPatch + ly;
    const bool valid = (ix < n_lin) & (iy < n_lin);
    // lanes beyond the lattice edge shadow the edge candidate and are dropped below
    const double dx = a.dlin[min(ix, n_lin - 1)];
    const double dy = a.dlin[min(iy, n_lin - 1)];
    const double dxy = rint(dy * inv_scaled) * kTwo24 + rint(dx * inv_scaled);
    const double4 * __restrict__ row = outer + static_cast<size_t>(t) * a.n_beams;

    double sum = 0.0;
    SkipState skip = skip_state(0.0, geo.no_skip);
    uint32_t b = 0;
    for (; b + kUnroll <= a.n_beams; b += kUnroll)
    {
      double4 o[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) o[u] = row[b + u];
      lane_beams<kUnroll, POW2, LDS_RECORDS>(g, c, o, dx, dy, dxy, sum, skip, geo.no_skip);
    }
    for (; b < a.n_beams; ++b)
    {
      const double4 one[1] = {row[b]};
      lane_beams<1, POW2, LDS_RECORDS>(g, c, one, dx, dy, dxy, sum, skip, geo.no_skip);
    }

    if (valid)
    {
      const double score = -sum;  // (:127)
      const uint64_t local = static_cast<uint64_t>(t) * per_theta + static_cast<uint64_t>(ix) * n_lin + iy;
      const double flat = static_cast<double>(
        static_cast<uint64_t>(a.th_begin + t * a.th_stride) * per_theta + static_cast<uint64_t>(ix) * n_lin + iy);
      if (score < best_s)
      {
        best_s = score;
        best_i = flat;
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
      if (a.scores != nullptr) a.scores[local] = score;
    }

    if (!DYNAMIC_ITEMS)
    {
      item += n_workers;
      continue;
    }
    // wave-level reduction of the item's per-lane records: complete in lane 63, which
    // writes the item's record and fetches the wave's next item
    wave_best_to_last_lane(best_s, best_i);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);

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
    best_i = kNoIndex;
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  }
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

// Window of grid cells [lo, hi] reachable along one axis, clipped to the grid.
bool axis_window(double pose, double reach, double origin, double inv_cell, uint32_t size,
                 int32_t * lo, int32_t * n)
{
  const double a = std::floor((pose - reach - origin) * inv_cell) - 1.0;
  const double b = std::floor((pose + reach - origin) * inv_cell) + 1.0;
  if (!(a == a) || !(b == b)) return false;  // NaN
  double lo_c = a < 0.0 ? 0.0 : a;
  double hi_c = b > static_cast<double>(size) - 1.0 ? static_cast<double>(size) - 1.0 : b;
  if (lo_c > hi_c)
  {
    // nothing of the grid is reachable: any one-cell window will do
    lo_c = hi_c = (a < 0.0 ? 0.0 : static_cast<double>(size) - 1.0);
  }
  *lo = static_cast<int32_t>(lo_c);
  *n = static_cast<int32_t>(hi_c - lo_c) + 1;
  return true;
}

// Map geometry for a search; false if the byte-per-axis cell coordinate cannot
// hold the padded window.
bool lane_geometry(const MatchArgs & args, size_t lds_per_block, LaneGeom * geo, size_t * map_bytes)
{
  const double lin_cells = args.dlin_absmax * args.grid.inv_cell_size;
  if (!(lin_cells >= 0.0) || lin_cells > kMaxMapCells) return false;
  if (!(args.beam_rmax >= 0.0) || !std::isfinite(args.beam_rmax)) return false;
  const int32_t pad = static_cast<int32_t>(2.0 * lin_cells) + 3;
  // points_inner = R * beam + pose + (dx, dy): within beam_rmax + |d|max of the pose per axis
  const double reach = args.beam_rmax + args.dlin_absmax;
  if (!axis_window(args.pose_x, reach, args.grid.origin_x, args.grid.inv_cell_size,
                   args.grid.size_x, &geo->win_x0, &geo->win_w) ||
      !axis_window(args.pose_y, reach, args.grid.origin_y, args.grid.inv_cell_size,
                   args.grid.size_y, &geo->win_y0, &geo->win_h))
  {
    return false;
  }
  const uint64_t need_w = static_cast<uint64_t>(geo->win_w) + 2 * pad;
  const uint64_t need_h = static_cast<uint64_t>(geo->win_h) + 2 * pad;
  if (need_w > kMaxMapCells || need_h > kMaxMapCells) return false;
  geo->pad = pad;
  // finest sub-cell resolution whose coordinates fit one byte and whose map leaves
  // room in LDS for the cell records whenever the coarsest map would
  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  const bool records_fit = grid_bytes + kMapStride * need_h <= lds_per_block;
  // (a small search does not repay copying a 16x larger map into every block)
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const bool small_search = static_cast<uint64_t>(args.th_end - args.th_begin) * p1 * p1 < 4096;
  int sub_log2 = small_search ? 0 : 2;
  for (; sub_log2 > 0; --sub_log2)
  {
    const uint64_t w = need_w << sub_log2, h = need_h << sub_log2;
    if (w > kMaxMapCells || h > kMaxMapCells) continue;
    const size_t bytes = static_cast<size_t>(kMapStride) * h + (records_fit ? grid_bytes : 0);
    if (bytes <= lds_per_block) break;
  }
  geo->sub_log2 = sub_log2;
  geo->unit_scale = kFracScale * static_cast<double>(1 << sub_log2);
  geo->map_h = static_cast<int32_t>(need_h << sub_log2);
  // lanes add |d| <= lin_cells * unit_scale (+0.5 rounding); one cell of margin each side
  const double reach_units = (lin_cells + 1.0) * geo->unit_scale;
  geo->k_min = reach_units;
  geo->k_max_x = static_cast<double>(need_w - 1) * geo->unit_scale - reach_units;
  geo->k_max_y = static_cast<double>(need_h - 1) * geo->unit_scale - reach_units;
  *map_bytes = static_cast<size_t>(kMapStride) * geo->map_h;
  return true;
}

bool lane_records_in_lds(const MatchArgs & args, size_t map_bytes, size_t lds_per_block)
{
  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  return grid_bytes + map_bytes <= lds_per_block;
}

}  // namespace

size_t match_lane_outer_doubles(const MatchArgs & args)
{
  // rotated-beam table + the occupancy-map image (at most 256 x 256 bytes)
  return static_cast<size_t>(args.th_end - args.th_begin) * args.n_beams * 4 +
         static_cast<size_t>(kMapStride) * kMaxMapCells / sizeof(double);
}

bool match_lane_supported(const MatchArgs & args, size_t lds_per_block)
{
  LaneGeom geo;
  size_t map_bytes = 0;
  if (args.grid.occ_bits == nullptr || !lane_geometry(args, lds_per_block, &geo, &map_bytes)) return false;
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  const uint64_t items = static_cast<uint64_t>(args.th_end - args.th_begin) * p1 * p1;
  // (24-bit multiplies index the grid rows and the LDS records)
  return map_bytes <= lds_per_block && items <= kMaxLaneItems && args.grid.size_x < (1u << 24) &&
         args.grid.ncell < (1u << 24);
}

hipError_t launch_match_lane(const MatchArgs & args_in, double * outer, double * workspace,
                             uint32_t max_workers, int cus, size_t lds_per_block, bool no_skip,
                             hipStream_t stream, uint32_t * n_workers_out, bool * lds_records_out)
{
  MatchArgs args = args_in;
  args.partials = workspace;
  LaneGeom geo;
  size_t map_bytes = 0;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes)) return hipErrorInvalidValue;
  geo.no_skip = no_skip ? 1 : 0;
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
  hipLaunchKernelGGL(outer_table_kernel, dim3(oblocks), dim3(256), 0, stream, args,
                     reinterpret_cast<double4 *>(outer), map_image, geo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;

  // Few work items (the plugin's default lattice is 720 of them): 256-thread blocks
  // put them on four times as many CUs -- as long as every wave still gets at most
  // one item; the LDS image allows one block per CU, so beyond that the larger block
  // is what puts more waves on a CU.
  const bool small = n_items <= static_cast<uint64_t>(cus) * (kLaneThreadsSmall / kWave);
  const uint32_t waves_per_block = (small ? kLaneThreadsSmall : kLaneThreads) / kWave;
  uint32_t blocks = static_cast<uint32_t>((n_items + waves_per_block - 1) / waves_per_block);
  uint32_t max_blocks = static_cast<uint32_t>(cus);
  if (max_blocks * waves_per_block > max_workers) max_blocks = max_workers / waves_per_block;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;

  const bool lds_records = lane_records_in_lds(args, map_bytes, lds_per_block);
  if (lds_records_out != nullptr) *lds_records_out = lds_records;
  const size_t lds_bytes =
    map_bytes +
    (lds_records ? static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double) : 0);
  auto launch = [&](auto kernel, int threads) -> hipError_t {
    if (lds_bytes > 48 * 1024)
    {
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(lds_bytes));
      if (e2 != hipSuccess) return e2;
    }
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
  e = small ? pick(std::integral_constant<int, kLaneThreadsSmall>{})
            : pick(std::integral_constant<int, kLaneThreads>{});
  if (n_workers_out != nullptr)
  {
    *n_workers_out = dynamic_items ? static_cast<uint32_t>(n_items) : blocks * waves_per_block;
  }
  return e;
}

}  // namespace ndt2d
