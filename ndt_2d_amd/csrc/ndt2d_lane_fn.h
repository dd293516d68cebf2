// Building blocks of the lane-per-candidate searches (ndt2d_match_lane.hip: large
// lattices; ndt2d_match_small.hip: small lattices): the fixed-point occupancy-map
// look-up, its exponent bound, the per-lane skip state and the group-of-beams body.
// See the header of ndt2d_match_lane.hip for the scheme.  Included by .hip units only.
#ifndef NDT2D_LANE_FN_H_
#define NDT2D_LANE_FN_H_

#include <cmath>

#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kPatch = 8;            // patch is kPatch x kPatch candidates = one wave
#ifndef NDT2D_LANE_UNROLL
#define NDT2D_LANE_UNROLL 8
#endif
constexpr int kUnroll = NDT2D_LANE_UNROLL;  // beams per look-up group (a divisor of 64)
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
// +kNearUnits on both packed fractions.  It is part of every beam's K: a lane within
// kNearUnits below a boundary then looks its map byte up in the sub-cell across it, whose
// box (widened by kBoxMargin = 64 units) holds the lane's point just as well, and such a
// lane is "near" -- it takes the reference's own index arithmetic, not the look-up's cell.
constexpr double kNearBias = kNearUnits * (kTwo24 + 1.0);
static_assert(kNearUnits == 4, "near_boundary() tests the biased fraction against 2 * kNearUnits");

struct LaneGeom
{
  int32_t pad;     // border cells on every side of the window in the map
  // The map is kept at 2^sub_log2 sub-cells per cell (as fine as the one-byte
  // coordinate and LDS allow): the finer the sub-cell, the tighter its bound on the
  // exponent, so fewer beams next to walls take the exact path.
  int32_t sub_log2;
  // ... or, for a window wider than 256 cells, at one map cell per 2^block_log2 x
  // 2^block_log2 block of grid cells (then sub_log2 = 0): the byte speaks for the whole
  // block, and a lane in a block that can score takes the reference's own index arithmetic.
  int32_t block_log2;
  double unit_scale;  // fixed-point units per cell = 2^16 << sub_log2 (>> block_log2)
  double inv_scaled;  // grid.inv_cell_size * unit_scale: fixed-point units per metre (a kernel argument,
                      // so that the lane kernels find it in SGPRs instead of holding the product in VGPRs)
  int32_t map_h;   // map rows of kMapStride bytes: (win_h + 2 * pad) << sub_log2 (>> block_log2)
  // Window of grid cells the map covers: every point this search can produce
  // (scan pose +- (longest beam + largest offset)) lies inside it or outside the
  // grid.  For small grids it is the whole grid.
  int32_t win_x0, win_y0, win_w, win_h;
  double k_min, k_max_x, k_max_y;  // clamp of the per-beam fixed-point coordinate
  int32_t no_skip;  // control mode: every beam takes the exact path
  // Patch pre-test (large search): the 64 candidates of a patch lie within box_span + 1
  // map sub-cells per axis of the patch's first lane; -1: no pre-test (wider patches,
  // lattices whose first lane is not the patch's corner, the control mode).
  int32_t box_span;
};

constexpr int kMaxBoxSpan = 3;   // the pre-test reads a 4 x 4 sub-cell box
constexpr int kMaxBlockLog2 = 3; // coarsest map: one byte per 8 x 8 grid cells (windows up to 2,048 cells)

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

// Map byte from a sub-cell's own occupancy bit and the bound on the exponents
// reachable from it (-inf: none): bit 0 = occupied, bit 1 = something reachable,
// bits 2..7 = level: 0 nothing reachable; l >= 1: every exponent reachable is
// <= l - 63; 63: no claim (also NaN).
__device__ __forceinline__ uint8_t map_byte(uint32_t self, double bound)
{
  uint32_t level = 0;
  if (bound > -HUGE_VAL || bound != bound)
  {
    level = !(bound <= 0.0) ? 63u : (bound < -62.0 ? 1u : static_cast<uint32_t>(63 + static_cast<int32_t>(ceil(bound))));
  }
  return static_cast<uint8_t>(self | (level != 0 ? 2u : 0u) | (level << 2));
}

// Byte of sub-cell (lx, ly) of grid cell (cx, cy) -- which may lie outside the grid --
// at 2^sub_log2 sub-cells per cell: its box in the world, widened by kBoxMargin of a
// sub-cell so that it holds every point whose rounded fixed-point coordinate lands in
// this sub-cell, and over the grid cells that widened box overlaps (its own, plus the
// neighbours across a cell boundary it touches) the maximum of Cell::score's exponent.
__device__ __forceinline__ uint8_t sub_cell_byte(const GridDesc & g, int32_t cx, int32_t cy,
                                                 int32_t lx, int32_t ly, int32_t sub_log2)
{
  const int32_t sx = static_cast<int32_t>(g.size_x), sy = static_cast<int32_t>(g.size_y);
  const int32_t sub = 1 << sub_log2;
  const double sub_size = g.cell_size / static_cast<double>(sub);
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
  return map_byte(self, bound);
}

// Byte of the block of `cells` x `cells` grid cells that starts at (cx0, cy0) -- any part of
// it may lie outside the grid -- for maps coarser than the grid (LaneGeom::block_log2): the
// box of the whole block, widened like a sub-cell's, and the maximum of Cell::score's
// exponent over that box for every cell the widened box overlaps (the block and the ring
// of cells around it).  bit 0: some cell of the block holds a distribution.
__device__ __forceinline__ uint8_t block_byte(const GridDesc & g, int32_t cx0, int32_t cy0, int32_t cells)
{
  const int32_t sx = static_cast<int32_t>(g.size_x), sy = static_cast<int32_t>(g.size_y);
  const double size = g.cell_size * static_cast<double>(cells);
  const double x0 = g.origin_x + static_cast<double>(cx0) * g.cell_size - kBoxMargin * size;
  const double y0 = g.origin_y + static_cast<double>(cy0) * g.cell_size - kBoxMargin * size;
  const double x1 = x0 + (1.0 + 2.0 * kBoxMargin) * size;
  const double y1 = y0 + (1.0 + 2.0 * kBoxMargin) * size;
  uint32_t self = 0;
  double bound = -HUGE_VAL;
  for (int32_t ny = cy0 - 1; ny <= cy0 + cells; ++ny)
  {
    for (int32_t nx = cx0 - 1; nx <= cx0 + cells; ++nx)
    {
      if (nx >= 0 && nx < sx && ny >= 0 && ny < sy)
      {
        const uint32_t cell = static_cast<uint32_t>(ny * sx + nx);
        if (((g.occ_bits[cell >> 5] >> (cell & 31u)) & 1u) != 0)
        {
          if (nx >= cx0 && nx < cx0 + cells && ny >= cy0 && ny < cy0 + cells) self = 1;
          const double e = exponent_upper_bound(g.cells_lds_image + static_cast<size_t>(cell) * kCellDoubles,
                                                x0, x1, y0, y1);
          bound = !(e <= bound) ? e : bound;   // NaN-propagating max
        }
      }
    }
  }
  return map_byte(self, bound);
}

// One occupancy-map byte per grid cell, for the grid extended by one cell on every
// side ([size_y + 2][size_x + 2], cell (cx, cy) at (cy + 1) * (size_x + 2) + cx + 1):
// the byte the lane-per-candidate search's map holds at one map cell per grid cell
// (ndt2d_lane_fn.h, sub_cell_byte with sub_log2 = 0).  It depends on the grid only, so
// it is prepared once here and the small-lattice search copies its window from it.
// FROM_CELLS6: occupancy and records come from the cells6 records themselves (the fused
// kernel of a host-installed grid, which has no packed records yet); else from the packed
// records and the bitmap (device build).  lane16 = this thread's lane in its 16-lane row,
// i = the extended-grid cell the row works on.
template <bool FROM_CELLS6>
__device__ __forceinline__ void cell_byte_row(const GridDesc & g, const double * cells6, uint32_t i,
                                              uint32_t j, uint8_t * bytes)
{
  // 16 lanes (one DPP row) per cell, lane j < 9 takes neighbour j of the 3 x 3 block: the
  // bound is a maximum over the neighbours that can score, each a closed form with a few
  // divisions -- nine of them one after the other in one lane was the whole kernel's time
  const uint32_t w = g.size_x + 2, h = g.size_y + 2;
  const bool live = i < w * h;
  const int32_t cx = static_cast<int32_t>(live ? i % w : 0) - 1, cy = static_cast<int32_t>(live ? i / w : 0) - 1;
  // the sub-cell box of sub_cell_byte() at one sub-cell per cell
  const double x0 = g.origin_x + (static_cast<double>(cx) - kBoxMargin) * g.cell_size;
  const double y0 = g.origin_y + (static_cast<double>(cy) - kBoxMargin) * g.cell_size;
  const double x1 = x0 + (1.0 + 2.0 * kBoxMargin) * g.cell_size;
  const double y1 = y0 + (1.0 + 2.0 * kBoxMargin) * g.cell_size;
  double bound = -HUGE_VAL;
  uint32_t self = 0;
  if (live && j < 9)
  {
    const int32_t nx = cx + static_cast<int32_t>(j % 3) - 1, ny = cy + static_cast<int32_t>(j / 3) - 1;
    if (nx >= 0 && nx < static_cast<int32_t>(g.size_x) && ny >= 0 && ny < static_cast<int32_t>(g.size_y))
    {
      const uint32_t cell = static_cast<uint32_t>(ny) * g.size_x + static_cast<uint32_t>(nx);
      if (FROM_CELLS6)
      {
        const double * c = cells6 + static_cast<size_t>(cell) * 6;
        if (!(c[5] < 5.0))
        {
          const double rec[5] = {c[0], c[1], -0.5 * c[2], -0.5 * c[3], -0.5 * c[4]};
          if (j == 4) self = 1;
          bound = exponent_upper_bound(rec, x0, x1, y0, y1);
        }
      }
      else if ((g.occ_bits[cell >> 5] >> (cell & 31u)) & 1u)
      {
        if (j == 4) self = 1;
        bound = exponent_upper_bound(g.cells_lds_image + static_cast<size_t>(cell) * kCellDoubles, x0, x1, y0, y1);
      }
    }
  }
  // NaN-propagating maximum over the row (a NaN bound means "no claim")
  uint32_t nan_any = bound != bound ? 1u : 0u;
  double m = bound != bound ? -HUGE_VAL : bound;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1)
  {
    m = fmax(m, __shfl_xor(m, off, 16));
    nan_any |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(nan_any), off, 16));
    self |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(self), off, 16));
  }
  if (live && j == 0) bytes[i] = map_byte(self, nan_any ? NAN : m);
}

// Map bytes of a grid installed as a list (ndt2d_build.hip): a byte depends on the 3 x 3 cells
// around it, so the bytes that are not zero lie in the 3 x 3 blocks around the listed cells that
// can score.  Row = (listed cell, neighbour position), 16 lanes per row; neighbouring listed
// cells compute the same byte twice and store the same value.  thread: the launch-wide index
// of this lane among the job's lanes.
__device__ __forceinline__ void sparse_byte_rows(const GridDesc & g, const SparseBytesJob & job, uint32_t thread)
{
  const uint32_t row = thread >> 4;
  const uint32_t k = row / 9, pos = row - k * 9;
  uint32_t target = 0xffffffffu;   // (cell_byte_row treats an index past the array as "no cell")
  if (k < job.n)
  {
    const uint32_t cell = job.cell_index[k];
    if (cell < g.ncell && !(job.cells6[static_cast<size_t>(k) * 6 + 5] < 5.0))
    {
      const uint32_t cx = cell % g.size_x, cy = cell / g.size_x;
      // extended-grid coordinates of the neighbour: (cx + 1 + dx, cy + 1 + dy), dx, dy in -1..1
      target = (cy + pos / 3) * (g.size_x + 2) + cx + pos % 3;
    }
  }
  cell_byte_row<false>(g, nullptr, target, thread & 15u, job.bytes);
}

// LDS bytes of the cell -> compact record table: kRankLead bytes whose last two hold the
// sentinel's rank -- entry -1, what a lane outside every occupied cell selects with one
// v_cndmask_b32 and the inline constant -1 -- then a u16 per grid cell and entry ncell
// (again the sentinel), rounded to 16.
constexpr uint32_t kRankLead = 16;
__host__ __device__ inline uint32_t compact_rank_bytes(uint32_t ncell)
{
  return kRankLead + (((ncell + 1u) * 2u + 15u) & ~15u);
}

#if defined(NDT2D_LANE_HIST) || defined(NDT2D_LANE_PATHS)
__device__ double * g_lane_hist = nullptr;   // set by the kernel from MatchArgs::scores
#endif
// experiments/lane_paths.py (-DNDT2D_LANE_PATHS): how often a wave takes each path of the search --
// the dynamic instruction mix is these counts times the paths' static instruction lists.
//   [200] work items            [201] 64-beam chunks pre-tested   [202] look-up groups of U beams (x U)
//   [203] groups with a live lane (per-beam tests run)            [204] beams with a live lane
//   [205] exact evaluations      [206] ... through the reference's index arithmetic (a lane near a boundary)
//   [207] evaluations that needed exp()   [208] skip-state refreshes   [209] items reduced (some sum != 0)
//   [210] single beams through the U = 1 form
#ifdef NDT2D_LANE_PATHS
#define NDT2D_PATH(slot, amount)                                                              \
  do {                                                                                        \
    if (g_lane_hist != nullptr && (threadIdx.x & 63u) == 0) atomicAdd(g_lane_hist + (slot), static_cast<double>(amount)); \
  } while (0)
#else
#define NDT2D_PATH(slot, amount) do {} while (0)
#endif

struct LaneCtx
{
  uint32_t rank_address;       // LDS byte address of entry 0 of the cell -> compact record table (COMPACT)
  uint32_t lds_cells_address;  // LDS byte address of the packed records (behind the map)
  uint32_t sub_log2;           // map sub-cell -> map cell: shift right
  uint32_t exact_index;        // the map is coarser than the grid: never take a cell from it
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

__device__ __forceinline__ uint32_t lds_u16_at(uint32_t address)
{
  typedef const __attribute__((address_space(3))) uint16_t * lds_u16_ptr;
  return *reinterpret_cast<lds_u16_ptr>(address);
}

// Lanes whose 16-bit fraction (the low half of v), biased by kNearUnits, is below
// 2 * kNearUnits: the lane mask of one 16-bit compare.
__device__ __forceinline__ uint64_t near_boundary(uint32_t v)
{
  uint64_t lanes;
  asm("v_cmp_gt_u16_e64 %0, 8, %1" : "=s"(lanes) : "v"(v));
  return lanes;
}

__device__ __forceinline__ uint2 lds_two_dwords_at(uint32_t address)
{
  typedef const __attribute__((address_space(3), aligned(4))) uint32_t * lds_u32_ptr;
  const lds_u32_ptr p = reinterpret_cast<lds_u32_ptr>(address);
  return make_uint2(p[0], p[1]);
}

// Patch pre-test, one BEAM per lane: can any of the patch's 64 candidates reach a
// distribution with this beam?  s is the packed fixed-point coordinate of the beam for
// the patch's first lane (its corner: the smallest dx and dy); the other lanes' coordinates
// lie within span sub-cells of it on either axis, so the map bytes of that box decide: all
// of level 0 (byte < 4: nothing reachable from the sub-cell) means every lane would skip
// the beam whatever its threshold.  Rows beyond `span` are not read; the columns beyond
// it are masked.  (Reads past the last map row, possible only for rows the candidates
// cannot reach, land in the block's own LDS behind the map or return 0.)
__device__ __forceinline__ bool patch_can_score(double s, int32_t span)
{
  const uint32_t lo = static_cast<uint32_t>(__double2loint(s));
  const uint32_t hi = static_cast<uint32_t>(__double2hiint(s));
  const uint32_t address = __builtin_amdgcn_perm(hi, lo, 0x0c0c0502u);   // (cell y << 8) | cell x
  const uint32_t base = address & ~3u;
  const uint32_t shift = address & 3u;
  uint32_t any = 0;
#pragma unroll
  for (int r = 0; r <= kMaxBoxSpan; ++r)
  {
    if (r <= span)
    {
      const uint2 w = lds_two_dwords_at(base + static_cast<uint32_t>(r) * kMapStride);
      any |= __builtin_amdgcn_alignbyte(w.y, w.x, shift);   // bytes x .. x + 3 of the row
    }
  }
  // levels live in bits 2..7 of a byte; keep columns 0 .. span
  const uint32_t columns = span >= 3 ? 0xfcfcfcfcu : (0xfcfcfcfcu >> (8 * (3 - span)));
  return (any & columns) != 0u;
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

// SCALAR_ROWS: the rows o[] are wave-uniform values held in scalar registers (the
// large search reads them with scalar loads); false: they sit in vector registers
// (the small search broadcasts them from LDS).
// COMPACT (with LDS_RECORDS): the LDS records are the compacted ones, addressed through
// the cell -> record table at c.rank_address (uint16 per grid cell, also in LDS).
template <int U, bool POW2, bool LDS_RECORDS, bool SCALAR_ROWS = true, bool COMPACT = false>
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
  NDT2D_PATH(U == 1 ? 210 : 202, U);
  if (wave_any(top >= skip_level))
  {
    NDT2D_PATH(203, 1);
    bool added = false;
    // have the beams' end points on their way before the first exact evaluation needs
    // them (left to itself the compiler loads each pair inside its own branch)
    if (SCALAR_ROWS)
    {
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : : "s"(o[u].x), "s"(o[u].y));
    }
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
        NDT2D_PATH(204, 1);
        // within kNearUnits of a unit boundary on either axis: the beams' coordinates
        // come with both 16-bit fractions biased by kNearUnits (see kNearBias), so that is
        // (frac + 4) mod 2^16 < 8, a 16-bit compare per axis; the y fraction straddles
        // the two words (bytes 3, 4)
        const uint32_t fy = __builtin_amdgcn_perm(hi[u], lo[u], 0x0c0c0403u);
        const bool occ = (m[u] & 1u) != 0;
        const uint64_t near_mask =
          (near_boundary(lo[u]) | near_boundary(fy) | ((no_skip | c.exact_index) != 0 ? ~0ull : 0ull)) & live_mask;
        const uint64_t occ_lanes = __builtin_amdgcn_ballot_w64(occ);
        const uint64_t occ_mask = occ_lanes & live_mask;
#ifdef NDT2D_LANE_NO_EXACT
        // (experiment: what the search costs WITHOUT its exact evaluations -- results are wrong)
        if ((occ_mask | near_mask) == 0x1234567ull)
#else
        if ((occ_mask | near_mask) != 0ull)
#endif
        {
          NDT2D_PATH(205, 1);
          if (near_mask != 0ull) NDT2D_PATH(206, 1);
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
            const uint32_t cell = __umul24(row, c.size_x) + (col - c.idx_bias);
            if (COMPACT)
            {
              // entry -1 of the rank table is the sentinel's: one select on the lane mask
              // the occupancy test left in scalar registers
              asm("v_cndmask_b32 %0, -1, %1, %2" : "=v"(idx) : "v"(cell), "s"(occ_lanes));
            }
            else
            {
              idx = occ ? cell : g.ncell;
            }
          }
          if (COMPACT) idx = lds_u16_at(c.rank_address + 2u * idx);
          const double e = LDS_RECORDS ? lds_record_exponent(c.lds_cells_address, idx, px, py)
                                       : indexed_exponent<false>(g, nullptr, idx, px, py);
#ifdef NDT2D_LANE_HIST
          // experiments/lane_useful_hist.py: how many lanes of an exact evaluation matter.
          // hist[k]: evaluations with k lanes flagged (live and occupied or near a boundary);
          // hist[70 + k]: with k lanes whose term can change their sum; [140] / [141]:
          // evaluations that did / did not need exp()
          if (g_lane_hist != nullptr)
          {
            const uint64_t flagged = occ_mask | near_mask;
            const uint64_t needed = __builtin_amdgcn_ballot_w64(!(e < skip_below)) & flagged;
            if ((threadIdx.x & 63u) == 0)
            {
              atomicAdd(g_lane_hist + __popcll(flagged), 1.0);
              atomicAdd(g_lane_hist + 70 + __popcll(needed), 1.0);
              atomicAdd(g_lane_hist + (needed != 0ull ? 140 : 141), 1.0);
            }
          }
#endif
          // !(e < bound) also keeps NaN exponents (degenerate cells) on the exact path
          if (wave_any(!(e < skip_below)))
          {
            NDT2D_PATH(207, 1);
            sum += exp_score(e);
            added = true;
          }
        }
      }
    }
    if (added) NDT2D_PATH(208, 1);
    if (added) skip = skip_state(sum, no_skip);
  }
}

// Window of grid cells [lo, hi] reachable along one axis, clipped to the grid.
inline bool axis_window(double pose, double reach, double origin, double inv_cell, uint32_t size,
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
// hold the padded window even at one map cell per 4 x 4 grid cells.
// coarse_map: the small search, which copies its map from a per-grid image -- one map cell per
// grid cell whatever the lattice (the bytes prepared when the grid was installed), or, for a
// window wider than 256 cells, per block of 2^block_log2 cells from the image of
// grid_block_bytes (blocks aligned to the grid's origin: the window's origin is moved down to
// a block boundary).
inline bool lane_geometry(const MatchArgs & args, size_t lds_per_block, LaneGeom * geo,
                          size_t * map_bytes, bool coarse_map = false)
{
  const double lin_cells = args.dlin_absmax * args.grid.inv_cell_size;
  if (!(lin_cells >= 0.0) || lin_cells > kMaxMapCells) return false;
  if (!(args.beam_rmax >= 0.0) || !std::isfinite(args.beam_rmax)) return false;
  int32_t pad = static_cast<int32_t>(2.0 * lin_cells) + 3;
  // points_inner = R * beam + pose + (dx, dy): within beam_rmax + |d|max of the pose per axis
  const double reach = args.beam_rmax + args.dlin_absmax;
  if (!axis_window(args.pose_x, reach, args.grid.origin_x, args.grid.inv_cell_size,
                   args.grid.size_x, &geo->win_x0, &geo->win_w) ||
      !axis_window(args.pose_y, reach, args.grid.origin_y, args.grid.inv_cell_size,
                   args.grid.size_y, &geo->win_y0, &geo->win_h))
  {
    return false;
  }
  uint64_t need_w = static_cast<uint64_t>(geo->win_w) + 2 * pad;
  uint64_t need_h = static_cast<uint64_t>(geo->win_h) + 2 * pad;
  // A window beyond the one-byte coordinate: one map cell per block of 2 x 2 or 4 x 4 grid
  // cells (the border and the window rounded up to whole blocks).
  int block_log2 = 0;
  while (need_w > (static_cast<uint64_t>(kMaxMapCells) << block_log2) ||
         need_h > (static_cast<uint64_t>(kMaxMapCells) << block_log2))
  {
    if (++block_log2 > kMaxBlockLog2) return false;
    const int32_t cells = 1 << block_log2;
    pad = (static_cast<int32_t>(2.0 * lin_cells) + 3 + cells - 1) / cells * cells;
    // (coarse_map: the window as it grows when its origin moves to a block boundary)
    const int32_t grow_x = coarse_map ? (geo->win_x0 & (cells - 1)) : 0;
    const int32_t grow_y = coarse_map ? (geo->win_y0 & (cells - 1)) : 0;
    need_w = (static_cast<uint64_t>(geo->win_w + grow_x) + 2 * pad + cells - 1) / cells * cells;
    need_h = (static_cast<uint64_t>(geo->win_h + grow_y) + 2 * pad + cells - 1) / cells * cells;
  }
  if (coarse_map && block_log2 > 0)
  {
    const int32_t cells = 1 << block_log2;
    geo->win_w += geo->win_x0 & (cells - 1);
    geo->win_h += geo->win_y0 & (cells - 1);
    geo->win_x0 &= ~(cells - 1);
    geo->win_y0 &= ~(cells - 1);
  }
  geo->pad = pad;
  geo->block_log2 = block_log2;
  // finest sub-cell resolution whose coordinates fit one byte and whose map leaves
  // room in LDS for the cell records whenever the coarsest map would
  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  const bool records_fit = grid_bytes + kMapStride * need_h <= lds_per_block;
  // (a small search does not repay copying a 16x larger map into every block)
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  // (the WHOLE lattice's items, not the launch's share: the map's resolution decides which
  // kernel forms fit LDS, and with them whether a mid-size lattice's beams are cut into parts)
  const bool small_search = static_cast<uint64_t>(args.n_th) * p1 * p1 < 4096;
  int sub_log2 = (small_search || coarse_map || block_log2 > 0) ? 0 : 2;
  for (; sub_log2 > 0; --sub_log2)
  {
    const uint64_t w = need_w << sub_log2, h = need_h << sub_log2;
    if (w > kMaxMapCells || h > kMaxMapCells) continue;
    const size_t bytes = static_cast<size_t>(kMapStride) * h + (records_fit ? grid_bytes : 0);
    if (bytes <= lds_per_block) break;
  }
  geo->sub_log2 = sub_log2;
  geo->unit_scale = kFracScale * static_cast<double>(1 << sub_log2) / static_cast<double>(1 << block_log2);
  geo->inv_scaled = args.grid.inv_cell_size * geo->unit_scale;
  geo->map_h = static_cast<int32_t>((need_h << sub_log2) >> block_log2);
  // lanes add |d| <= lin_cells * unit_scale (+0.5 rounding); one cell of margin each side
  const double reach_units = (lin_cells + 1.0) * geo->unit_scale;
  geo->k_min = reach_units;
  geo->k_max_x = static_cast<double>(need_w - 1) * geo->unit_scale - reach_units;
  geo->k_max_y = static_cast<double>(need_h - 1) * geo->unit_scale - reach_units;
  *map_bytes = static_cast<size_t>(kMapStride) * geo->map_h;
  // lanes differ from the first by rint(d_i * s) - rint(d_0 * s) <= span * s + 1 fixed-point
  // units per axis; a box that starts in the first lane's sub-cell ends floor of that later
  geo->box_span = -1;
  if (args.patch_span >= 0.0)
  {
    const double span_units = std::ceil(args.patch_span * args.grid.inv_cell_size * geo->unit_scale) + 2.0;
    const double sub_cells = std::floor((65535.0 + span_units) / 65536.0);
    if (sub_cells <= static_cast<double>(kMaxBoxSpan)) geo->box_span = static_cast<int32_t>(sub_cells);
  }
  return true;
}

}  // namespace

}  // namespace ndt2d

#endif  // NDT2D_LANE_FN_H_
