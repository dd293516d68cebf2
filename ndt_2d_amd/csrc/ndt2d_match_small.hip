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
//           wave adds the likelihoods of its chunk of beams -- the look-up groups w % C,
//           w % C + C, ... of four beams each -- in beam order, with the same
//           two-instruction map look-up / bit-exact skipping / exact reference arithmetic
//           as the large kernel (ndt2d_lane_fn.h);
//   combine the C partial sums of a candidate are added in chunk order,
//           ((p_0 + p_1) + p_2) + ..., by the slot's first wave, which then keeps the
//           reference's strict-< best (:128-134) and the covariance accumulators
//           (:137-140) and leaves one 12-double record per (theta, patch) item.
//
// A candidate's sum is therefore a fixed-order sum of in-order chunk sums: deterministic,
// independent of timing, and within a few ulps of the reference's single running sum.
//
// Tiles.  The 64 candidates of a wave are an 8 x 8 patch of translations, or -- for
// lattices of at most 32 x 32 translations, the node's own -- 64 consecutive candidates
// of the theta step in the reference's visiting order (a 21 x 21 lattice is 7 such tiles
// but 9 patches with a quarter of their lanes idle; the tile still spans a fraction of a
// cell, which is what the wave-uniform skipping needs).
//
// Final reduction in the same launch.  The search ends every matchScan call, and a second
// launch for a few hundred records costs more than the records do (launch gap + 8 us on
// its own).  The launch carries ONE MORE BLOCK than the search needs -- the last one, so
// every other block has been dispatched before it -- which does nothing but reduce: every
// record is stored with agent-scope 8-byte atomics, then (its stores acknowledged) the
// launch's sequence number into the record's `done` word; thread r of the reducing block
// polls done[r], reads record r back the same way (the valid both-sides form of
// MI355X_MICROARCH.md: no L2 write-back, no stale L1/L2 line) and the block reduces them in
// a fixed order, applying "no candidate scored below 0 -> no index" and writing the
// result record to HBM and, behind a flag, to host-coherent memory.  (Round 2 had the block
// that drew the launch's last ticket do this: an atomic's round trip and two barriers on
// the critical path of every block, and the reduction started only after the last one.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ndt2d_lane_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kSmallMaxWaves = 16;
// beams per look-up group: 4 (the large search uses 8) keeps the kernel at 78 VGPRs = 6
// waves per SIMD; the search is latency bound, so the resident waves are what counts
constexpr int kSmallUnroll = 4;
constexpr uint32_t kSmallMaxBeams = 2048;     // 64 KB of LDS rows
constexpr uint32_t kRowDoubles = 3;           // {ox, oy, K}: 24 bytes (at 32, three 720-beam blocks do not fit a CU)

struct SmallPlan
{
  uint32_t patches_per_block;   // P: tile slots of a block (all of one theta step)
  uint32_t chunks;              // C: beam chunks = waves per tile slot
  uint32_t chunk_beams;         // beams per chunk (a multiple of kSmallUnroll)
  uint32_t blocks_per_theta;    // ceil(tiles / P)
  uint32_t need_w;              // map columns in use: window + 2 * pad
  uint32_t tiles;               // tiles of one theta step
  uint32_t linear;              // tiles are runs of 64 consecutive candidates, not 8 x 8 patches
  uint32_t no_tail;             // (experiments) leave out the final reduction
  uint32_t search_blocks;       // blocks of the search proper; block search_blocks (if launched) reduces
  // Blocks are dispatched in index order and a mid-size lattice needs more than one round of the
  // chip's block slots: with centre_first the block index walks the PATCHES from the lattice's
  // centre outwards (all theta steps of a patch together, the middle step first) instead of
  // theta by theta, row-major.  The search is centred on the caller's guess, so the patches
  // around the centre are the ones whose every beam hits the map -- a block of those runs ten
  // times as long as one at the rim, and must not be in the last round (1,352 blocks of 720 beams:
  // 85 -> 75 us; dealing one heavy block beside two light ones per CU instead was measured too:
  // 81 us -- a heavy block's 48 us are its own chain, not its neighbours' issue slots).  Which
  // block evaluates which tile leaves no trace in the records (indexed by (theta, tile)): same bits.
  uint32_t centre_first;
};

// j-th element of 0 .. n-1 visited from the middle outwards: mid, mid + 1, mid - 1, mid + 2, ...
__host__ __device__ inline uint32_t middle_out(uint32_t j, uint32_t n)
{
  const uint32_t mid = (n - 1u) / 2u;
  return (j & 1u) ? mid + (j + 1u) / 2u : mid - j / 2u;
}

// What the launch's last block needs for the final reduction.
struct SmallFinal
{
  // one word per record: the sequence number of the launch whose record is in place (zero when
  // the workspace is allocated; sequence numbers start at 1 and only rise)
  unsigned long long * done;
  double * record_out;          // device, 12 doubles
  double * record_out2;         // device, optional
  double * host_out;            // host-coherent, optional; flag at host_out[kHostFlagSlot]
  unsigned long long seq;
};

// The search tables as kernel arguments: [dth | cos | sin (n_th each) | dlin (n_lin)].
struct SmallTables
{
  double v[kArgTableDoubles];
};

__device__ __forceinline__ void store_agent(double * p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The reducing block of a launch (see the header): waits for the n_records records of the
// launch `fin.seq`, reduces them, publishes the result.  scratch: LDS for one record per wave.
__device__ __forceinline__ void small_final_reduction(const MatchArgs & a, const SmallFinal & fin,
                                                      uint32_t n_records, double * scratch)
{
  const uint32_t n_threads = blockDim.x;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t n_waves = n_threads >> 6;
  // Every lane takes whole records (r = thread, thread + n_threads, ...; the 12 loads of a
  // record in flight together), each wave reduces its lanes over the DPP network, the waves'
  // results meet in LDS and one thread per column adds them in wave order.  Fixed order throughout.
  double bs = 0.0, bi = kNoIndex;
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  bool gave_up = false;
  for (uint32_t r = threadIdx.x; r < n_records; r += n_threads)
  {
    // Forward progress does NOT rest on the order workgroups are dispatched in (which HIP leaves
    // undefined): the producers wait for nothing, and this block holds ONE of the chip's block
    // slots (hundreds of this kernel's size) while it polls -- whatever the order, the producers still
    // pending are dispatched into the others and finish.  What the index order this hardware
    // does dispatch in buys is efficiency only: the reducer (index search_blocks, the last) takes
    // its slot when every producer is resident or done, not before.  The bound below is for what
    // cannot be reasoned away (another process holding the chip for seconds): the block then
    // publishes "gave up" instead of a record and the host's wait returns NDT2D_ERR_HIP for this
    // call -- no trap, the context stays usable.
    BoundedPoll poll;
    while (__hip_atomic_load(fin.done + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fin.seq)
    {
      __builtin_amdgcn_s_sleep(2);
      if (poll.expired())
      {
        gave_up = true;
        break;
      }
    }
    if (gave_up) break;
    // The record's words are read only after its `done` word has been seen -- in program order, which
    // is all it takes: the producer stored them with agent-scope (sc1) stores, waited for their
    // acknowledgement, then stored `done`; the loads below are sc1 loads, served by L2.
    // (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 stores AND sc1 loads" need no fence.
    // Until round 6 an agent-scope acquire -- `buffer_inv sc1` -- stood here.)  The compiler keeps
    // the order:
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    const double * p = a.partials + static_cast<size_t>(r) * kRecord;
    double v[kRecord];
    // six 16-byte sc1 loads per record instead of twelve 8-byte ones: the reducing block pulls all
    // records through ONE CU's address path, a request per lane and load (round 6)
    static_assert(kRecord == 12, "a record is six 16-byte pieces");
    // (ONE statement: the compiler does not know that a hand-written load's register is not ready
    // until the counter says so)
    double2 q0, q1, q2, q3, q4, q5;
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                 "global_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %6, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %6, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %5, %6, off offset:80 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5)
                 : "v"(p)
                 : "memory");
    v[0] = q0.x; v[1] = q0.y; v[2] = q1.x; v[3] = q1.y; v[4] = q2.x; v[5] = q2.y;
    v[6] = q3.x; v[7] = q3.y; v[8] = q4.x; v[9] = q4.y; v[10] = q5.x; v[11] = q5.y;
    merge_best(v[0], v[1], bs, bi);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] += v[2 + k];
  }
  // Did any thread give up?  One word per wave behind the records' scratch, written with the
  // wave's record and read behind the same barrier -- no barrier of its own on the path every
  // search takes.  (Not __syncthreads_or(): the device library's workgroup reduction brings
  // static LDS, and this kernel's map must start at LDS offset 0.)
  uint32_t * wave_gave_up = reinterpret_cast<uint32_t *>(scratch + static_cast<size_t>(kSmallMaxWaves) * kRecord);
  const bool wave_flag = __builtin_amdgcn_ballot_w64(gave_up) != 0ull;
  // (a wave none of whose lanes holds a record -- six of the fifteen at the plugin's defaults -- has
  // the neutral record already and leaves its SIMD to the waves that reduce)
  if (wave * kWave < n_records)
  {
    wave_best_to_last_lane(bs, bi);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);
  }
  if (lane == kWave - 1)
  {
    scratch[wave * kRecord + 0] = bs;
    scratch[wave * kRecord + 1] = bi;
#pragma unroll
    for (int k = 0; k < 10; ++k) scratch[wave * kRecord + 2 + k] = acc[k];
    wave_gave_up[wave] = wave_flag ? 1u : 0u;
  }
  __syncthreads();
  bool any_gave_up = false;
  if (threadIdx.x < kWave)
  {
    for (uint32_t w = 0; w < n_waves; ++w) any_gave_up |= wave_gave_up[w] != 0u;
  }
  if (threadIdx.x < kRecord)
  {
    const uint32_t k = threadIdx.x;
    double val;
    if (k < 2)
    {
      double s0 = scratch[0], i0 = scratch[1];
      for (uint32_t w = 1; w < n_waves; ++w)
      {
        const double sw = scratch[w * kRecord], iw = scratch[w * kRecord + 1];
        merge_best(sw, iw, s0, i0);
      }
      val = k == 0 ? s0 : (s0 < 0.0 ? i0 : -1.0);   // no candidate scored below 0: no index
    }
    else
    {
      val = scratch[k];
      for (uint32_t w = 1; w < n_waves; ++w) val += scratch[w * kRecord + k];
    }
    // a record never came: no result -- NaN score, no index -- and the flag says why (the host's
    // wait returns NDT2D_ERR_HIP for this call; no trap, the context stays usable)
    if (any_gave_up) val = k == 1 ? -1.0 : __longlong_as_double(0x7ff8000000000000ll);
    fin.record_out[k] = val;
    if (fin.record_out2 != nullptr) fin.record_out2[k] = val;
    if (fin.host_out != nullptr) store_host(fin.host_out + k, val);
  }
  if (fin.host_out != nullptr && wave == 0)
  {
    // the record (lanes 0..11 of this wave) has been acknowledged before the flag leaves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) raise_host_flag(fin.host_out + kHostFlagSlot, any_gave_up ? (fin.seq | kHostFlagGaveUp) : fin.seq);
  }
#ifdef NDT2D_SMALL_TRACE
  // when the flag had left (final reduction and publish done)
  if (a.scores != nullptr && threadIdx.x == 0)
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double * tr = a.scores + static_cast<size_t>(8192) * kSmallMaxWaves * 8 - 8;
    tr[0] = static_cast<double>(wall_clock64());
    tr[1] = static_cast<double>(blockIdx.x);
  }
#endif
}

// COMPACT: the grid came with compacted records (GridDesc::compact_records, small maps
// installed from the host): the block keeps them and the cell -> record table in LDS, so
// an exact evaluation reads its record from LDS instead of gathering it through L2 -- in
// the node's own searches (tight lattice around a good guess) nearly every beam is one.
// ARG_TABLES: the search tables arrive as kernel arguments (`tab`), not through a.dth ..
// a.dlin: a matchScan whose beams the device already holds then needs no copy at all.
template <bool POW2, bool COMPACT, bool ARG_TABLES>
__global__ void __launch_bounds__(kSmallMaxWaves * kWave) __attribute__((amdgpu_waves_per_eu(6)))
match_small_kernel(const MatchArgs a,
                                                                             const LaneGeom geo,
                                                                             const SmallPlan plan,
                                                                             const SmallFinal fin,
                                                                             const SmallTables tab)
{
  auto table = [&](uint32_t which, uint32_t i) -> double {
    // which: 0 dth, 1 cos, 2 sin (index: theta step); 3 dlin (index: translation)
    if (ARG_TABLES) return tab.v[which * a.n_th + i];
    return (which == 0 ? a.dth : which == 1 ? a.cos_th : which == 2 ? a.sin_th : a.dlin)[i];
  };
  // LDS: [map, at offset 0 so that the packed cell bytes are the address]
  //      [COMPACT: cell ranks (uint16), compact records][rows][partials]
  extern __shared__ __align__(16) double lds[];
#ifdef NDT2D_SMALL_TRACE
  const unsigned long long t_start = __builtin_readcyclecounter();
  const unsigned long long w_start = wall_clock64();   // 100 MHz, one time base for the chip
#endif
  const GridDesc & g = a.grid;
  // (the map is addressed absolutely, from LDS offset 0: no static __shared__ here -- the
  // launcher checks, prepare_absolute_lds_kernel)
  if (blockIdx.x == plan.search_blocks)
  {
    // the launch's last block reduces (every search block has been dispatched before it)
    small_final_reduction(a, fin, plan.search_blocks / plan.blocks_per_theta * plan.tiles, lds);
    return;
  }
  uint8_t * lds_map = reinterpret_cast<uint8_t *>(lds);
  const uint32_t map_bytes = static_cast<uint32_t>(geo.map_h) * kMapStride;
  const uint32_t rank_bytes = COMPACT ? compact_rank_bytes(g.ncell) : 0u;
  const uint32_t compact_bytes = COMPACT ? (g.n_occ + 1) * kCellDoubles * static_cast<uint32_t>(sizeof(double)) : 0u;
  double * rows = lds + (map_bytes + rank_bytes + compact_bytes) / sizeof(double);
  double * partials = rows + max(static_cast<size_t>(a.n_beams) * kRowDoubles,
                                 static_cast<size_t>(kSmallMaxWaves) * kRecord);

  const uint32_t n_threads = blockDim.x;
  uint32_t t_local = blockIdx.x / plan.blocks_per_theta;
  uint32_t first_patch = (blockIdx.x - t_local * plan.blocks_per_theta) * plan.patches_per_block;
  if (plan.centre_first)
  {
    // (one tile per block, 8 x 8 patches: see SmallPlan::centre_first)
    const uint32_t c = blockIdx.x;
    const uint32_t n_th_local = plan.search_blocks / plan.blocks_per_theta;
    const uint32_t patch_rank = c / n_th_local;
    const uint32_t patches_1d = (a.n_lin + kPatch - 1) / kPatch;
    t_local = middle_out(c - patch_rank * n_th_local, n_th_local);
    first_patch = middle_out(patch_rank / patches_1d, patches_1d) * patches_1d + middle_out(patch_rank % patches_1d, patches_1d);
  }
  const uint32_t ith = a.th_begin + t_local * a.th_stride;

  // map window: one byte per grid cell, copied from the grid's extended byte image;
  // cells further than one cell outside the grid cannot be reached by anything: 0.
  // A real lidar's window is most of a 245 x 245 grid: 60 KB per block, and byte by byte
  // (a wave per row, a lane per column) the copy was 12 of such a block's 17 us
  // (experiments/small_trace.py real30).  Now: 16-column pieces -- the global side read
  // unaligned, the LDS row stride is 256 -- with all of a thread's loads in flight before
  // its first store; the (at most two) pieces of a row that straddle the image's edge are
  // left to a second, byte-wise pass over just those pieces.
  {
    // (a window wider than 256 cells: the image at one byte per block of cells; win_x0 - pad is
    // a multiple of the block size there, so the shift below is exact)
    const int32_t bl = geo.block_log2;
    const uint8_t * const image = bl > 0 ? g.block_bytes : g.cell_bytes;
    const int32_t ew = static_cast<int32_t>((g.size_x + (1u << bl) - 1u) >> bl) + 2;
    const int32_t eh = static_cast<int32_t>((g.size_y + (1u << bl) - 1u) >> bl) + 2;
    const int32_t x_shift = ((geo.win_x0 - geo.pad) >> bl) + 1;    // image column of map column 0
    const int32_t y_shift = ((geo.win_y0 - geo.pad) >> bl) + 1;
    const uint32_t pieces = (plan.need_w + 15u) / 16u;   // per row, <= 16
    const uint32_t n_vec = static_cast<uint32_t>(geo.map_h) * 16u;
    typedef uint32_t unaligned_u32 __attribute__((aligned(1)));
    constexpr int kCopyUnroll = 4;   // (eight in flight cost the gather forms of the kernel spilled registers)
    for (uint32_t base = threadIdx.x; base < n_vec; base += n_threads * kCopyUnroll)
    {
      uint4 v[kCopyUnroll];
#pragma unroll
      for (int u = 0; u < kCopyUnroll; ++u)
      {
        const uint32_t w = base + static_cast<uint32_t>(u) * n_threads;
        const int32_t ey = static_cast<int32_t>(w >> 4) + y_shift;
        const int32_t ex = static_cast<int32_t>((w & 15u) * 16u) + x_shift;
        v[u] = uint4{0u, 0u, 0u, 0u};
        if (w < n_vec && (w & 15u) < pieces && ey >= 0 && ey < eh && ex >= 0 && ex + 16 <= ew)
        {
          const unaligned_u32 * s32 =
            reinterpret_cast<const unaligned_u32 *>(image + static_cast<size_t>(ey) * ew + ex);
          v[u] = uint4{s32[0], s32[1], s32[2], s32[3]};
        }
      }
#pragma unroll
      for (int u = 0; u < kCopyUnroll; ++u)
      {
        const uint32_t w = base + static_cast<uint32_t>(u) * n_threads;
        if (w < n_vec && (w & 15u) < pieces)
        {
          *reinterpret_cast<uint4 *>(lds_map + (w >> 4) * kMapStride + (w & 15u) * 16u) = v[u];
        }
      }
    }
    // the pieces that hold column 0 / column ew - 1 of the image (those the pass above left
    // at zero although some of their columns are inside): piece index, or none
    const int32_t left = x_shift < 0 && (-x_shift) % 16 != 0 ? (-x_shift) / 16 : -1;
    const int32_t right = (ew - x_shift) % 16 != 0 ? (ew - x_shift) / 16 : -1;
    __syncthreads();
    const uint32_t n_edge = static_cast<uint32_t>(geo.map_h) * 32u;
    constexpr int kEdgeUnroll = 2;   // (loads of two trips in flight together; four spill in the gather forms)
    for (uint32_t base = threadIdx.x; base < n_edge; base += n_threads * kEdgeUnroll)
    {
      uint8_t bv[kEdgeUnroll];
      uint32_t at[kEdgeUnroll];
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u)
      {
        const uint32_t e = base + static_cast<uint32_t>(u) * n_threads;
        const uint32_t my = e >> 5;
        const int32_t piece = (e & 16u) ? right : left;
        const bool listed = e < n_edge && piece >= 0 && piece < static_cast<int32_t>(pieces) &&
                            !((e & 16u) && right == left);
        const uint32_t mx = static_cast<uint32_t>(listed ? piece : 0) * 16u + (e & 15u);
        const int32_t ey = static_cast<int32_t>(my) + y_shift;
        const int32_t ex = static_cast<int32_t>(mx) + x_shift;
        const bool inside = listed && ex >= 0 && ex < ew && ey >= 0 && ey < eh;
        at[u] = inside ? my * kMapStride + mx : 0xffffffffu;
        bv[u] = inside ? image[ey * ew + ex] : static_cast<uint8_t>(0);
      }
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u)
      {
        if (at[u] != 0xffffffffu) lds_map[at[u]] = bv[u];
      }
    }
  }
  if (COMPACT)
  {
    // ranks and records are contiguous in HBM ([records][ranks]) and 16-byte aligned
    const uint4 * src_rank = reinterpret_cast<const uint4 *>(g.cell_rank);
    uint4 * dst_rank = reinterpret_cast<uint4 *>(lds_map + map_bytes + kRankLead);
    for (uint32_t i = threadIdx.x; i < (rank_bytes - kRankLead) / 16; i += n_threads) dst_rank[i] = src_rank[i];
    if (threadIdx.x == 0)
    {
      reinterpret_cast<uint16_t *>(lds_map + map_bytes + kRankLead)[-1] = static_cast<uint16_t>(g.n_occ);
    }
    const double2 * src_rec = reinterpret_cast<const double2 *>(g.compact_records);
    double2 * dst_rec = reinterpret_cast<double2 *>(lds_map + map_bytes + rank_bytes);
    for (uint32_t i = threadIdx.x; i < compact_bytes / 16; i += n_threads) dst_rec[i] = src_rec[i];
  }
  // points_outer (:106-115) and the packed fixed-point map coordinate of each beam
  {
    const double ct = table(1, ith);
    const double st = table(2, ith);
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
      rows[kRowDoubles * b + 0] = ox;
      rows[kRowDoubles * b + 1] = oy;
      rows[kRowDoubles * b + 2] = kTwo52 + (rint(ky) * kTwo24 + rint(kx)) + kNearBias;
    }
  }
  __syncthreads();
#ifdef NDT2D_SMALL_TRACE
  const unsigned long long t_setup = __builtin_readcyclecounter();
#endif

  LaneCtx c;
  c.rank_address = map_bytes + (COMPACT ? kRankLead : 0u);
  c.lds_cells_address = map_bytes + rank_bytes;   // (not COMPACT: records are gathered from HBM)
  c.sub_log2 = 0;
  c.exact_index = geo.block_log2 > 0 ? 1u : 0u;   // a byte per block: the cell comes from the reference's own arithmetic
  c.idx_bias = static_cast<uint32_t>(geo.pad - geo.win_y0) * g.size_x +
               static_cast<uint32_t>(geo.pad - geo.win_x0);
  c.size_x = g.size_x;

  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = wave / plan.chunks;
  const uint32_t chunk = wave - slot * plan.chunks;
  const uint32_t n_lin = a.n_lin;
  const uint32_t tile = first_patch + slot;
  const bool active = slot < plan.patches_per_block && tile < plan.tiles;

  // lane -> candidate (ix, iy) of the theta step
  uint32_t ix, iy;
  if (plan.linear)
  {
    const uint32_t f = (active ? tile : 0u) * kWave + lane;
    ix = f / n_lin;
    iy = f - ix * n_lin;
  }
  else
  {
    const uint32_t patches_1d = (n_lin + kPatch - 1) / kPatch;
    const uint32_t pxi = active ? tile / patches_1d : 0;
    const uint32_t pyi = active ? tile - pxi * patches_1d : 0;
    ix = pxi * kPatch + (lane >> 3);
    iy = pyi * kPatch + (lane & 7);
  }
  const bool valid = active & (ix < n_lin) & (iy < n_lin);
  // lanes beyond the lattice edge shadow an edge candidate and are dropped below
  const double dx = table(3, min(ix, n_lin - 1));
  const double dy = table(3, min(iy, n_lin - 1));

  if (active)
  {
    const double inv_scaled = g.inv_cell_size * geo.unit_scale;
    const double dxy = rint(dy * inv_scaled) * kTwo24 + rint(dx * inv_scaled);
    // Chunk j walks the look-up groups j, j + C, j + 2C, ... (in that order): beams that
    // are expensive -- the sectors of the scan that face walls inside the map -- are then
    // spread over all the chunks instead of making one wave the block waits for.
    auto row_at = [&](uint32_t beam) -> double4 {
      const double * r = rows + kRowDoubles * beam;
      double4 o;
      o.x = r[0];
      o.y = r[1];
      o.z = r[2];
      o.w = 0.0;
      return o;
    };
    double sum = 0.0;
    SkipState skip = skip_state(0.0, geo.no_skip);
    for (uint32_t b = chunk * kSmallUnroll; b < a.n_beams; b += plan.chunks * kSmallUnroll)
    {
      if (b + kSmallUnroll <= a.n_beams)
      {
        double4 o[kSmallUnroll];
#pragma unroll
        for (int u = 0; u < kSmallUnroll; ++u) o[u] = row_at(b + u);
        lane_beams<kSmallUnroll, POW2, COMPACT, false, COMPACT>(g, c, o, dx, dy, dxy, sum, skip, geo.no_skip);
      }
      else
      {
        for (uint32_t t = b; t < a.n_beams; ++t)
        {
          const double4 one[1] = {row_at(t)};
          lane_beams<1, POW2, COMPACT, false, COMPACT>(g, c, one, dx, dy, dxy, sum, skip, geo.no_skip);
        }
      }
    }
    partials[wave * kWave + lane] = sum;
  }
#ifdef NDT2D_SMALL_TRACE
  const unsigned long long t_main = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef NDT2D_SMALL_TRACE
  const unsigned long long t_barrier = __builtin_readcyclecounter();
#endif

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
      const double dt = table(0, ith);
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
#ifndef NDT2D_SMALL_TRACE
      if (a.scores != nullptr) a.scores[static_cast<uint64_t>(t_local) * per_theta + in_theta] = score;
#endif
    }
    wave_best_to_last_lane(best_s, best_i);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = wave_sum_to_last_lane(acc[k]);
    if (lane == kWave - 1)
    {
      double * out = a.partials + (static_cast<size_t>(t_local) * plan.tiles + tile) * kRecord;
      store_agent(out + 0, best_s);
      store_agent(out + 1, best_i);
#pragma unroll
      for (int k = 0; k < 10; ++k) store_agent(out + 2 + k, acc[k]);
    }
    // the record is in L2 (its stores acknowledged) before its `done` word says so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == kWave - 1 && !NDT2D_TEST_DROPS_DONE(static_cast<size_t>(t_local) * plan.tiles + tile))
    {
      __hip_atomic_store(fin.done + (static_cast<size_t>(t_local) * plan.tiles + tile), fin.seq, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef NDT2D_SMALL_TRACE
  // experiments/small_trace.py: per wave {block start, setup done, beams done, barrier
  // passed, records written} in shader clocks, into the (otherwise unused) scores array
  if (a.scores != nullptr && lane == 0)
  {
    double * tr = a.scores + (static_cast<size_t>(blockIdx.x) * kSmallMaxWaves + wave) * 8;
    tr[0] = static_cast<double>(t_start);
    tr[1] = static_cast<double>(t_setup);
    tr[2] = static_cast<double>(t_main);
    tr[3] = static_cast<double>(t_barrier);
    tr[4] = static_cast<double>(__builtin_readcyclecounter());
    tr[5] = static_cast<double>(w_start);
    tr[6] = active ? 1.0 : 0.0;
    tr[7] = static_cast<double>(wall_clock64());
  }
#endif
}

size_t small_compact_bytes(const MatchArgs & args)
{
  return compact_rank_bytes(args.grid.ncell) +
         (static_cast<size_t>(args.grid.n_occ) + 1) * kCellDoubles * sizeof(double);
}

size_t small_lds_bytes(const MatchArgs & args, const LaneGeom & geo, uint32_t waves, bool compact)
{
  // (the rows' area is reused by the final reduction for one record per wave)
  size_t row_doubles = static_cast<size_t>(args.n_beams) * kRowDoubles;
  if (row_doubles < static_cast<size_t>(kSmallMaxWaves) * kRecord) row_doubles = kSmallMaxWaves * kRecord;
  return static_cast<size_t>(geo.map_h) * kMapStride + (compact ? small_compact_bytes(args) : 0) +
         row_doubles * sizeof(double) + static_cast<size_t>(waves) * kWave * sizeof(double);
}

// records in LDS when the grid has compacted ones and the block's image stays small
// enough for two blocks per CU
bool small_use_compact(const MatchArgs & args, const LaneGeom & geo)
{
  return args.grid.n_occ > 0 && args.grid.compact_records != nullptr &&
         small_lds_bytes(args, geo, kSmallMaxWaves, true) <= 72 * 1024;
}

// tiles (waves' worth of candidates) of one theta step
constexpr uint32_t kSmallLinearBelow = 32;
uint32_t small_tiles(uint32_t n_lin)
{
  if (n_lin <= kSmallLinearBelow) return (n_lin * n_lin + kWave - 1) / kWave;
  const uint32_t p1 = (n_lin + kPatch - 1) / kPatch;
  return p1 * p1;
}

// How a lattice is cut into blocks, every wave with whole look-up groups.  The cut
// depends on the beams and the lattice's translations only, never on the theta steps of
// this launch: a candidate's chunks, hence the bits of its score, are the same whichever
// rank of a sharded search evaluates it.
SmallPlan small_plan(const MatchArgs & args, const LaneGeom & geo, int cus)
{
  SmallPlan plan{};
  plan.linear = args.n_lin <= kSmallLinearBelow ? 1u : 0u;
  plan.tiles = small_tiles(args.n_lin);
  plan.need_w = static_cast<uint32_t>((geo.win_w + 2 * geo.pad + (1 << geo.block_log2) - 1) >> geo.block_log2);
  plan.no_tail = std::getenv("NDT2D_SMALL_NOTAIL") != nullptr ? 1u : 0u;
  const uint32_t groups = (args.n_beams + kSmallUnroll - 1) / kSmallUnroll;
  // (C beam chunks per tile, P tiles per block), P * C <= 16 waves.  Measured over the
  // plans of three shapes (experiments/small_plan_sweep.py; per-wave stamps:
  // experiments/small_trace.py): a wave needs ~0.36 us per beam whatever shares its SIMD up
  // to two waves, and a block ~3.4 us of setup.  Eight chunks -- two waves per SIMD -- are
  // the best cut of a 720-beam scan, each tile a block of its own (more, smaller blocks
  // balance better once a block runs for tens of microseconds); a 100-beam scan is cut
  // into chunks of five groups and three tiles share a block's setup.  The plan depends on
  // the beams and the lattice only, never on the launch's share of it.
  uint32_t best_c = (groups + 4) / 5;
  if (best_c > 8) best_c = 8;
  if (best_c < 1) best_c = 1;
  uint32_t best_p = 1;
  {
    const uint32_t cg = (groups + best_c - 1) / best_c;
    const uint32_t chunks = (groups + cg - 1) / cg;
    if (cg < 12) best_p = static_cast<uint32_t>(kSmallMaxWaves) / chunks;
  }
  if (const char * env = std::getenv("NDT2D_SMALL_CHUNKS"))   // tuning knobs for
  {                                                             // experiments/small_plan_sweep.py
    const uint32_t c = static_cast<uint32_t>(std::atoi(env));
    if (c >= 1 && c <= static_cast<uint32_t>(kSmallMaxWaves)) best_c = c < groups ? c : groups;
    best_p = kSmallMaxWaves / best_c;
  }
  if (const char * env = std::getenv("NDT2D_SMALL_PATCHES"))
  {
    const uint32_t p = static_cast<uint32_t>(std::atoi(env));
    if (p >= 1 && p * best_c <= static_cast<uint32_t>(kSmallMaxWaves)) best_p = p;
  }
  const uint32_t chunk_groups = (groups + best_c - 1) / best_c;
  plan.chunks = (groups + chunk_groups - 1) / chunk_groups;
  plan.chunk_beams = chunk_groups * kSmallUnroll;
  uint32_t p = best_p;
  if (p * plan.chunks > static_cast<uint32_t>(kSmallMaxWaves)) p = kSmallMaxWaves / plan.chunks;
  if (p > plan.tiles) p = plan.tiles;
  if (p < 1) p = 1;
  // Round 6: fewer tiles per block wherever the whole launch still fits the chip at once (six waves
  // per SIMD: 24 per CU).  A block ends with its heaviest tile and its waves meet at block-wide
  // barriers; with every tile a block of its own the light tiles are out of the way at once -- the
  // plugin's default search (560 tiles of 5 chunks): three tiles per block 24.8 us, one 22.4
  // (experiments/small_plan_sweep.py).  Which block evaluates a tile leaves no trace in its record:
  // same bits.  (The chunks -- which DO decide the bits of a score -- stay as they are.)
  if (best_p > 1 && std::getenv("NDT2D_SMALL_PATCHES") == nullptr && cus > 0)
  {
    const uint32_t n_th = args.th_end - args.th_begin;
    const bool compact = small_use_compact(args, geo);
    for (uint32_t q = 1; q < p; ++q)
    {
      // (blocks of a CU: by its wave slots and by its 160 KB of LDS -- a real lidar's 245 x 245 map is a
      // 61 KB window per block: two blocks per CU, the launch does not fit, three tiles go on sharing one
      // copy of it; tried without this test: matchScan on that map 28.6 -> 40 us)
      uint32_t per_cu = 24u / (q * plan.chunks);
      const size_t lds = small_lds_bytes(args, geo, q * plan.chunks, compact);
      const uint32_t by_lds = static_cast<uint32_t>((160u * 1024u) / (lds > 0 ? lds : 1));
      if (by_lds < per_cu) per_cu = by_lds;
      const uint64_t blocks = static_cast<uint64_t>(n_th) * ((plan.tiles + q - 1) / q);
      if (per_cu >= 1 && blocks <= static_cast<uint64_t>(cus) * per_cu)
      {
        p = q;
        break;
      }
    }
  }
  plan.patches_per_block = p;
  plan.blocks_per_theta = (plan.tiles + p - 1) / p;
  {
    static const bool knob_off = [] {   // A/B knob: NDT2D_SMALL_CENTRE_FIRST=0
      const char * v = std::getenv("NDT2D_SMALL_CENTRE_FIRST");
      return v != nullptr && v[0] == '0';
    }();
    plan.centre_first = (!knob_off && p == 1 && plan.linear == 0) ? 1u : 0u;
  }
  return plan;
}

}  // namespace

bool match_small_takes_arg_tables(const MatchArgs & args)
{
  return args.host_tables != nullptr &&
         3 * static_cast<uint64_t>(args.n_th) + args.n_lin <= kArgTableDoubles;
}

int match_small_block_log2(const MatchArgs & args, size_t lds_per_block)
{
  LaneGeom geo;
  size_t map_bytes = 0;
  if (args.grid.cell_bytes == nullptr || args.n_beams == 0 || args.n_beams > kSmallMaxBeams) return -1;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes, true) || geo.sub_log2 != 0) return -1;
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  if (static_cast<uint64_t>(args.n_th) * p1 * p1 > kSmallMaxItems) return -1;
  return geo.block_log2;
}

namespace
{

__global__ void __launch_bounds__(256) grid_block_bytes_kernel(const GridDesc g, uint32_t k, uint32_t bw,
                                                               uint32_t bh, uint8_t * out)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= bw * bh) return;
  const int32_t cells = 1 << k;
  const int32_t bx = static_cast<int32_t>(i % bw) - 1, by = static_cast<int32_t>(i / bw) - 1;
  out[i] = block_byte(g, bx * cells, by * cells, cells);
}

}  // namespace

size_t grid_block_bytes_size(const GridDesc & g, uint32_t block_log2)
{
  const size_t bw = ((static_cast<size_t>(g.size_x) + (1u << block_log2) - 1) >> block_log2) + 2;
  const size_t bh = ((static_cast<size_t>(g.size_y) + (1u << block_log2) - 1) >> block_log2) + 2;
  return bw * bh;
}

hipError_t grid_block_bytes_launch(const GridDesc & g, uint32_t block_log2, uint8_t * out, hipStream_t stream)
{
  const uint32_t bw = ((g.size_x + (1u << block_log2) - 1) >> block_log2) + 2;
  const uint32_t bh = ((g.size_y + (1u << block_log2) - 1) >> block_log2) + 2;
  hipLaunchKernelGGL(grid_block_bytes_kernel, dim3((bw * bh + 255) / 256), dim3(256), 0, stream, g, block_log2,
                     bw, bh, out);
  return hipGetLastError();
}

bool match_small_supported(const MatchArgs & args, size_t lds_per_block)
{
  LaneGeom geo;
  size_t map_bytes = 0;
  if (args.grid.cell_bytes == nullptr || args.n_beams == 0 || args.n_beams > kSmallMaxBeams) return false;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes, true)) return false;
  if (geo.sub_log2 != 0) return false;
  // a window wider than 256 cells needs the grid's block bytes at the geometry's block size
  if (geo.block_log2 > 0 && (args.grid.block_bytes == nullptr ||
                             args.grid.block_bytes_log2 != static_cast<uint32_t>(geo.block_log2)))
  {
    return false;
  }
  const uint64_t p1 = (args.n_lin + kPatch - 1) / kPatch;
  // the WHOLE lattice's work items (as choose_mapping counts them): every shard of a search
  // must make the same choice, or a candidate's bits would depend on the sharding; a
  // launch's share never has more items than the lattice, so the workspace bound holds too
  const uint64_t items = static_cast<uint64_t>(args.n_th) * p1 * p1;
  return args.th_end - args.th_begin <= args.n_th && items <= kSmallMaxItems && args.grid.size_x < (1u << 24) && args.grid.ncell < (1u << 24) &&
         small_lds_bytes(args, geo, kSmallMaxWaves, false) <= lds_per_block;
}

hipError_t launch_match_small(const MatchArgs & args_in, double * workspace, unsigned long long * done,
                              int cus, size_t lds_per_block, bool no_skip, double * record_out,
                              double * record_out2, double * host_record, unsigned long long seq,
                              hipStream_t stream)
{
  MatchArgs args = args_in;
  args.partials = workspace;
  LaneGeom geo;
  size_t map_bytes = 0;
  if (!lane_geometry(args, lds_per_block, &geo, &map_bytes, true) || geo.sub_log2 != 0 ||
      (geo.block_log2 > 0 && (args.grid.block_bytes == nullptr ||
                              args.grid.block_bytes_log2 != static_cast<uint32_t>(geo.block_log2))))
  {
    return hipErrorInvalidValue;
  }
  geo.no_skip = no_skip ? 1 : 0;
  SmallPlan plan = small_plan(args, geo, cus);
  const uint32_t waves = plan.patches_per_block * plan.chunks;
  plan.search_blocks = (args.th_end - args.th_begin) * plan.blocks_per_theta;
  // (+ the reducing block, see the file header)
  const uint32_t blocks = plan.search_blocks + (plan.no_tail ? 0u : 1u);
  SmallFinal fin;
  fin.done = done;
  fin.record_out = record_out;
  fin.record_out2 = record_out2;
  fin.host_out = host_record;
  fin.seq = seq;
  const bool arg_tables = match_small_takes_arg_tables(args);
  SmallTables tab;
  if (arg_tables)
  {
    std::memcpy(tab.v, args.host_tables, (3 * static_cast<size_t>(args.n_th) + args.n_lin) * sizeof(double));
  }
  const bool compact = small_use_compact(args, geo);
  const size_t lds_bytes = small_lds_bytes(args, geo, waves, compact);
  auto launch = [&](auto kernel) -> hipError_t {
    // (no static LDS in front of the map: prepare_absolute_lds_kernel, ndt2d_kernels.h)
    const hipError_t e2 = prepare_absolute_lds_kernel(reinterpret_cast<const void *>(kernel), lds_bytes);
    if (e2 != hipSuccess) return e2;
    if (std::getenv("NDT2D_SMALL_DEBUG") != nullptr)
    {
      int per_cu = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel),
                                                         static_cast<int>(waves * kWave), lds_bytes);
      std::fprintf(stderr, "match_small: %u blocks x %u waves, %zu B LDS, C=%u P=%u, %d blocks per CU (API)\n",
                   blocks, waves, lds_bytes, plan.chunks, plan.patches_per_block, per_cu);
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves * kWave), lds_bytes, stream, args, geo, plan, fin, tab);
    return hipGetLastError();
  };
  auto pick = [&](auto pow2_tag, auto compact_tag) -> hipError_t {
    constexpr bool P2 = decltype(pow2_tag)::value;
    constexpr bool CP = decltype(compact_tag)::value;
    return arg_tables ? launch(match_small_kernel<P2, CP, true>) : launch(match_small_kernel<P2, CP, false>);
  };
  using T = std::true_type;
  using F = std::false_type;
  return compact ? (args.grid.pow2 ? pick(T{}, T{}) : pick(F{}, T{}))
                 : (args.grid.pow2 ? pick(T{}, F{}) : pick(F{}, F{}));
}

}  // namespace ndt2d

#ifdef NDT2D_TEST_HOOKS
// Test builds only (libndt2d_hip_hooks.so; not declared in include/ndt2d_hip.h): make the producer
// of record / pose `which - 1` of this translation unit's kernels withhold its `done` word
// (0: normal operation) -- tests/test_gpu_bounded_poll.py.
extern "C" int ndt2d_test_drop_done_small(int which)
{
  return hipMemcpyToSymbol(HIP_SYMBOL(ndt2d::g_test_drop_done), &which, sizeof(int)) == hipSuccess ? 0 : 3;
}
#endif
