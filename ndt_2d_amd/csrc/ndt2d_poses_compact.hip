// Batched ScanMatcherNDT::scorePoints (reference src/scan_matcher_ndt.cpp:156-178),
// the loop body of ParticleFilter::measure (src/particle_filter.cpp:81-87), with
// per-wave compaction of the work that is not an exact zero.
//
// One pose (particle) per lane.  For random poses only a few percent of the
// (pose, beam) pairs land in a cell that holds a distribution; every other pair
// contributes exactly 0.0 (src/ndt_model.cpp:107,165-169).  Evaluating the
// Gaussian for all 64 lanes whenever one lane needs it wastes the VALU, and
// gathering a 48-byte record per lane from 64 different cache lines saturates
// the texture-addresser.  So each wave runs two phases:
//
//   A (dense, per beam)   transform the beam by the lane's pose (:171-173), exact
//                         NDT::getIndex (src/ndt_model.cpp:203-218), look the cell up
//                         in an occupancy bitmap held in LDS; occupied lanes append
//                         {px, py, cell | lane} to the wave's LDS queue (ballot +
//                         mbcnt prefix).
//   B (every 64 items)    lane j takes queue item j: gathers the cell record (one
//                         64-byte line), evaluates Cell::score (:105-116) and adds it
//                         to its owner's running sum with an LDS f64 atomic.
//
// The queue is FIFO in beam order and a (pose, beam) pair appears at most once, so
// each owner's terms arrive in the reference's beam order; only exact zeros are
// left out of the sums.
//
// SCREEN (default): phase A does not need the exact point, only a conservative answer
// to "can this pair hit a cell with a distribution?".  It evaluates the cell
// coordinate in FP32 -- u = U0 + (c/cell) bx - (s/cell) by with per-lane FP32
// constants, u and v as the two halves of v_pk_fma_f32: two instructions instead
// of eight FP64 operations -- takes the cell by floor, and queues the pair
// {beam | lane} if that cell is occupied OR the coordinate lies within `guard` (a
// bound of the FP32 error, see screen_guard()) of a cell boundary or of the grid's
// edge.  Phase B then redoes the transform and
// NDT::getIndex exactly in FP64 for the few queued pairs.  Every pair that can
// contribute is queued, in the same order, and a queued pair that turns out to be
// empty adds exactly +0.0, so the scores are those of the unscreened kernel bit for
// bit (variant "compact-exact" is that kernel; the tests compare the two).
#include <cstdlib>
#include <cstring>

#include "ndt2d_device_fn.h"
#include "ndt2d_lane_fn.h"
#include "ndt2d_poses_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kQueueCap = 128;          // items per wave (ring); < 64 pending + <= 64 pushed
// The screened kernel appends a whole block of 32 beams' candidates at once (at most 64 pending +
// what fits; a block with more takes the round-by-round path): a larger ring.
constexpr int kQueueCapScreen = 256;
constexpr uint32_t kCellBits = 26;      // queue meta word = cell index | lane << 26
constexpr uint32_t kCellMask = (1u << kCellBits) - 1;
// (kChunks, ndt2d_poses_fn.h: a lane's score is the in-order sum of its chunk sums whatever
// number of waves -- 1, 2, 4 or 8 -- shares the 64 poses of a group, so the result does not
// depend on the launch geometry or on how a particle set is sharded.  It differs from the
// reference's single running sum by a few ulps.)
constexpr int kPhaseA = 2;               // beams per dense step
constexpr size_t kLdsBudget = 160 * 1024;

template <int THREADS, bool SCREEN>
struct CompactLayout
{
  static constexpr int kWaves = THREADS / kWave;
  static constexpr int kCap = SCREEN ? kQueueCapScreen : kQueueCap;
  // occupancy bitmap (at LDS offset 0), then doubles: [stats kWaves*8][sums THREADS]
  //          [q_px kWaves*cap][q_py kWaves*cap] (unscreened kernel only: the screened one
  //          queues {beam | lane} words)[q_meta (u32) kWaves*cap/2]; then beams (f64), beams
  //          (f32, SCREEN only) and, when several waves share a group, the chunk sums
  //          [groups][kChunks][64].
  static constexpr size_t kPointDoubles = SCREEN ? 0 : static_cast<size_t>(kWaves) * kCap;
  static constexpr size_t kFixedDoubles =
    static_cast<size_t>(kWaves) * 8 + THREADS + 2 * kPointDoubles + static_cast<size_t>(kWaves) * kCap / 2;
};

// SPLIT waves share the 64 poses of a group; each takes kChunks / SPLIT chunks.
template <int THREADS, bool POW2, bool SCREEN, bool COARSE = false>
// (4 waves per SIMD = at most 128 VGPRs, also for the 256-thread geometry: the
// accumulators that no longer fit are spilled on the once-per-pose path, and a CU
// holds four blocks instead of three)
#ifndef NDT2D_POSES_WAVES_PER_EU
#define NDT2D_POSES_WAVES_PER_EU 4
#endif
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(NDT2D_POSES_WAVES_PER_EU)))
score_poses_compact_kernel(const PosesArgs a, const uint32_t split)
{
  using L = CompactLayout<THREADS, SCREEN>;
  extern __shared__ __align__(16) double lds[];
  const GridDesc & g = a.grid;
  // The occupancy bitmap comes first: lds_word_at() addresses it absolutely, so it
  // must start at LDS offset 0 -- no static __shared__ in this kernel (the launcher checks:
  // prepare_absolute_lds_kernel, ndt2d_kernels.h).
  uint32_t * lds_bits = reinterpret_cast<uint32_t *>(lds);
  // COARSE: one bit per block of 2^k x 2^k cells (PosesArgs::coarse_bits) instead of one per cell
  const uint32_t k_log2 = COARSE ? a.coarse_log2 : 0u;
  const uint32_t bits_sx = COARSE ? (g.size_x + (1u << k_log2) - 1u) >> k_log2 : g.size_x;
  const uint32_t bits_sy = COARSE ? (g.size_y + (1u << k_log2) - 1u) >> k_log2 : g.size_y;
  const uint32_t bits_outside = COARSE ? bits_sx * bits_sy : g.ncell;   // a bit that is never set
  const uint32_t n_words = (bits_outside + 1 + 31) / 32;
  const uint32_t * bits_src = COARSE ? a.coarse_bits : g.occ_bits;
  double * sh_stats = lds + ((n_words + 3) & ~3u) / 2;
  double * sh_sum = sh_stats + L::kWaves * 8;
  double * q_px_all = sh_sum + THREADS;
  double * q_py_all = q_px_all + L::kPointDoubles;
  uint32_t * q_meta_all = reinterpret_cast<uint32_t *>(q_py_all + L::kPointDoubles);
  double * lds_beams = reinterpret_cast<double *>(q_meta_all + L::kWaves * L::kCap);
  float * lds_beams_f = reinterpret_cast<float *>(lds_beams + 2 * ((a.n_beams + 1) & ~1u));
  // chunk sums [group][chunk][lane], only used (and allocated) when split > 1
  double * sh_chunk =
    reinterpret_cast<double *>(lds_beams_f + (SCREEN ? 2 * ((a.n_beams + 1) & ~1u) + kScreenPadFloats : 0));

  for (uint32_t i = threadIdx.x; i < 2 * a.n_beams; i += THREADS)
  {
    const double v = a.beams_xy[i];
    lds_beams[i] = v;
    if (SCREEN) lds_beams_f[i] = static_cast<float>(v);
  }
  for (uint32_t i = threadIdx.x; i < n_words; i += THREADS) lds_bits[i] = bits_src[i];
  __syncthreads();

  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t group = wave / split;          // group of 64 poses inside the block
  const uint32_t part = wave - group * split;   // which share of the chunks
  const uint32_t groups_per_block = L::kWaves / split;
  constexpr uint32_t kCap = static_cast<uint32_t>(L::kCap);
  double * q_px = q_px_all + (SCREEN ? 0 : wave * kCap);
  double * q_py = q_py_all + (SCREEN ? 0 : wave * kCap);
  uint32_t * q_meta = q_meta_all + wave * kCap;
  double * my_sums = sh_sum + wave * kWave;
  double * my_chunks = sh_chunk + static_cast<size_t>(group) * kChunks * kWave;
  const uint64_t lanes_below = (1ull << lane) - 1ull;
  const uint32_t chunk_len = (a.n_beams + kChunks - 1) / kChunks;
  const uint32_t chunks_per_part = kChunks / split;

  // The eight moment sums of updateStatistics are kept per WAVE in LDS (sh_stats), added to
  // once per group of 64 poses after a wave-level reduction: carried in registers across the
  // pose loop they cost 16 VGPRs of the hot loop's budget (and were spilled: 44 bytes of
  // scratch per lane, 15 MB of HBM writes per cfg-3 launch).
  if (lane < 8) sh_stats[wave * 8 + lane] = 0.0;

  // SCREEN: what a pose needs for the exact transform in phase B, read from its owner
  // lane with wave shuffles
  double pose_x = 0.0, pose_y = 0.0, pose_c = 0.0, pose_s = 0.0;

  // Phase B: lanes [0, n) each evaluate one queued item
  auto drain = [&](uint32_t head, uint32_t n) {
    if (SCREEN)
    {
      const uint32_t slot = (head + lane) & (kCap - 1);
      const uint32_t meta = lane < n ? q_meta[slot] : 0u;
      const int owner = static_cast<int>(meta >> kBeamBits);
      // (all lanes take part in the shuffles)
      const double ox = __shfl(pose_x, owner, kWave);
      const double oy = __shfl(pose_y, owner, kWave);
      const double oc = __shfl(pose_c, owner, kWave);
      const double os = __shfl(pose_s, owner, kWave);
      if (lane < n)
      {
        const double2 p = reinterpret_cast<const double2 *>(lds_beams)[meta & ((1u << kBeamBits) - 1u)];
        // p = t * (x, y, 1) (:172-173), exactly as the unscreened phase A has it
        const double px = ox + (oc * p.x - os * p.y);
        const double py = oy + (os * p.x + oc * p.y);
        const double e = indexed_exponent<false>(g, nullptr, cell_index<POW2>(g, px, py), px, py);
        atomicAdd(&my_sums[owner], exp_score(e));
      }
    }
    else if (lane < n)
    {
      const uint32_t slot = (head + lane) & (kCap - 1);
      const double px = q_px[slot];
      const double py = q_py[slot];
      const uint32_t meta = q_meta[slot];
      const double e = indexed_exponent<false>(g, nullptr, meta & kCellMask, px, py);
      atomicAdd(&my_sums[meta >> kCellBits], exp_score(e));
    }
  };

  const uint64_t poses_per_block = static_cast<uint64_t>(groups_per_block) * kWave;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * poses_per_block;
  const uint64_t n_round = (a.n_poses + stride - 1) / stride * stride;
  for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * poses_per_block; base < n_round;
       base += stride)
  {
    const uint64_t i = base + static_cast<uint64_t>(group) * kWave + lane;
    const bool valid = i < a.n_poses;
    double x = kPadCoord, y = kPadCoord, th = 0.0;  // padding lanes: every beam is outside
    if (valid)
    {
      x = a.poses_xyt[3 * i];
      y = a.poses_xyt[3 * i + 1];
      th = a.poses_xyt[3 * i + 2];
    }
    // toEigen(pose): AngleAxisd(theta, Z) -> [[c,-s],[s,c]] (conversions.hpp:64-68)
    double s, c;
    sincos(th, &s, &c);

    // SCREEN: FP32 cell coordinate of a beam end, u = u0 + ci * bx - si * by,
    // v = v0 + si * bx + ci * by.  A pose further from the grid than any beam reaches
    // (or not finite) cannot hit it: its coordinate is parked far outside.
    f32x2 uv0 = {0.0f, 0.0f}, rot_x = {0.0f, 0.0f}, rot_y = {0.0f, 0.0f};
    if (SCREEN)
    {
      pose_x = x;
      pose_y = y;
      pose_c = c;
      pose_s = s;
      const double inv = 1.0 / g.cell_size;
      const double du = (x - g.origin_x) * inv, dv = (y - g.origin_y) * inv;
      const double reach = a.beam_rmax * inv + 2.0;
      const bool in_band = du > -reach && du < static_cast<double>(g.size_x) + reach &&
                           dv > -reach && dv < static_cast<double>(g.size_y) + reach;
      // (parked at the centre of cell (-2, -2): outside, and not near any boundary)
      const double guard = static_cast<double>(a.screen_guard);
      const float ci = in_band ? static_cast<float>(c * inv) : 0.0f;
      const float si = in_band ? static_cast<float>(s * inv) : 0.0f;
      uv0 = f32x2{in_band ? static_cast<float>(du + guard) : -1.5f,
                  in_band ? static_cast<float>(dv + guard) : -1.5f};
      rot_x = f32x2{ci, si};
      rot_y = f32x2{-si, ci};
    }

    // (unscreened control path) the bitmap bit of an exact cell index; ncell = outside
    auto bit_of_cell = [&](uint32_t cell) -> uint32_t {
      if (!COARSE) return cell;
      if (cell >= g.ncell) return bits_outside;
      const uint32_t cy = cell / g.size_x, cx = cell - cy * g.size_x;
      return (cy >> k_log2) * bits_sx + (cx >> k_log2);
    };
    (void)bit_of_cell;
    double total = 0.0;
    for (uint32_t cj = 0; cj < chunks_per_part; ++cj)
    {
      const uint32_t chunk = part * chunks_per_part + cj;
      const uint32_t k0 = min(chunk * chunk_len, a.n_beams);
      const uint32_t k1 = min(k0 + chunk_len, a.n_beams);
      my_sums[lane] = 0.0;
      uint32_t head = 0, count = 0;
      // one beam's queue step: occupied lanes append, a full queue is drained
      auto push = [&](double px, double py, uint32_t idx, bool occ) {
        const uint64_t mask = __ballot(occ);
        if (mask != 0)
        {
          if (occ)
          {
            const uint32_t slot =
              (head + count + static_cast<uint32_t>(__popcll(mask & lanes_below))) & (kCap - 1);
            q_px[slot] = px;
            q_py[slot] = py;
            q_meta[slot] = idx | (lane << kCellBits);
          }
          count += static_cast<uint32_t>(__popcll(mask));
          if (count >= kWave)
          {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain(head, kWave);
            head = (head + kWave) & (kCap - 1);
            count -= kWave;
          }
        }
      };
      // SCREEN: can this pair contribute?  Yes if the FP32 cell is occupied, or if the
      // coordinate is so close to a cell boundary (or to the grid's edge) that the
      // exact point may lie in the neighbouring cell; no if the cell is empty or the
      // point is more than a cell outside the grid.
      // Written branch-free (bitwise & / |, an unconditional bitmap read of the
      // never-set bit `ncell` for points outside) and in two halves, so that the LDS
      // reads of the kPhaseA beams of a step are in flight together.
      struct Screen
      {
        uint32_t idx;   // bitmap bit to test
        float edge;     // min over the axes of fract(coordinate + guard): near a boundary if < 2 guard
      };
      auto screen_address = [&](float2 bf) -> Screen {
        // (u, v) + guard: one packed FMA per beam coordinate
        const f32x2 uv = __builtin_elementwise_fma(
          rot_x, f32x2{bf.x, bf.x}, __builtin_elementwise_fma(rot_y, f32x2{bf.y, bf.y}, uv0));
        int iu, iv;  // floor, one instruction each
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iu) : "v"(uv.x));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iv) : "v"(uv.y));
        Screen r;
        r.idx = screen_bit_index(iu, iv, k_log2, bits_sx, bits_outside);
        // The coordinate carries +guard, so "within guard of a cell boundary" reads
        // fract < 2 guard, and where that is false floor() is the exact point's cell.
        // (Far outside the grid it also fires now and then: a harmless false candidate.)
        r.edge = fminf(__builtin_amdgcn_fractf(uv.x), __builtin_amdgcn_fractf(uv.y));
        return r;
      };
      // mask = 2 * mask + (edge < limit): compare into vcc, add-with-carry shifts it in
      auto shift_in_near = [](uint32_t & mask, float edge, float limit) {
        asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
            : "+v"(mask)
            : "v"(edge), "v"(limit)
            : "vcc");
      };
      // (mask = the wave's ballot of `candidate`, taken by the caller from the compare itself)
      auto push_screened = [&](uint32_t beam, bool candidate, uint64_t mask) {
        {
          if (candidate)
          {
            const uint32_t slot =
              (head + count + static_cast<uint32_t>(__popcll(mask & lanes_below))) & (kCap - 1);
            q_meta[slot] = beam | (lane << kBeamBits);
          }
          count += static_cast<uint32_t>(__popcll(mask));
          if (count >= kWave)
          {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            drain(head, kWave);
            head = (head + kWave) & (kCap - 1);
            count -= kWave;
          }
        }
      };
      // kPhaseA beams per step so that their LDS reads (beam, bitmap word) overlap
      uint32_t k = k0;
      if (SCREEN)
      {
        // Nearly every beam has a candidate in SOME lane, so a queue step per beam would
        // run every time.  Instead each lane collects the candidates of kScreenBlock beams
        // as bits of one register, and the block is queued in rounds: round r appends
        // every lane's r-th candidate.  A lane's pairs still enter the queue in beam
        // order, which is all its running sum depends on; the rounds are as many as
        // the busiest lane has candidates (a few), not kScreenBlock.
        constexpr int kScreenStep = 4;
        constexpr int kScreenBlock = 32;
        const float near_limit = 2.0f * a.screen_guard;
        for (; k < k1; k += kScreenBlock)
        {
          // occupancy bits are funnel-shifted in from the top (v_lshrrev + v_alignbit); the
          // "near" bits are shifted in first-beam-first and bit-reversed at the end of the block.  A block
          // that runs past the chunk's end screens whatever follows in LDS (the f32 beam
          // array is padded by a block) and drops those bits.
          uint32_t cmask = 0, near_rev = 0;
#pragma unroll
          for (int b0 = 0; b0 < kScreenBlock; b0 += kScreenStep)
          {
            Screen sc[kScreenStep];
            uint32_t word[kScreenStep];
#pragma unroll
            for (int u = 0; u < kScreenStep; ++u)
            {
              sc[u] = screen_address(reinterpret_cast<const float2 *>(lds_beams_f)[k + b0 + u]);
              word[u] = lds_word_at((sc[u].idx >> 3) & ~3u);
            }
            // All kScreenStep bitmap reads are issued before the first is waited for.  Left to
            // itself the compiler consumes beam u's word right behind beam u + 1's address
            // arithmetic -- one LDS round trip per beam on the wave's path instead of one per
            // step (round 5: cfg-3 - 4 %, cfg-5 - 3 %; also why 2 / 8 / 16 beams per step had
            // measured the same as 4).  Reading the next step's beams ahead as well: no gain.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kScreenStep; ++u)
            {
              // bit (idx & 31) of the word enters at the top; after 32 beams beam j is bit j
              cmask = __builtin_amdgcn_alignbit(word[u] >> (sc[u].idx & 31u), cmask, 1u);
              shift_in_near(near_rev, sc[u].edge, near_limit);
            }
          }
          cmask |= __builtin_bitreverse32(near_rev);
          if (k1 - k < static_cast<uint32_t>(kScreenBlock)) cmask &= (1u << (k1 - k)) - 1u;
          // The block's candidates are appended lane by lane (lane 0's in beam order, then lane
          // 1's ...: a lane's pairs still enter the queue in beam order): every lane knows its
          // first slot from a prefix sum of the lanes' counts and writes its own entries -- no
          // ballot and no rank per entry.  (Round 5; until then the block was appended in rounds,
          // round r = every lane's r-th candidate: the same number of loop trips, twice the
          // instructions each and a scalar count per trip.  The rounds remain for a block that
          // does not fit the ring whole.)
          const uint32_t mine = static_cast<uint32_t>(__builtin_popcount(cmask));
          const uint32_t upto = wave_inclusive_scan_u32(mine);
          const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(upto), kWave - 1));
          if (total == 0u) continue;
          if (count + total <= kCap)
          {
            uint32_t slot = head + count + (upto - mine);
            const uint32_t tag = (lane << kBeamBits) + k;   // + bit index = beam | lane << 26
            while (cmask != 0u)
            {
              const uint32_t lowest = static_cast<uint32_t>(__ffs(static_cast<int>(cmask))) - 1u;
              q_meta[slot & (kCap - 1)] = tag + lowest;
              ++slot;
              cmask &= cmask - 1u;
            }
            count += total;
            while (count >= kWave)
            {
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
              drain(head, kWave);
              head = (head + kWave) & (kCap - 1);
              count -= kWave;
            }
            continue;
          }
          for (uint64_t pending = __builtin_amdgcn_ballot_w64(cmask != 0u); pending != 0ull;
               pending = __builtin_amdgcn_ballot_w64(cmask != 0u))
          {
            const uint32_t lowest = static_cast<uint32_t>(__ffs(static_cast<int>(cmask))) - 1u;
            push_screened(k + lowest, cmask != 0u, pending);
            cmask &= cmask - 1u;
          }
        }
        k = k1;
      }
      for (; k + kPhaseA <= k1; k += kPhaseA)
      {
        double px[kPhaseA], py[kPhaseA];
        uint32_t idx[kPhaseA], word[kPhaseA], bit[kPhaseA];
#pragma unroll
        for (int u = 0; u < kPhaseA; ++u)
        {
          const double2 p = reinterpret_cast<const double2 *>(lds_beams)[k + u];
          // p = t * (x, y, 1) (:172-173): translation + (c*x + (-s)*y), (s*x + c*y)
          px[u] = x + (c * p.x - s * p.y);
          py[u] = y + (s * p.x + c * p.y);
          idx[u] = cell_index<POW2>(g, px[u], py[u]);
          bit[u] = bit_of_cell(idx[u]);
          word[u] = lds_bits[bit[u] >> 5];
        }
#pragma unroll
        for (int u = 0; u < kPhaseA; ++u)
        {
          push(px[u], py[u], idx[u], ((word[u] >> (bit[u] & 31u)) & 1u) != 0);
        }
      }
      for (; k < k1; ++k)
      {
        const double2 p = reinterpret_cast<const double2 *>(lds_beams)[k];
        const double px = x + (c * p.x - s * p.y);
        const double py = y + (s * p.x + c * p.y);
        const uint32_t idx = cell_index<POW2>(g, px, py);
        const uint32_t bit = bit_of_cell(idx);
        push(px, py, idx, ((lds_bits[bit >> 5] >> (bit & 31u)) & 1u) != 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      drain(head, count);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const double csum = my_sums[lane];
      if (split > 1)
      {
        my_chunks[chunk * kWave + lane] = csum;
      }
      else
      {
        total = chunk == 0 ? csum : total + csum;  // ((c_0 + c_1) + c_2) + ...
      }
    }
    if (split > 1) __syncthreads();

    if (part == 0)
    {
      double sum = total;
      if (split > 1)
      {
        sum = my_chunks[lane];
#pragma unroll
        for (int j = 1; j < kChunks; ++j) sum += my_chunks[j * kWave + lane];
      }
      // score = sum of (-likelihood) / n  ==  -(sum) / n (:175-177)
      const double score = -sum / static_cast<double>(a.n_beams);
      if (valid) a.scores[i] = score;
      if (a.partials != nullptr)
      {
        // sums for ParticleFilter::updateStatistics (particle_filter.cpp:166-200)
        // (padding lanes hold a huge coordinate: they contribute exact zeros)
        const double w = valid ? score : 0.0;
        const double xv = valid ? x : 0.0, yv = valid ? y : 0.0;
        double st[8];
        st[0] = w;
        st[1] = w * xv;
        st[2] = w * yv;
        st[3] = w * c;
        st[4] = w * s;
        st[5] = w * xv * xv;
        st[6] = w * xv * yv;
        st[7] = w * yv * yv;
#pragma unroll
        for (int k = 0; k < 8; ++k) st[k] = wave_sum(st[k]);
        if (lane == 0)
        {
#pragma unroll
          for (int k = 0; k < 8; ++k) sh_stats[wave * 8 + k] += st[k];
        }
      }
    }
    if (split > 1) __syncthreads();
  }

  if (a.partials != nullptr)
  {
    __syncthreads();
    if (threadIdx.x < 8)
    {
      double v = 0.0;
      for (int w = 0; w < L::kWaves; ++w) v += sh_stats[w * 8 + threadIdx.x];
      a.partials[static_cast<size_t>(blockIdx.x) * 8 + threadIdx.x] = v;
    }
  }
}

// A handful of poses -- ScanMatcherNDT::scorePoints / scoreScan called for ONE pose
// (reference src/scan_matcher_ndt.cpp:151-178; the unchanged ParticleFilter::measure
// calls it once per particle, src/particle_filter.cpp:81-87): one block per pose, a
// thread per beam, so the call costs a launch and a few microseconds instead of a
// lane walking all the beams.  The terms are then added in the order the batched
// kernel above adds them -- chunk sums c_j in beam order, ((c_0 + c_1) + c_2) + ... --
// so scorePoints(points, pose) == scorePoses(points, [pose])[0] bit for bit.
// poses: device array, or (poses == nullptr) the values in `few` (kernel arguments:
// no upload for up to kFewPoses poses).  flag (optional, host-coherent memory):
// receives `seq` once every score is written, for a host that spins instead of
// synchronising the stream.
// ARG_BEAMS: the (few) beams arrive as kernel arguments too -- a scoreScan on a new scan
// then queues no copy at all; block 0 also leaves them in beams_out, the context's beam
// buffer, for the calls that follow on the same scan (matchScan).
//
// f.stats: the launch is a whole ParticleFilter::measure for a small particle set (the
// node's default is at most 500, src/ndt_mapper.cpp:81-82): every block leaves its score
// in f.dev_scores (agent-scope store), and the block that finishes last runs
// updateStatistics (src/particle_filter.cpp:163-218) over all of them -- total weight,
// normalised weights, weighted mean, circular mean, x/y covariance, the theta variance of
// the second pass -- writing weights and result straight to host-coherent memory.
template <bool POW2, bool ARG_BEAMS>
__global__ void __launch_bounds__(kFewThreads) score_few_kernel(const PosesArgs a, const FewPoses few,
                                                                const FewOut f,
                                                                const FewBeams arg_beams)
{
  extern __shared__ __align__(16) double lds[];
  double * terms = lds;                         // [n_beams]
  double * chunk_sums = lds + a.n_beams;        // [kChunks + 8]: chunk sums, then a scratch row
  const GridDesc & g = a.grid;
  const uint32_t i = blockIdx.x;
  const uint32_t n = static_cast<uint32_t>(a.n_poses);
  // Blocks behind the poses' (and the waiting block's): the map bytes a list install left to
  // do (FewOut::side).  Nothing in this launch reads them or waits for them.
  const uint32_t first_side = n + ((f.flag != nullptr && n > 1) ? 1u : 0u);
  if (i >= first_side)
  {
    sparse_byte_rows(g, f.side, (i - first_side) * kFewThreads + threadIdx.x);
    return;
  }
  // With a completion flag and more than one pose the launch has one more block than poses:
  // the last one, dispatched after all the others, waits for their `done` words and raises
  // the flag (or runs updateStatistics first).  A ticket drawn by every block put an atomic's
  // round trip on the path of the block that happened to finish last.
  if (i == n)   // (n > 1 and a flag: see first_side)
  {
    bool gave_up = false;
    for (uint32_t k = threadIdx.x; k < n && !gave_up; k += kFewThreads)
    {
      // (forward progress as for ndt2d_match_small.hip's reducing block: the poses' blocks wait for
      // nothing and this block holds one slot while it polls, so the dispatch order is a matter
      // of efficiency, not of correctness; the poll is bounded all the same)
      BoundedPoll poll;
      while (__hip_atomic_load(f.done + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != f.seq)
      {
        __builtin_amdgcn_s_sleep(2);
        if (poll.expired())
        {
          gave_up = true;
          break;
        }
      }
    }
    // what the other blocks wrote before their `done` words is read after this point only -- with
    // agent-scope (sc1) loads of what they stored with agent-scope stores and had acknowledged before
    // `done`: program order suffices, no cache invalidation (see small_final_reduction,
    // ndt2d_match_small.hip); the compiler keeps the order:
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    // (any thread?  through the block's scratch row -- one word per wave -- rather than the
    // device library's workgroup reduction, which brings static LDS of its own)
    {
      uint32_t * wave_gave_up = reinterpret_cast<uint32_t *>(chunk_sums + kChunks);
      const bool wave_flag = __builtin_amdgcn_ballot_w64(gave_up) != 0ull;
      if ((threadIdx.x & (kWave - 1)) == 0) wave_gave_up[threadIdx.x >> 6] = wave_flag ? 1u : 0u;
      __syncthreads();
      bool any = false;
      for (uint32_t w = 0; w < kFewThreads / kWave; ++w) any |= wave_gave_up[w] != 0u;
      gave_up = any;
      __syncthreads();
    }
    if (gave_up)
    {
      // a pose's score never came: the flag says so (the host's wait returns NDT2D_ERR_HIP for
      // this call; no trap -- the context stays usable)
      if (threadIdx.x == 0) raise_host_flag(reinterpret_cast<double *>(f.flag), f.seq | kHostFlagGaveUp);
      return;
    }
    if (!f.stats)
    {
      if (threadIdx.x == 0) raise_host_flag(reinterpret_cast<double *>(f.flag), f.seq);
      return;
    }
  }
  else
  {
  if (ARG_BEAMS && i == 0)
  {
    for (uint32_t k = threadIdx.x; k < 2 * a.n_beams; k += kFewThreads) f.beams_out[k] = arg_beams.xy[k];
  }
  const double * pose = a.poses_xyt != nullptr ? a.poses_xyt + 3 * static_cast<size_t>(i) : few.xyt + 3 * i;
  const double x = pose[0], y = pose[1], th = pose[2];
  // toEigen(pose): AngleAxisd(theta, Z) -> [[c,-s],[s,c]] (conversions.hpp:64-68)
  double s, c;
  sincos(th, &s, &c);
  for (uint32_t k = threadIdx.x; k < a.n_beams; k += kFewThreads)
  {
    double2 p;
    if (ARG_BEAMS)
    {
      p.x = arg_beams.xy[2 * k];
      p.y = arg_beams.xy[2 * k + 1];
    }
    else
    {
      p = reinterpret_cast<const double2 *>(a.beams_xy)[k];
    }
    // p = t * (x, y, 1) (:172-173)
    const double px = x + (c * p.x - s * p.y);
    const double py = y + (s * p.x + c * p.y);
    const double e = indexed_exponent<false>(g, nullptr, cell_index<POW2>(g, px, py), px, py);
    terms[k] = exp_score(e);
  }
  __syncthreads();
  const uint32_t chunk_len = (a.n_beams + kChunks - 1) / kChunks;
  if (threadIdx.x < kChunks)
  {
    const uint32_t k0 = min(threadIdx.x * chunk_len, a.n_beams);
    const uint32_t k1 = min(k0 + chunk_len, a.n_beams);
    double csum = 0.0;
    for (uint32_t k = k0; k < k1; ++k) csum += terms[k];
    chunk_sums[threadIdx.x] = csum;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double sum = chunk_sums[0];
#pragma unroll
    for (int j = 1; j < kChunks; ++j) sum += chunk_sums[j];
    // score = sum of (-likelihood) / n  ==  -(sum) / n (:175-177)
    const double score = -sum / static_cast<double>(a.n_beams);
    if (f.flag == nullptr)
    {
      a.scores[i] = score;
    }
    else
    {
      // scores go to host-coherent memory (or, with f.stats, to the device array the
      // statistics read); once the store has been acknowledged the pose's `done` word says so
      if (f.stats)
      {
        __hip_atomic_store(f.dev_scores + i, score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f.dev_poses != nullptr)
        {
          __hip_atomic_store(f.dev_poses + 3 * static_cast<size_t>(i) + 0, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(f.dev_poses + 3 * static_cast<size_t>(i) + 1, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(f.dev_poses + 3 * static_cast<size_t>(i) + 2, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      else
      {
        store_host(a.scores + i, score);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (n > 1)
      {
        if (!NDT2D_TEST_DROPS_DONE(i)) __hip_atomic_store(f.done + i, f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      else if (!f.stats)
      {
        raise_host_flag(reinterpret_cast<double *>(f.flag), f.seq);   // a single pose signals for itself
      }
    }
  }
  // (a single pose with statistics goes on by itself; every other pose block is done)
  if (!(f.flag != nullptr && f.stats && n == 1)) return;
  __syncthreads();
  }

  // ---- updateStatistics by the launch's last block (src/particle_filter.cpp:163-218) ----
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = threadIdx.x >> 6;
  double * row = chunk_sums + kChunks;          // 8 doubles of scratch
  double * wave_rows = terms;                   // [4 waves][8]: needs n_beams >= 32 or the pad below
  auto pose_of = [&](uint32_t k, double & px, double & py, double & pt) {
    if (f.dev_poses != nullptr)
    {
      const double * q = f.dev_poses + 3 * static_cast<size_t>(k);
      px = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      py = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pt = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    const double * q = a.poses_xyt != nullptr ? a.poses_xyt + 3 * static_cast<size_t>(k) : few.xyt + 3 * k;
    px = q[0];
    py = q[1];
    pt = q[2];
  };
  // block-wide sum of eight per-thread values, fixed order; the result in row[0..7]
  auto block_sum8 = [&](double (&v)[8]) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = wave_sum_to_last_lane(v[k]);
    if (lane == kWave - 1)
    {
#pragma unroll
      for (int k = 0; k < 8; ++k) wave_rows[wave * 8 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 8)
    {
      double t = wave_rows[threadIdx.x];
      for (int w = 1; w < kFewThreads / kWave; ++w) t += wave_rows[w * 8 + threadIdx.x];
      row[threadIdx.x] = t;
    }
    __syncthreads();
  };
  double st[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) st[k] = 0.0;
  for (uint32_t k = threadIdx.x; k < n; k += kFewThreads)
  {
    const double w = __hip_atomic_load(f.dev_scores + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double px, py, pt, ps, pc;
    pose_of(k, px, py, pt);
    sincos(pt, &ps, &pc);
    st[0] += w;
    st[1] += w * px;
    st[2] += w * py;
    st[3] += w * pc;
    st[4] += w * ps;
    st[5] += w * px * px;
    st[6] += w * px * py;
    st[7] += w * py * py;
  }
  block_sum8(st);
  const double sum_w = row[0];
  const double mean_x = row[1] / sum_w, mean_y = row[2] / sum_w;
  const double mean_th = atan2(row[4] / sum_w, row[3] / sum_w);
  const double cov_xx = row[5] / sum_w - mean_x * mean_x;
  const double cov_xy = row[6] / sum_w - mean_x * mean_y;
  const double cov_yy = row[7] / sum_w - mean_y * mean_y;
  double var[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) var[k] = 0.0;
  for (uint32_t k = threadIdx.x; k < n; k += kFewThreads)
  {
    const double w = __hip_atomic_load(f.dev_scores + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / sum_w;
    store_host(a.scores + k, w);                               // the normalised weight (:171-174)
    double px, py, pt;
    pose_of(k, px, py, pt);
    const double d = shortest_angular_distance(pt, mean_th);   // (:213-217)
    var[0] += w * d * d;
  }
  block_sum8(var);
  if (threadIdx.x == 0)
  {
    store_host(f.host_out + 0, sum_w);
    store_host(f.host_out + 1, mean_x);
    store_host(f.host_out + 2, mean_y);
    store_host(f.host_out + 3, mean_th);
    store_host(f.host_out + 4, cov_xx);
    store_host(f.host_out + 5, cov_xy);
    store_host(f.host_out + 6, cov_yy);
    store_host(f.host_out + 7, row[0]);
  }
  // every thread's weights have been acknowledged before the flag leaves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) raise_host_flag(reinterpret_cast<double *>(f.flag), f.seq);
}

size_t compact_lds_bytes(const PosesArgs & args, int threads, uint32_t split, bool screen)
{
  const size_t fixed = threads == 1024 ? (screen ? CompactLayout<1024, true>::kFixedDoubles
                                                 : CompactLayout<1024, false>::kFixedDoubles)
                       : threads == 512 ? (screen ? CompactLayout<512, true>::kFixedDoubles
                                                  : CompactLayout<512, false>::kFixedDoubles)
                                        : (screen ? CompactLayout<256, true>::kFixedDoubles
                                                  : CompactLayout<256, false>::kFixedDoubles);
  // f64 beams + (screening) their f32 copy, 1.5 doubles per coordinate
  const size_t beams = static_cast<size_t>(3) * ((args.n_beams + 1) & ~1u) + kScreenPadFloats / 2;
  const size_t words = args.coarse_log2 > 0 ? poses_coarse_words(args.grid, args.coarse_log2)
                                            : (static_cast<size_t>(args.grid.ncell) + 1 + 31) / 32;
  const size_t chunk_sums =
    split > 1 ? static_cast<size_t>(threads / kWave / split) * kChunks * kWave : 0;
  return (fixed + beams + chunk_sums) * sizeof(double) + ((words + 3) & ~size_t(3)) * 4;
}

// Waves sharing one group of 64 poses: enough to put ~8 waves on every SIMD (the
// gathers of phase B are latency bound), at most kChunks and one block.
uint32_t choose_split(const PosesArgs & args, int cus, uint32_t waves_per_block)
{
  const uint64_t groups = (args.n_poses + kWave - 1) / kWave;
  const uint64_t want_waves = static_cast<uint64_t>(cus) * 4 * 8;
  uint32_t split = 1;
  while (split < static_cast<uint32_t>(kChunks) && split < waves_per_block &&
         groups * split < want_waves)
  {
    split *= 2;
  }
  return split;
}

// (A/B knob NDT2D_POSES_EIGHT_WAVES: 1 forces the eight-wave groups wherever they fit, 0 turns them off.)
// Measured (experiments/particles_rounds.py, cfg-3 map, kernel ms, four- vs eight-wave groups): 16,384
// particles 0.0377 / 0.0262, 32,768 0.0425 / 0.0351, 81,920 0.0758 / 0.0694, 98,304 0.0832 / 0.0813,
// 100,000 0.0848 / 0.0865: worth it while the eight-wave launch stays within three rounds of the chip's
// wave slots.  The scores do not depend on it (chunk sums are added in a fixed order whatever the split).
bool poses_eight_wave_groups(uint64_t groups, int cus)
{
  if (const char * env = std::getenv("NDT2D_POSES_EIGHT_WAVES")) return env[0] == '1';
  return groups * 8 <= static_cast<uint64_t>(cus) * 16 * 3;
}

template <int THREADS>
hipError_t launch_compact(const PosesArgs & args, uint32_t blocks, uint32_t split,
                          size_t lds_bytes, bool screen, hipStream_t stream)
{
  auto launch = [&](auto kernel) -> hipError_t {
    const hipError_t e = prepare_absolute_lds_kernel(reinterpret_cast<const void *>(kernel), lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), lds_bytes, stream, args, split);
    return hipGetLastError();
  };
  if (args.coarse_log2 > 0)
  {
    // (one bit per block of cells: large grids only, i.e. the 1024-thread geometry)
    if (THREADS != 1024 || args.coarse_bits == nullptr) return hipErrorInvalidValue;
    if (screen)
    {
      return args.grid.pow2 ? launch(score_poses_compact_kernel<1024, true, true, true>)
                            : launch(score_poses_compact_kernel<1024, false, true, true>);
    }
    return args.grid.pow2 ? launch(score_poses_compact_kernel<1024, true, false, true>)
                          : launch(score_poses_compact_kernel<1024, false, false, true>);
  }
  if (screen)
  {
    return args.grid.pow2 ? launch(score_poses_compact_kernel<THREADS, true, true>)
                          : launch(score_poses_compact_kernel<THREADS, false, true>);
  }
  return args.grid.pow2 ? launch(score_poses_compact_kernel<THREADS, true, false>)
                        : launch(score_poses_compact_kernel<THREADS, false, false>);
}

}  // namespace

size_t few_lds_bytes(const PosesArgs & args)
{
  // terms (at least the 4 x 8 doubles the statistics pass puts there) + chunk sums + scratch row
  const size_t terms = args.n_beams < 32 ? 32 : args.n_beams;
  return (terms + kChunks + 8) * sizeof(double);
}

bool score_few_supported(const PosesArgs & args, size_t lds_per_block)
{
  return args.n_poses <= kFewPosesMax && few_lds_bytes(args) <= lds_per_block;
}

hipError_t launch_score_few(const PosesArgs & args, const FewPoses * few, const FewOut & out,
                            const double * host_beams, hipStream_t stream)
{
  static const FewPoses none = {};
  const bool arg_beams = host_beams != nullptr && out.beams_out != nullptr && args.n_beams <= kArgBeams;
  if (host_beams != nullptr && !arg_beams) return hipErrorInvalidValue;
  if (args.poses_xyt == nullptr && args.n_poses > kFewPoses) return hipErrorInvalidValue;
  FewBeams fb{};
  if (arg_beams) std::memcpy(fb.xy, host_beams, 2 * static_cast<size_t>(args.n_beams) * sizeof(double));
  const size_t lds_bytes = few_lds_bytes(args);
  auto launch = [&](auto kernel) -> hipError_t {
    if (lds_bytes > 48 * 1024)
    {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds_bytes));
      if (e != hipSuccess) return e;
    }
    // (+ the block that waits for the others, see the kernel)
    const uint32_t extra = (out.flag != nullptr && args.n_poses > 1) ? 1u : 0u;
    static_assert(kFewThreads == 256, "sparse_bytes_blocks counts blocks of 256 lanes");
    static_assert(sizeof(PosesArgs) + sizeof(FewPoses) + sizeof(FewOut) + sizeof(FewBeams) <= 4096,
                  "kernel arguments of score_few_kernel");
    const uint32_t side = out.side.n > 0 ? sparse_bytes_blocks(out.side) : 0u;
    hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(args.n_poses) + extra + side), dim3(kFewThreads),
                       lds_bytes, stream, args, few != nullptr ? *few : none, out, fb);
    return hipGetLastError();
  };
  if (arg_beams)
  {
    return args.grid.pow2 ? launch(score_few_kernel<true, true>) : launch(score_few_kernel<false, true>);
  }
  return args.grid.pow2 ? launch(score_few_kernel<true, false>) : launch(score_few_kernel<false, false>);
}

size_t poses_coarse_words(const GridDesc & g, uint32_t coarse_log2)
{
  const size_t sx = (static_cast<size_t>(g.size_x) + (1u << coarse_log2) - 1) >> coarse_log2;
  const size_t sy = (static_cast<size_t>(g.size_y) + (1u << coarse_log2) - 1) >> coarse_log2;
  return (sx * sy + 1 + 31) / 32;
}

int poses_coarse_log2(const PosesArgs & args_in, size_t lds_per_block)
{
  if (args_in.grid.occ_bits == nullptr || args_in.grid.ncell >= kCellMask) return -1;
  PosesArgs args = args_in;
  for (uint32_t k = 0; k <= 4; ++k)
  {
    args.coarse_log2 = k;
    if (compact_lds_bytes(args, 1024, 1, false) <= lds_per_block) return static_cast<int>(k);
  }
  return -1;
}

bool poses_compact_supported(const PosesArgs & args, size_t lds_per_block)
{
  const int k = poses_coarse_log2(args, lds_per_block);
  return k >= 0 && static_cast<uint32_t>(k) == args.coarse_log2 && (k == 0 || args.coarse_bits != nullptr);
}

namespace
{

// bit (by, bx) = some cell of the block of 2^k x 2^k cells at (bx << k, by << k) holds a
// distribution; bit sx * sy ("outside") and the padding of the last word stay zero
__global__ void __launch_bounds__(256) coarse_bits_kernel(const GridDesc g, uint32_t k, uint32_t sx, uint32_t sy,
                                                          uint32_t * out)
{
  const uint32_t word = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n_bits = sx * sy;
  if (word >= (n_bits + 1 + 31) / 32) return;
  uint32_t v = 0;
  for (uint32_t j = 0; j < 32; ++j)
  {
    const uint32_t b = word * 32 + j;
    if (b >= n_bits) break;
    const uint32_t by = b / sx, bx = b - by * sx;
    bool any = false;
    for (uint32_t cy = by << k; cy < min((by + 1) << k, g.size_y) && !any; ++cy)
    {
      for (uint32_t cx = bx << k; cx < min((bx + 1) << k, g.size_x); ++cx)
      {
        const uint32_t cell = cy * g.size_x + cx;
        if ((g.occ_bits[cell >> 5] >> (cell & 31u)) & 1u)
        {
          any = true;
          break;
        }
      }
    }
    v |= any ? (1u << j) : 0u;
  }
  out[word] = v;
}

}  // namespace

hipError_t poses_coarse_bits_launch(const GridDesc & g, uint32_t coarse_log2, uint32_t * out, hipStream_t stream)
{
  const uint32_t sx = (g.size_x + (1u << coarse_log2) - 1) >> coarse_log2;
  const uint32_t sy = (g.size_y + (1u << coarse_log2) - 1) >> coarse_log2;
  const uint32_t words = static_cast<uint32_t>(poses_coarse_words(g, coarse_log2));
  hipLaunchKernelGGL(coarse_bits_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, g, coarse_log2, sx, sy, out);
  return hipGetLastError();
}

hipError_t launch_poses_compact(const PosesArgs & args_in, int cus, bool screen,
                                hipStream_t stream, uint32_t * blocks_out)
{
  PosesArgs args = args_in;
  // screening needs a finite bound of the beams' reach, and a grid the FP32 cell
  // coordinate resolves to well under a cell
  const double reach_cells = args.beam_rmax / args.grid.cell_size;
  const double magnitude =
    static_cast<double>(args.grid.size_x > args.grid.size_y ? args.grid.size_x : args.grid.size_y) +
    2.0 * reach_cells + 4.0;
  if (!(magnitude < 65536.0)) screen = false;   // also catches NaN / inf reach
  if (args.grid.size_x >= (1u << 23)) screen = false;  // signed 24-bit multiply of the bitmap index
  args.screen_guard = screen ? screen_guard(static_cast<float>(magnitude)) : 0.0f;
  // Small LDS image: 256-thread blocks, several per CU.  Large occupancy bitmap:
  // one 1024-thread block per CU shares it.
  uint32_t split = choose_split(args, cus, 4);
  const size_t small = compact_lds_bytes(args, 256, split, screen);
  // (the coarse-bitmap kernels exist in the 1024-thread geometry only -- also when the coarse
  // bitmap itself is small: a grid of a million cells screens with 33 KB of bits)
  const bool use_small = small <= 48 * 1024 && args.coarse_log2 == 0;
  const int threads = use_small ? 256 : 1024;
  const uint32_t waves_per_block = static_cast<uint32_t>(threads / kWave);
  size_t lds_bytes = small;
  if (!use_small)
  {
    split = choose_split(args, cus, waves_per_block);
    lds_bytes = compact_lds_bytes(args, 1024, split, screen);
    while (split > 1 && lds_bytes > kLdsBudget)
    {
      split /= 2;
      lds_bytes = compact_lds_bytes(args, 1024, split, screen);
    }
  }
  const uint64_t groups = (args.n_poses + kWave - 1) / kWave;
  const uint64_t groups_per_block = waves_per_block / split;
  const uint64_t need = (groups + groups_per_block - 1) / groups_per_block;
  uint64_t cap = use_small ? 4096 : static_cast<uint64_t>(cus);
  if (const char * env = std::getenv("NDT2D_POSES_BLOCK_CAP"))   // A/B knob (experiments/particles_rounds.py)
  {
    const long v = std::atol(env);
    if (v > 0 && use_small) cap = static_cast<uint64_t>(v);
  }
  const uint32_t blocks = static_cast<uint32_t>(need < cap ? need : cap);
  if (blocks_out != nullptr) *blocks_out = blocks;
  // Eight waves per group of 64 poses (512-thread blocks, one group each) when the particle set is
  // too small to fill the chip otherwise: a wave then walks one chunk of the beams instead of two.
  if (use_small && poses_eight_wave_groups(groups, cus))
  {
    const uint32_t split8 = choose_split(args, cus, 8);
    const size_t lds8 = compact_lds_bytes(args, 512, split8, screen);
    if (split8 == 8 && lds8 <= 48 * 1024)
    {
      const uint64_t need8 = groups;   // one group per block
      const uint32_t blocks8 = static_cast<uint32_t>(need8 < cap ? need8 : cap);
      if (blocks_out != nullptr) *blocks_out = blocks8;
      return launch_compact<512>(args, blocks8, split8, lds8, screen, stream);
    }
  }
  return use_small ? launch_compact<256>(args, blocks, split, lds_bytes, screen, stream)
                   : launch_compact<1024>(args, blocks, split, lds_bytes, screen, stream);
}

}  // namespace ndt2d

#ifdef NDT2D_TEST_HOOKS
// Test builds only (libndt2d_hip_hooks.so; not declared in include/ndt2d_hip.h): make the producer
// of record / pose `which - 1` of this translation unit's kernels withhold its `done` word
// (0: normal operation) -- tests/test_gpu_bounded_poll.py.
extern "C" int ndt2d_test_drop_done_few(int which)
{
  return hipMemcpyToSymbol(HIP_SYMBOL(ndt2d::g_test_drop_done), &which, sizeof(int)) == hipSuccess ? 0 : 3;
}
#endif
