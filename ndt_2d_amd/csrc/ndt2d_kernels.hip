// Hand-written HIP kernels for ndt_2d's NDT scoring hot path, gfx950 (MI355X).
//
//   match_kernel        ScanMatcherNDT::matchScan's exhaustive (theta, x, y)
//                       search, reference src/scan_matcher_ndt.cpp:103-143:
//                       one 64-lane wavefront per candidate pose, lanes stride
//                       the beams, wave-level reduction of the per-point
//                       likelihoods (NDT::likelihood, src/ndt_model.cpp:178-187).
//   score_poses_kernel  ScanMatcherNDT::scorePoints for a batch of poses
//                       (src/scan_matcher_ndt.cpp:156-178), i.e. the loop body
//                       of ParticleFilter::measure (src/particle_filter.cpp:81-87):
//                       one candidate pose (particle) per lane, beams in order.
//
// All arithmetic is IEEE double in the reference's operation order; this file
// is compiled with -ffp-contract=off so no multiply-add is fused that the
// reference's x86-64 build keeps separate.  No MFMA: there is no dense
// contraction on this path.
#include <mutex>
#include <string>
#include <vector>

#include "ndt2d_device_fn.h"

namespace ndt2d
{

namespace
{

constexpr int kMatchThreads = 1024;
constexpr int kMatchWaves = kMatchThreads / kWave;

// ---------------------------------------------------------------------------
// matchScan search: one wavefront per candidate pose.
//
// A work item is `chunk` consecutive (ix, iy) candidates of one theta; wave w of
// the whole launch takes items w, w + W, ...  Per item the wave rotates its
// NBL beams per lane into registers (points_outer, :108-115), then per
// candidate shifts them (:121-125), gathers the cell records from the LDS copy
// of the grid and reduces the likelihoods across the wave.
// ---------------------------------------------------------------------------
template <int NBL, bool LDS_GRID, bool POW2>
__global__ void __launch_bounds__(kMatchThreads) match_kernel(const MatchArgs a)
{
  extern __shared__ __align__(16) double lds_cells[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  if (LDS_GRID)
  {
    stage_grid_to_lds(a.grid, lds_cells);
    __syncthreads();
  }

  const uint32_t n_lin = a.n_lin;
  const uint32_t m = n_lin * n_lin;                       // candidates per theta
  const uint32_t cps = (m + a.chunk - 1) / a.chunk;       // chunks per theta
  const uint64_t n_items = static_cast<uint64_t>(a.th_end - a.th_begin) * cps;
  const uint32_t n_workers = gridDim.x * kMatchWaves;
  const uint32_t worker = wave * gridDim.x + blockIdx.x;  // interleave CUs

  // Lane L < 10 owns covariance accumulator L of {k00,k01,k02,k11,k12,k22,u0,u1,u2,s}
  // (src/scan_matcher_ndt.cpp:137-140): acc += (xa * xb) * score with
  // xa, xb in {dx, dy, dth, 1}; (x * 1) * score == x * score exactly.
  int sel_a = 4, sel_b = 4;  // 0: dx, 1: dy, 2: dth, 3: 1.0, 4: 0.0
  switch (lane)
  {
    case 0: sel_a = 0; sel_b = 0; break;
    case 1: sel_a = 0; sel_b = 1; break;
    case 2: sel_a = 0; sel_b = 2; break;
    case 3: sel_a = 1; sel_b = 1; break;
    case 4: sel_a = 1; sel_b = 2; break;
    case 5: sel_a = 2; sel_b = 2; break;
    case 6: sel_a = 0; sel_b = 3; break;
    case 7: sel_a = 1; sel_b = 3; break;
    case 8: sel_a = 2; sel_b = 3; break;
    case 9: sel_a = 3; sel_b = 3; break;
    default: break;
  }

  double best_s = 0.0;       // `double best_score = 0;` (:83)
  double best_i = kNoIndex;
  double acc = 0.0;

  for (uint64_t item = worker; item < n_items; item += n_workers)
  {
    const uint32_t t_local = static_cast<uint32_t>(item / cps);
    const uint32_t ith = a.th_begin + t_local * a.th_stride;
    const uint32_t c = static_cast<uint32_t>(item % cps);
    const double ct = a.cos_th[ith];
    const double st = a.sin_th[ith];
    const double dt = a.dth[ith];

    // points_outer (:108-115); NBL == 0 is the any-beam-count variant that
    // rotates on the fly instead of keeping points_outer in registers.
    constexpr int kRegs = NBL > 0 ? NBL : 1;
    double ox[kRegs], oy[kRegs];
#pragma unroll
    for (int j = 0; j < NBL; ++j)
    {
      const uint32_t b = lane + kWave * j;
      if (b < a.n_beams)
      {
        const double2 p = reinterpret_cast<const double2 *>(a.beams_xy)[b];
        ox[j] = p.x * ct - p.y * st + a.pose_x;
        oy[j] = p.x * st + p.y * ct + a.pose_y;
      }
      else
      {
        ox[j] = kPadCoord;
        oy[j] = kPadCoord;
      }
    }

    const uint32_t f0 = c * a.chunk;
    const uint32_t f1 = min(f0 + a.chunk, m);
    uint32_t ix = f0 / n_lin;
    uint32_t iy = f0 - ix * n_lin;
    for (uint32_t f = f0; f < f1; ++f)
    {
      const double dx = a.dlin[ix];
      const double dy = a.dlin[iy];

      // points_inner + NDT::likelihood(points_inner) (:121-127)
      double sum = 0.0;
      if (NBL > 0)
      {
#pragma unroll
        for (int j = 0; j < NBL; ++j)
        {
          const double px = ox[j] + dx;
          const double py = oy[j] + dy;
          sum += point_likelihood<LDS_GRID, POW2>(a.grid, lds_cells, px, py);
        }
      }
      else
      {
        for (uint32_t b = lane; b < a.n_beams; b += kWave)
        {
          const double2 p = reinterpret_cast<const double2 *>(a.beams_xy)[b];
          const double px = (p.x * ct - p.y * st + a.pose_x) + dx;
          const double py = (p.x * st + p.y * ct + a.pose_y) + dy;
          sum += point_likelihood<LDS_GRID, POW2>(a.grid, lds_cells, px, py);
        }
      }
      sum = wave_sum(sum);
      const double score = -sum;

      const double flat = static_cast<double>(static_cast<uint64_t>(ith) * m + f);
      if (score < 0.0) merge_best(score, flat, best_s, best_i);   // (:128; marks near-ties, see merge_best)

      const double xa = sel_a == 0 ? dx : (sel_a == 1 ? dy : (sel_a == 2 ? dt : (sel_a == 3 ? 1.0 : 0.0)));
      const double xb = sel_b == 0 ? dx : (sel_b == 1 ? dy : (sel_b == 2 ? dt : (sel_b == 3 ? 1.0 : 0.0)));
      acc += (xa * xb) * score;

      if (a.scores != nullptr && lane == 0)
      {
        a.scores[static_cast<uint64_t>(t_local) * m + f] = score;
      }

      if (++iy == n_lin)
      {
        iy = 0;
        ++ix;
      }
    }
  }

  double * out = a.partials + static_cast<size_t>(worker) * kRecord;
  if (lane == 0)
  {
    out[0] = best_s;
    out[1] = best_i;
  }
  if (lane < 10)
  {
    out[2 + lane] = acc;
  }
}

// Reduction of partial records, fixed order (deterministic).
// record = {best_score, best_index (exact double, -1 if none), acc[10]}.
// Block b reduces records [b * per_block, min(n, (b + 1) * per_block)) to one.  As the
// last stage (one block, `final`), it also applies "no candidate scored below 0 ->
// no index" and writes the result record(s); as a first stage it writes record b of
// `out`.
__global__ void __launch_bounds__(256) match_reduce_kernel(const double * partials, uint32_t n,
                                                           uint32_t per_block, double * out,
                                                           double * out2, int final)
{
  __shared__ double sh[256 * kRecord];
  const int t = threadIdx.x;
  const uint32_t begin = blockIdx.x * per_block;
  const uint32_t end = min(n, begin + per_block);
  double v[kRecord];
  v[0] = 0.0;
  v[1] = kNoIndex;
#pragma unroll
  for (int k = 2; k < kRecord; ++k) v[k] = 0.0;
  for (uint32_t w = begin + t; w < end; w += 256)
  {
    const double * p = partials + static_cast<size_t>(w) * kRecord;
    merge_best(p[0], p[1], v[0], v[1]);
#pragma unroll
    for (int k = 2; k < kRecord; ++k) v[k] += p[k];
  }
#pragma unroll
  for (int k = 0; k < kRecord; ++k) sh[t * kRecord + k] = v[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1)
  {
    if (t < s)
    {
      double * mine = sh + t * kRecord;
      const double * other = sh + (t + s) * kRecord;
      merge_best(other[0], other[1], mine[0], mine[1]);
#pragma unroll
      for (int k = 2; k < kRecord; ++k) mine[k] += other[k];
    }
    __syncthreads();
  }
  if (t < kRecord)
  {
    double val = sh[t];
    if (final)
    {
      if (t == 1 && !(sh[0] < 0.0)) val = -1.0;
      out[t] = val;
      if (out2 != nullptr) out2[t] = val;
    }
    else
    {
      out[static_cast<size_t>(blockIdx.x) * kRecord + t] = val;
    }
  }
}

// Last reduction stage, latency-shaped (it ends every matchScan call): 11 waves, wave 0
// reduces the (score, index) pairs, wave 1 + k the accumulator column k; a lane adds the
// records lane, lane + 64, ... in order and the wave finishes over the DPP network --
// fixed order, no LDS, no barrier before the results are written.  It applies "no
// candidate scored below 0 -> no index" and writes the record to out (device), out2
// (device, optional) and host_out (host-coherent memory, optional) followed there by
// `seq` at host_out[kHostFlagSlot], which the host spins on.
constexpr int kFinalThreads = 11 * kWave;
constexpr uint32_t kFinalAloneMax = 1024;   // records the last stage takes on its own (launch_match)
__global__ void __launch_bounds__(kFinalThreads) match_reduce_final_kernel(
  const double * partials, uint32_t n, double * out, double * out2, double * host_out,
  unsigned long long seq)
{
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = threadIdx.x >> 6;
  if (wave == 0)
  {
    double bs = 0.0, bi = kNoIndex;
    for (uint32_t r = lane; r < n; r += kWave)
    {
      const double2 p = *reinterpret_cast<const double2 *>(partials + static_cast<size_t>(r) * kRecord);
      merge_best(p.x, p.y, bs, bi);
    }
    wave_best_to_last_lane(bs, bi);
    if (lane == kWave - 1)
    {
      if (!(bs < 0.0)) bi = -1.0;
      out[0] = bs;
      out[1] = bi;
      if (out2 != nullptr)
      {
        out2[0] = bs;
        out2[1] = bi;
      }
      if (host_out != nullptr)
      {
        store_host(host_out + 0, bs);
        store_host(host_out + 1, bi);
      }
    }
  }
  else
  {
    const uint32_t k = 1 + wave;   // columns 2 .. 11
    double v = 0.0;
    for (uint32_t r = lane; r < n; r += kWave) v += partials[static_cast<size_t>(r) * kRecord + k];
    v = wave_sum_to_last_lane(v);
    if (lane == kWave - 1)
    {
      out[k] = v;
      if (out2 != nullptr) out2[k] = v;
      if (host_out != nullptr) store_host(host_out + k, v);
    }
  }
  if (host_out != nullptr)
  {
    // every wave's store has been acknowledged before the flag leaves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) raise_host_flag(host_out + kHostFlagSlot, seq);
  }
}

// Near-tie adjudication, device side (see merge_best in ndt2d_device_fn.h): every candidate with
// an index below `hi` that scored below 0 and within the tolerance of the best.  Runs only behind a
// search whose winner came back marked.
__global__ void __launch_bounds__(256) collect_near_kernel(const double * scores, uint64_t n, uint64_t hi,
                                                           const double * record, double rel, double abs_tol,
                                                           unsigned long long * out, uint32_t cap)
{
  const double best = record[0];
  const double limit = best + (fabs(best) * rel + abs_tol);
  const uint64_t end = hi < n ? hi : n;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < end; i += static_cast<uint64_t>(gridDim.x) * 256ull)
  {
    const double s = scores[i];
    if (s < 0.0 && s <= limit)
    {
      const unsigned long long k = atomicAdd(out, 1ull);
      if (k < cap) out[1 + k] = i;
    }
  }
}

// ---------------------------------------------------------------------------
// Batched scorePoints: one pose (particle) per lane; beams are walked in order,
// so each lane's sum has exactly the reference's sequential order (:168-175).
// ---------------------------------------------------------------------------
template <int THREADS, bool LDS_GRID, bool POW2>
__global__ void __launch_bounds__(THREADS) score_poses_kernel(const PosesArgs a)
{
  // One dynamic LDS region (keeps its base 16-byte aligned):
  // [per-wave statistics][beams][grid records]
  extern __shared__ __align__(16) double lds[];
  double * sh_stats = lds;                                    // [THREADS/64][8]
  double * lds_beams = lds + (THREADS / kWave) * 8;           // [n_beams][2]
  double * lds_cells = lds_beams + 2 * ((a.n_beams + 1) & ~1u);

  for (uint32_t i = threadIdx.x; i < 2 * a.n_beams; i += THREADS)
  {
    lds_beams[i] = a.beams_xy[i];
  }
  if (LDS_GRID)
  {
    stage_grid_to_lds(a.grid, lds_cells);
  }
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  double st[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) st[k] = 0.0;

  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * THREADS;
  const uint64_t n_round = (a.n_poses + stride - 1) / stride * stride;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * THREADS + threadIdx.x; i < n_round;
       i += stride)
  {
    const bool valid = i < a.n_poses;
    double x = 0.0, y = 0.0, th = 0.0;
    if (valid)
    {
      x = a.poses_xyt[3 * i];
      y = a.poses_xyt[3 * i + 1];
      th = a.poses_xyt[3 * i + 2];
    }
    // toEigen(pose): AngleAxisd(theta, Z) -> [[c,-s],[s,c]] (conversions.hpp:64-68)
    double s, c;
    sincos(th, &s, &c);

    double sum = 0.0;
    for (uint32_t k = 0; k < a.n_beams; ++k)
    {
      const double2 p = reinterpret_cast<const double2 *>(lds_beams)[k];
      // p = t * (x, y, 1) (:172-173): translation + (c*x + (-s)*y), (s*x + c*y)
      const double px = x + (c * p.x - s * p.y);
      const double py = y + (s * p.x + c * p.y);
      sum += point_likelihood<LDS_GRID, POW2>(a.grid, lds_cells, px, py);
    }
    // score += -likelihood  ==  -(sum of likelihoods), bit-for-bit; (:177)
    const double score = -sum / static_cast<double>(a.n_beams);
    if (valid)
    {
      a.scores[i] = score;
      // sums for ParticleFilter::updateStatistics (particle_filter.cpp:166-200)
      const double w = score;
      st[0] += w;
      st[1] += w * x;
      st[2] += w * y;
      st[3] += w * c;
      st[4] += w * s;
      st[5] += w * x * x;
      st[6] += w * x * y;
      st[7] += w * y * y;
    }
  }

  if (a.partials != nullptr)
  {
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = wave_sum(st[k]);
    if (lane == 0)
    {
#pragma unroll
      for (int k = 0; k < 8; ++k) sh_stats[wave * 8 + k] = st[k];
    }
    __syncthreads();
    if (threadIdx.x < 8)
    {
      double v = 0.0;
      for (int w = 0; w < THREADS / kWave; ++w) v += sh_stats[w * 8 + threadIdx.x];
      a.partials[static_cast<size_t>(blockIdx.x) * 8 + threadIdx.x] = v;
    }
  }
}

// host_out (optional, host-coherent memory as the device addresses it): the eight sums are
// stored there as well, system scope, followed by `seq` at host_out[kPoseSumsFlagOffset] -- a
// host thread spins on it instead of synchronising the stream (the sharded particle path:
// every device's sums meet on the host between its two launches).
__global__ void __launch_bounds__(256) poses_reduce_kernel(const double * partials,
                                                           uint32_t n_blocks, double * stats,
                                                           double * host_out, unsigned long long seq)
{
  __shared__ double sh[256 * 8];
  const int t = threadIdx.x;
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.0;
  for (uint32_t b = t; b < n_blocks; b += 256)
  {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += partials[static_cast<size_t>(b) * 8 + k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[t * 8 + k] = v[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1)
  {
    if (t < s)
    {
#pragma unroll
      for (int k = 0; k < 8; ++k) sh[t * 8 + k] += sh[(t + s) * 8 + k];
    }
    __syncthreads();
  }
  if (t < 8) stats[t] = sh[t];
  if (host_out != nullptr)
  {
    // (the eight stores and the flag come from ONE lane, in program order, each waited for)
    if (t == 0)
    {
#pragma unroll
      for (int k = 0; k < 8; ++k) store_host(host_out + k, sh[k]);
      raise_host_flag(host_out + kPoseSumsFlagOffset, seq);
    }
  }
}

__global__ void __launch_bounds__(64) sum_moment_rows_kernel(const double * rows, uint32_t n_rows, double * out)
{
  if (threadIdx.x >= 8) return;
  double v = rows[threadIdx.x];
  for (uint32_t r = 1; r < n_rows; ++r) v += rows[static_cast<size_t>(r) * 8 + threadIdx.x];
  out[threadIdx.x] = v;
}

constexpr uint32_t kMaxMatchBlocks = 512;
constexpr uint32_t kMaxPosesBlocks = 4096;
constexpr size_t kLdsPerCu = 160 * 1024;

struct DeviceLimits
{
  int cus;
  size_t lds_per_block;
};

// Immutable per-device facts, queried once per device (hipGetDeviceProperties is
// far too slow for the launch path of a 100-microsecond search).
DeviceLimits device_limits()
{
  constexpr int kMaxDevices = 64;
  static std::mutex mu;
  static bool known[kMaxDevices] = {};
  static DeviceLimits cache[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
  {
    return DeviceLimits{256, kLdsPerCu};
  }
  std::lock_guard<std::mutex> lock(mu);
  if (!known[dev])
  {
    DeviceLimits lim{256, kLdsPerCu};
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    {
      lim.cus = v;
    }
    v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
        v > 0)
    {
      lim.lds_per_block = static_cast<size_t>(v);
    }
    cache[dev] = lim;
    known[dev] = true;
  }
  return cache[dev];
}

template <int NBL, bool LDS_GRID, bool POW2>
hipError_t launch_match_variant(const MatchArgs & args, uint32_t blocks, size_t lds_bytes,
                                hipStream_t stream)
{
  auto kernel = match_kernel<NBL, LDS_GRID, POW2>;
  if (lds_bytes > 48 * 1024)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kMatchThreads), lds_bytes, stream, args);
  return hipGetLastError();
}

template <bool LDS_GRID, bool POW2>
hipError_t dispatch_match_nbl(const MatchArgs & args, uint32_t blocks, size_t lds_bytes,
                              hipStream_t stream)
{
  const uint32_t nbl = (args.n_beams + kWave - 1) / kWave;
  if (nbl <= 2) return launch_match_variant<2, LDS_GRID, POW2>(args, blocks, lds_bytes, stream);
  if (nbl <= 4) return launch_match_variant<4, LDS_GRID, POW2>(args, blocks, lds_bytes, stream);
  if (nbl <= 8) return launch_match_variant<8, LDS_GRID, POW2>(args, blocks, lds_bytes, stream);
  if (nbl <= 12) return launch_match_variant<12, LDS_GRID, POW2>(args, blocks, lds_bytes, stream);
  if (nbl <= 16) return launch_match_variant<16, LDS_GRID, POW2>(args, blocks, lds_bytes, stream);
  return launch_match_variant<0, LDS_GRID, POW2>(args, blocks, lds_bytes, stream);
}

}  // namespace

hipError_t prepare_absolute_lds_kernel(const void * kernel, size_t dynamic_lds_bytes)
{
  struct Seen
  {
    hipError_t status;
    size_t granted[64];   // per device: the dynamic LDS size hipFuncSetAttribute has been given
  };
  static std::mutex mu;
  static std::unordered_map<const void *, Seen> seen;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) device = 0;
  std::lock_guard<std::mutex> lock(mu);
  auto it = seen.find(kernel);
  if (it == seen.end())
  {
    Seen s{};
    hipFuncAttributes attr{};
    s.status = hipFuncGetAttributes(&attr, kernel);
    if (s.status == hipSuccess && attr.sharedSizeBytes != 0) s.status = hipErrorInvalidDeviceFunction;
    it = seen.emplace(kernel, s).first;
  }
  if (it->second.status != hipSuccess) return it->second.status;
  if (dynamic_lds_bytes > 48 * 1024 && dynamic_lds_bytes > it->second.granted[device])
  {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(dynamic_lds_bytes));
    if (e != hipSuccess) return e;
    it->second.granted[device] = dynamic_lds_bytes;
  }
  return hipSuccess;
}


// Workspace layout: [work-item counters, 2 KB][256 records of the first reduction
// stage for each of up to kMaxLaneSlabs slabs][partial records: one per wave (wave
// mapping) or per work item of a slab (lane mapping)]
constexpr size_t kCounterDoubles = static_cast<size_t>(kItemShards) * kItemShardStride * sizeof(uint32_t) / sizeof(double);
constexpr size_t kWorkspaceHead = kCounterDoubles + 256 * static_cast<size_t>(kMaxLaneSlabs) * kRecord;

size_t match_workspace_head_doubles() { return kWorkspaceHead; }

size_t match_workspace_doubles(const MatchArgs & args)
{
  const uint64_t p1 = (args.n_lin + 7) / 8;
  uint32_t slab_th = 0, n_slabs = 0;
  uint64_t records = 0;   // (no cut into slabs: the wave mapping will run)
  if (lane_slabs(args, &slab_th, &n_slabs)) records = static_cast<uint64_t>(slab_th) * p1 * p1;
  const uint64_t waves = static_cast<uint64_t>(kMaxMatchBlocks) * kMatchWaves;
  if (records < waves) records = waves;
  // (the small-lattice search: kSmallMaxItems records and, behind them, as many `done` words)
  const uint64_t small_doubles = kSmallMaxItems * (static_cast<uint64_t>(kRecord) + 1);
  const uint64_t doubles = records * kRecord < small_doubles ? small_doubles : records * kRecord;
  return kWorkspaceHead + static_cast<size_t>(doubles);
}

namespace
{

enum class Mapping { kInvalid, kSmall, kLane, kWave };

// Name of a lane-per-candidate launch: where the records live (0 gathered from HBM, 1 the
// whole grid's in LDS, 2 compacted in LDS), whether the map is one byte per block of grid
// cells (windows wider than 256 cells), whether a candidate's beams were cut into parts.
const char * lane_variant_name(int records, bool block_map, bool pow2, bool parts)
{
  static const std::vector<std::string> names = [] {
    std::vector<std::string> v;
    for (int r = 0; r < 3; ++r)
      for (int b = 0; b < 2; ++b)
        for (int p2 = 0; p2 < 2; ++p2)
          for (int pa = 0; pa < 2; ++pa)
          {
            std::string n = "match/lane-per-candidate/";
            n += r == 2 ? "lds-grid/compact-records" : (r == 1 ? "lds-grid" : "lds-map+global-records");
            if (b) n += "/block-map";
            if (pa) n += "/beam-parts";
            n += p2 ? "/pow2" : "/div";
            v.push_back(n);
          }
    return v;
  }();
  return names[static_cast<size_t>(((records * 2 + (block_map ? 1 : 0)) * 2 + (pow2 ? 1 : 0)) * 2 + (parts ? 1 : 0))].c_str();
}

// Candidate mapping.  Lane-per-candidate (64 translations of one theta step per wave,
// occupancy-map look-up, bit-exact skipping) whenever its LDS image fits: the
// small-lattice form (a block per theta step and up to P tiles, beams split across its
// waves, no pre-kernel, final reduction in the same launch) below kSmallBelowItems... work
// items (by the scan's length), the persistent large-lattice form above.  Otherwise wave-per-candidate.
// (The choice follows the whole lattice, so every shard of a search makes the same one
// and a candidate's score has the same bits whichever rank evaluates it.)
Mapping choose_mapping(const MatchArgs & args, bool outer_available, int force_variant,
                       const DeviceLimits & lim)
{
  const int force_grid = force_variant & kVariantGridMask;
  const uint64_t p1 = (args.n_lin + 7) / 8;
  const uint64_t items = static_cast<uint64_t>(args.n_th) * p1 * p1;
  bool use_lane = outer_available && force_grid != kVariantGlobal &&
                  match_lane_supported(args, lim.lds_per_block);
  bool use_small = force_grid == kVariantAuto && match_small_supported(args, lim.lds_per_block);
  if (force_variant & kVariantWave) use_lane = use_small = false;
  if (force_variant & kVariantSmall) return use_small ? Mapping::kSmall : Mapping::kInvalid;
  if (force_variant & kVariantLane) return use_lane ? Mapping::kLane : Mapping::kInvalid;
  const uint64_t small_below = args.n_beams <= kShortScanBeams ? kSmallBelowItemsShortScan : kSmallBelowItemsLongScan;
  if (use_small && use_lane) return items < small_below ? Mapping::kSmall : Mapping::kLane;
  if (use_small) return Mapping::kSmall;
  // (a small lattice the small-lattice form cannot take -- a window wider than 256 cells:
  // a wave alone with one expensive work item takes 0.2 ms, the wave mapping is the faster one)
  if (use_lane) return items < kWaveBelowItems ? Mapping::kWave : Mapping::kLane;
  return Mapping::kWave;
}

}  // namespace

bool match_needs_device_tables(const MatchArgs & args, bool outer_available, int force_variant)
{
  if (args.n_beams == 0) return true;
  return !(choose_mapping(args, outer_available, force_variant, device_limits()) == Mapping::kSmall &&
           match_small_takes_arg_tables(args));
}

hipError_t launch_match(const MatchArgs & args_in, double * workspace, double * outer,
                        double * record_out, double * record_out2, double * host_record,
                        unsigned long long seq, int force_variant, hipStream_t stream,
                        hipEvent_t ev_main_start, hipEvent_t ev_main_done, LaunchInfo * info)
{
  MatchArgs args = args_in;
  if (args.n_beams == 0) return hipErrorInvalidValue;
  args.next_item = reinterpret_cast<uint32_t *>(workspace);
  double * const staged = workspace + kCounterDoubles;
  workspace += kWorkspaceHead;
  const DeviceLimits lim = device_limits();
  const int force_grid = force_variant & kVariantGridMask;
  const bool pow2 = args.grid.pow2 != 0;

  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  bool use_lds = grid_bytes <= lim.lds_per_block;
  if (force_grid == kVariantGlobal) use_lds = false;
  if (force_grid == kVariantLds && !use_lds) return hipErrorInvalidValue;

  const Mapping mapping = choose_mapping(args, outer != nullptr, force_variant, lim);
  if (mapping == Mapping::kInvalid) return hipErrorInvalidValue;
  const bool use_small = mapping == Mapping::kSmall;
  const bool use_lane = mapping == Mapping::kLane;

  hipError_t e;
  uint32_t n_workers = 0;
  int lane_records_mode = 1;
  if (ev_main_start != nullptr && !use_lane)
  {
    e = hipEventRecord(ev_main_start, stream);
    if (e != hipSuccess) return e;
  }
  if (use_small)
  {
    // search and final reduction in one launch; the records' `done` words sit behind the
    // largest record table the small-lattice search can have (zero when the workspace is
    // allocated; a launch stores its own sequence number there)
    e = launch_match_small(args, workspace,
                           reinterpret_cast<unsigned long long *>(workspace + kSmallMaxItems * kRecord), lim.cus,
                           lim.lds_per_block, (force_variant & kVariantNoSkip) != 0, record_out,
                           record_out2, host_record, seq, stream);
    if (e != hipSuccess) return e;
    if (ev_main_done != nullptr)
    {
      e = hipEventRecord(ev_main_done, stream);
      if (e != hipSuccess) return e;
    }
    if (info != nullptr)
    {
      info->variant = pow2 ? "match/lane-per-candidate/small-lattice/pow2"
                           : "match/lane-per-candidate/small-lattice/div";
      info->n_kernels = 1;
    }
    return hipSuccess;
  }
  uint32_t n_staged = 0;   // records the slabs of a multi-slab lane search left in `staged`
  uint32_t slabs_run = 1;
  uint32_t lane_parts = 1;
  if (use_lane)
  {
    uint32_t slab_th = args.th_end - args.th_begin, n_slabs = 1;
    if (!lane_slabs(args, &slab_th, &n_slabs)) return hipErrorInvalidValue;
    if (n_slabs == 1)
    {
      e = launch_match_lane(args, outer, workspace, kMaxMatchBlocks * kMatchWaves, lim.cus,
                            lim.lds_per_block, (force_variant & kVariantNoSkip) != 0, stream,
                            ev_main_start, &n_workers, &lane_records_mode, &lane_parts);
      if (e != hipSuccess) return e;
    }
    else
    {
      // Slab after slab on the stream: table pre-kernel, search, first reduction stage into
      // the slab's own 256 records -- the per-item records of a slab are dead before the
      // next slab's search overwrites them.  Records carry whole-lattice flat indices, so
      // the final stage below treats the slabs' records like any others.
      const uint32_t th_total = args.th_end - args.th_begin;
      for (uint32_t s = 0; s < n_slabs; ++s)
      {
        const uint32_t t0 = s * slab_th;
        const uint32_t cnt = min(slab_th, th_total - t0);
        MatchArgs sub = args;
        sub.th_begin = args.th_begin + t0 * args.th_stride;
        sub.th_end = sub.th_begin + cnt;
        if (args.scores != nullptr) sub.scores = args.scores + static_cast<size_t>(t0) * args.n_lin * args.n_lin;
        uint32_t n_slab_records = 0;
        e = launch_match_lane(sub, outer, workspace, kMaxMatchBlocks * kMatchWaves, lim.cus,
                              lim.lds_per_block, (force_variant & kVariantNoSkip) != 0, stream,
                              s == 0 ? ev_main_start : nullptr, &n_slab_records, &lane_records_mode,
                              &lane_parts);
        if (e != hipSuccess) return e;
        if (s + 1 == n_slabs && ev_main_done != nullptr)
        {
          e = hipEventRecord(ev_main_done, stream);
          if (e != hipSuccess) return e;
          ev_main_done = nullptr;
        }
        const uint32_t per_block = (n_slab_records + 255) / 256;
        const uint32_t stage_blocks = (n_slab_records + per_block - 1) / per_block;
        hipLaunchKernelGGL(match_reduce_kernel, dim3(stage_blocks), dim3(256), 0, stream, workspace,
                           n_slab_records, per_block, staged + static_cast<size_t>(n_staged) * kRecord,
                           static_cast<double *>(nullptr), 0);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        n_staged += stage_blocks;
      }
      slabs_run = n_slabs;
    }
  }
  else
  {
    // Blocks: persistent, one 1024-thread block per CU (one LDS grid copy per CU).
    uint32_t max_blocks = static_cast<uint32_t>(lim.cus);
    if (max_blocks > kMaxMatchBlocks) max_blocks = kMaxMatchBlocks;

    const uint64_t m = static_cast<uint64_t>(args.n_lin) * args.n_lin;
    const uint64_t total = m * (args.th_end - args.th_begin);
    // Work item size: aim for >= 8 items per wave, between 8 and 128 candidates.
    uint64_t chunk = total / (static_cast<uint64_t>(max_blocks) * kMatchWaves * 8);
    if (chunk < 8) chunk = 8;
    if (chunk > 128) chunk = 128;
    if (chunk > m) chunk = m;
    args.chunk = static_cast<uint32_t>(chunk);
    const uint64_t cps = (m + chunk - 1) / chunk;
    const uint64_t n_items = cps * (args.th_end - args.th_begin);
    uint32_t blocks = static_cast<uint32_t>((n_items + kMatchWaves - 1) / kMatchWaves);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    args.partials = workspace;
    n_workers = blocks * kMatchWaves;

    if (use_lds)
    {
      e = pow2 ? dispatch_match_nbl<true, true>(args, blocks, grid_bytes, stream)
               : dispatch_match_nbl<true, false>(args, blocks, grid_bytes, stream);
    }
    else
    {
      e = pow2 ? dispatch_match_nbl<false, true>(args, blocks, 0, stream)
               : dispatch_match_nbl<false, false>(args, blocks, 0, stream);
    }
    if (e != hipSuccess) return e;
  }
  if (ev_main_done != nullptr)
  {
    e = hipEventRecord(ev_main_done, stream);
    if (e != hipSuccess) return e;
  }

  // Up to kFinalAloneMax records: one block.  More (the lane mapping leaves one per work item):
  // 256 blocks first, each over a contiguous share.  (The one block reads a record's words from
  // 64 different cache lines per load instruction, one line per cycle on its CU: 31 us for the
  // 6,760 records of a mid-size lattice, experiments/pmc_midsize.sh -- against 5 + 4 us in two stages.)
  const double * records = workspace;
  bool staged_here = false;
  if (n_staged > 0)
  {
    records = staged;
    n_workers = n_staged;
  }
  else if (n_workers > kFinalAloneMax)
  {
    const uint32_t per_block = (n_workers + 255) / 256;
    const uint32_t stage_blocks = (n_workers + per_block - 1) / per_block;
    hipLaunchKernelGGL(match_reduce_kernel, dim3(stage_blocks), dim3(256), 0, stream, records,
                       n_workers, per_block, staged, static_cast<double *>(nullptr), 0);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    records = staged;
    n_workers = stage_blocks;
    staged_here = true;
  }
  hipLaunchKernelGGL(match_reduce_final_kernel, dim3(1), dim3(kFinalThreads), 0, stream, records,
                     n_workers, record_out, record_out2, host_record, seq);
  e = hipGetLastError();
  if (info != nullptr)
  {
    if (use_lane)
    {
      info->variant = lane_variant_name(lane_records_mode & 3, (lane_records_mode & 4) != 0, pow2, lane_parts > 1);
      // per slab: table pre-kernel, search, (combine,) first reduction stage; then the final stage
      info->n_kernels = n_staged > 0 ? static_cast<int>((lane_parts > 1 ? 4 : 3) * slabs_run + 1)
                                     : (lane_parts > 1 ? 4 : 3) + (staged_here ? 1 : 0);
    }
    else
    {
      info->variant = use_lds ? (pow2 ? "match/wave-per-candidate/lds-grid/pow2"
                                      : "match/wave-per-candidate/lds-grid/div")
                              : (pow2 ? "match/wave-per-candidate/global-grid/pow2"
                                      : "match/wave-per-candidate/global-grid/div");
      info->n_kernels = 2;
    }
  }
  return e;
}

namespace
{

// ParticleFilter::updateStatistics on the device (reference
// src/particle_filter.cpp:163-218) from the (all-reduced) moment sums
//   stats = {sum w, sum w x, sum w y, sum w cos, sum w sin, sum w xx, sum w xy, sum w yy}:
// every weight is normalised in place (:171-174) and the weighted squared angular
// distance to the circular mean (:213-217) is reduced per block.
// totals.by_value: the moment sums arrive as kernel arguments (the sharded particle path sums
// the devices' rows on the host and launches with the totals -- no upload) instead of `stats`.
__global__ void __launch_bounds__(256) pf_finalize_kernel(const double * poses_xyt, uint64_t n,
                                                          double * weights, const double * stats,
                                                          const PoseTotals totals, double * partials)
{
  __shared__ double sh[4];
  // (read member by member: a pointer to the argument would put the struct in scratch memory)
  const double sum_w = totals.by_value ? totals.v[0] : stats[0];
  const double sum_c = totals.by_value ? totals.v[3] : stats[3];
  const double sum_s = totals.by_value ? totals.v[4] : stats[4];
  const double mean_th = atan2(sum_s / sum_w, sum_c / sum_w);
  double acc = 0.0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256)
  {
    const double w = weights[i] / sum_w;
    weights[i] = w;
    const double d = shortest_angular_distance(poses_xyt[3 * i + 2], mean_th);
    acc += w * d * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// out = {sum w, mean x, mean y, mean theta, cov xx, cov xy, cov yy, theta variance increment}
// host_out (optional): the eight results also go to host-coherent memory (system-scope stores;
// the caller synchronises the stream behind its weight copy, so no flag follows).
__global__ void __launch_bounds__(256) pf_finalize_reduce_kernel(const double * partials,
                                                                 uint32_t n_blocks,
                                                                 const double * stats, const PoseTotals totals,
                                                                 double * out, double * host_out)
{
  __shared__ double sh[256];
  double v = 0.0;
  for (uint32_t b = threadIdx.x; b < n_blocks; b += 256) v += partials[b];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1)
  {
    if (static_cast<int>(threadIdx.x) < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    double st[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = totals.by_value ? totals.v[k] : stats[k];
    const double sum_w = st[0];
    const double mean_x = st[1] / sum_w, mean_y = st[2] / sum_w;
    out[0] = sum_w;
    out[1] = mean_x;
    out[2] = mean_y;
    out[3] = atan2(st[4] / sum_w, st[3] / sum_w);
    out[4] = st[5] / sum_w - mean_x * mean_x;
    out[5] = st[6] / sum_w - mean_x * mean_y;
    out[6] = st[7] / sum_w - mean_y * mean_y;
    out[7] = sh[0];
    if (host_out != nullptr)
    {
#pragma unroll
      for (int k = 0; k < 8; ++k) store_host(host_out + k, out[k]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

}  // namespace

hipError_t launch_collect_near(const double * scores, uint64_t n, uint64_t hi, const double * record, double rel,
                               double abs_tol, unsigned long long * out, uint32_t cap, hipStream_t stream)
{
  hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  const uint64_t blocks = (n + 255) / 256;
  collect_near_kernel<<<dim3(static_cast<uint32_t>(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096)), dim3(256), 0, stream>>>(
    scores, n, hi, record, rel, abs_tol, out, cap);
  return hipGetLastError();
}

hipError_t launch_pf_finalize(const double * poses_xyt, uint64_t n_poses, double * weights,
                              const double * stats, const double * totals_by_value, double * workspace,
                              double * out, double * host_out, hipStream_t stream)
{
  if (n_poses == 0 || (stats == nullptr && totals_by_value == nullptr)) return hipErrorInvalidValue;
  PoseTotals totals{};
  if (totals_by_value != nullptr)
  {
    for (int k = 0; k < 8; ++k) totals.v[k] = totals_by_value[k];
    totals.by_value = 1;
  }
  uint64_t need = (n_poses + 255) / 256;
  const uint32_t blocks = static_cast<uint32_t>(need < kMaxPosesBlocks ? need : kMaxPosesBlocks);
  hipLaunchKernelGGL(pf_finalize_kernel, dim3(blocks), dim3(256), 0, stream, poses_xyt, n_poses,
                     weights, stats, totals, workspace);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(pf_finalize_reduce_kernel, dim3(1), dim3(256), 0, stream, workspace, blocks,
                     stats, totals, out, host_out);
  return hipGetLastError();
}

hipError_t launch_sum_moment_rows(const double * rows, uint32_t n_rows, double * out, hipStream_t stream)
{
  hipLaunchKernelGGL(sum_moment_rows_kernel, dim3(1), dim3(64), 0, stream, rows, n_rows, out);
  return hipGetLastError();
}

size_t poses_workspace_doubles(uint64_t)
{
  return static_cast<size_t>(kMaxPosesBlocks) * 8;
}

namespace
{

template <int THREADS, bool LDS_GRID, bool POW2>
hipError_t launch_poses_variant(const PosesArgs & args, uint32_t blocks, size_t lds_bytes,
                                hipStream_t stream)
{
  auto kernel = score_poses_kernel<THREADS, LDS_GRID, POW2>;
  if (lds_bytes > 48 * 1024)
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), lds_bytes, stream, args);
  return hipGetLastError();
}

}  // namespace

size_t poses_lds_per_block() { return device_limits().lds_per_block; }

hipError_t launch_score_poses(const PosesArgs & args_in, double * workspace, double * stats_out,
                              int force_variant, hipStream_t stream, hipEvent_t ev_main_done,
                              LaunchInfo * info, double * host_sums, unsigned long long host_seq)
{
  PosesArgs args = args_in;
  if (args.n_beams == 0 || args.n_poses == 0) return hipErrorInvalidValue;
  const DeviceLimits lim = device_limits();
  // beams + the per-wave statistics slots (64 B per wave, 16 waves at most)
  const size_t beams_bytes =
    static_cast<size_t>(2) * ((args.n_beams + 1) & ~1u) * sizeof(double) + 16 * 8 * sizeof(double);
  const size_t grid_bytes = static_cast<size_t>(args.grid.ncell + 1) * kCellDoubles * sizeof(double);
  bool use_lds = beams_bytes + grid_bytes <= lim.lds_per_block;
  if ((force_variant & kVariantGridMask) == kVariantGlobal) use_lds = false;
  if ((force_variant & kVariantGridMask) == kVariantLds && !use_lds) return hipErrorInvalidValue;
  if (beams_bytes > lim.lds_per_block) return hipErrorInvalidValue;

  args.partials = stats_out != nullptr ? workspace : nullptr;
  const bool pow2 = args.grid.pow2 != 0;
  hipError_t e;
  uint32_t blocks;
  // Default: per-wave compaction of the occupied (pose, beam) pairs; the dense
  // kernels remain for grids whose occupancy bitmap does not fit in LDS and for
  // A/B measurement ("dense", "lds", "global").
  const bool use_compact = (force_variant == kVariantAuto || force_variant == kVariantNoSkip) &&
                           poses_compact_supported(args, lim.lds_per_block);
  if (use_compact)
  {
    e = launch_poses_compact(args, lim.cus, force_variant == kVariantAuto, stream, &blocks);
  }
  else if (use_lds)

  {
    constexpr int T = 1024;  // one block per CU owns the LDS grid copy: make it 16 waves
    uint64_t need = (args.n_poses + T - 1) / T;
    blocks = static_cast<uint32_t>(need < static_cast<uint64_t>(lim.cus) ? need : lim.cus);
    const size_t lds_bytes = beams_bytes + grid_bytes;
    e = pow2 ? launch_poses_variant<T, true, true>(args, blocks, lds_bytes, stream)
             : launch_poses_variant<T, true, false>(args, blocks, lds_bytes, stream);
  }
  else
  {
    constexpr int T = 256;
    uint64_t need = (args.n_poses + T - 1) / T;
    blocks = static_cast<uint32_t>(need < kMaxPosesBlocks ? need : kMaxPosesBlocks);
    e = pow2 ? launch_poses_variant<T, false, true>(args, blocks, beams_bytes, stream)
             : launch_poses_variant<T, false, false>(args, blocks, beams_bytes, stream);
  }
  if (e != hipSuccess) return e;
  if (ev_main_done != nullptr)
  {
    e = hipEventRecord(ev_main_done, stream);
    if (e != hipSuccess) return e;
  }
  int n_kernels = 1;
  if (stats_out != nullptr)
  {
    hipLaunchKernelGGL(poses_reduce_kernel, dim3(1), dim3(256), 0, stream, workspace, blocks,
                       stats_out, host_sums, host_seq);
    e = hipGetLastError();
    n_kernels = 2;
  }
  if (info != nullptr)
  {
    info->variant = use_compact ? (args.coarse_log2 > 0
                                     ? (pow2 ? "poses/lane-per-pose/compact/coarse-bitmap/pow2"
                                             : "poses/lane-per-pose/compact/coarse-bitmap/div")
                                     : (pow2 ? "poses/lane-per-pose/compact/pow2"
                                             : "poses/lane-per-pose/compact/div"))
                    : use_lds   ? (pow2 ? "poses/lane-per-pose/lds-grid/pow2"
                                        : "poses/lane-per-pose/lds-grid/div")
                                : (pow2 ? "poses/lane-per-pose/global-grid/pow2"
                                        : "poses/lane-per-pose/global-grid/div");
    info->n_kernels = n_kernels;
  }
  return e;
}

}  // namespace ndt2d
