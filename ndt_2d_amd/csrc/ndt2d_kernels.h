// Internal interface between the HIP kernels (ndt2d_kernels.hip) and the
// device-layer C-ABI (ndt2d_device.hip).  Not installed; the public boundary is
// include/ndt2d_hip.h.
#ifndef NDT2D_KERNELS_H_
#define NDT2D_KERNELS_H_

#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

// cos(t) and sin(t) of one argument on the HOST, as a GCC-built reference gets them:
// wherever the reference writes the pair (`cos(pose.theta)`, `sin(pose.theta)`, ...)
// GCC merges the two calls into one glibc sincos(), whose sine can differ from sin()'s
// in the last ulp (t = 0.4710119964311561).  Host code of this library therefore asks
// for the pair explicitly instead of leaving it to the compiler that builds it.
inline void ndt2d_cos_sin(double t, double * c, double * s)
{
  sincos(t, s, c);
}

namespace ndt2d
{

// Doubles per packed device cell record.  {mean_x, mean_y, h00, h01, h11, occ}
// with h = -0.5 * information (exact scaling) and occ = 1.0 for cells that can
// score (n >= 5, reference src/ndt_model.cpp:107), 0.0 otherwise.  Cells that
// cannot score carry a sentinel mean so their exponent is -inf and exp() gives
// the reference's exact 0.0 without a branch.  Record ncell is such a sentinel
// and stands for "outside the grid" (src/ndt_model.cpp:165-169,205-208,212-215).
constexpr int kCellDoubles = 6;
// Stride (doubles) of the HBM-resident copy used when the grid does not fit in
// LDS: one 64-byte line per cell, so a gather touches exactly one line.
constexpr int kCellStrideGlobal = 8;

struct GridDesc
{
  const double * cells_lds_image;  // [ncell + 1][kCellDoubles], source of the LDS fill
  const double * cells_global;     // [ncell + 1][kCellStrideGlobal]
  const uint32_t * occ_bits;       // bit i = cell i holds a distribution; ncell + 1 bits (last = 0)
  // [size_y + 2][size_x + 2] occupancy-map bytes of the grid extended by one cell on
  // every side (ndt2d_build.hip, cell_bytes_kernel); may be null (no small-lattice search)
  const uint8_t * cell_bytes;
  // Compacted records of the cells that can score, for small maps installed from the host
  // (ndt2d_set_grid): compact_records[n_occ + 1][kCellDoubles] (the last one the
  // sentinel) and cell_rank[ncell + 1] = a cell's record in it (n_occ for cells that
  // cannot score and for "outside").  A few KB that every block of the small-lattice
  // search keeps in LDS.  n_occ == 0: not available.
  const double * compact_records;
  const uint16_t * cell_rank;
  uint32_t n_occ;
  // Map bytes at one byte per block of 2^block_bytes_log2 x 2^block_bytes_log2 cells, blocks
  // aligned to the grid's origin, extended by one block on every side:
  // [ceil(size_y / B) + 2][ceil(size_x / B) + 2] (grid_block_bytes_launch; the small-lattice
  // search on windows wider than 256 cells).  Null / 0: not made.
  const uint8_t * block_bytes;
  uint32_t block_bytes_log2;
  uint32_t size_x, size_y, ncell;
  double cell_size;
  double inv_cell_size;  // exact iff pow2
  int pow2;              // cell_size is a power of two: t * inv == t / cell_size bit-for-bit
  double origin_x, origin_y;
};

struct MatchArgs
{
  GridDesc grid;
  const double * beams_xy;  // [n_beams][2] robot frame
  uint32_t n_beams;
  const double * dth;       // [n_th]
  const double * cos_th;    // [n_th]
  const double * sin_th;    // [n_th]
  const double * dlin;      // [n_lin]
  double dlin_absmax;       // max |dlin[i]| (host side copy of the search extent)
  double beam_rmax;         // max |beam| (host side copy of the scan's reach)
  // max over the 8-step patches i = 0, 8, ... of dlin[i + 7] - dlin[i] (host side); negative
  // if dlin is not ascending within a patch: the first lane of a patch is then not its corner
  double patch_span;
  uint32_t n_th, n_lin;
  // theta steps th_begin + k * th_stride, k in [0, th_end - th_begin): a contiguous slab
  // (stride 1) or one rank's share of an interleaved sharding
  uint32_t th_begin, th_end;
  uint32_t th_stride;
  double pose_x, pose_y;
  double * scores;          // optional, slab-local flat order
  // The search tables [dth | cos | sin (n_th each) | dlin (n_lin)] in HOST memory when
  // they have not been uploaded (then dth .. dlin above point at buffers the upload
  // would fill): a launch that passes them as kernel arguments needs no copy.
  const double * host_tables;
  // Lane search of a mid-size lattice: a candidate's beams are cut into beam_parts runs of
  // part_beams beams (a multiple of 64), each (theta, patch, part) a work item of its own
  // that leaves the 64 lanes' partial sums in part_sums[(item * beam_parts + part) * 64 + lane];
  // a second kernel adds a candidate's parts in order.  beam_parts <= 1: off.
  uint32_t beam_parts, part_beams;
  double * part_sums;
  double * partials;        // [n_workers][NDT2D_MATCH_RECORD_DOUBLES]
  uint32_t * next_item;     // work-item counters of the lane-per-candidate search (kItemShards, kItemShardStride apart)
  uint32_t chunk;           // candidates per work item
};

struct PosesArgs
{
  GridDesc grid;
  const double * beams_xy;
  uint32_t n_beams;
  const double * poses_xyt;  // [n_poses][3]
  uint64_t n_poses;
  double * scores;           // [n_poses]
  double * partials;         // [n_blocks][8]  (may be null)
  double beam_rmax;          // max |beam| (host side copy of the scan's reach)
  float screen_guard;        // set by launch_poses_compact: FP32 screening error bound, cells
  // Grids whose occupancy bitmap does not fit LDS are screened with one bit per block of
  // 2^coarse_log2 x 2^coarse_log2 cells (bit = some cell of the block holds a distribution;
  // the last bit, index coarse_sx * coarse_sy, stands for "outside" and is never set):
  // coarse_bits from poses_coarse_bits_launch, null / 0 for the cell-level bitmap grid.occ_bits.
  const uint32_t * coarse_bits;
  uint32_t coarse_log2;
};

// NDT build on the device (ndt2d_build.hip).  `grid` carries the geometry only.
struct BuildArgs
{
  GridDesc grid;
  const double * points_xy;   // [n_points][2] robot frame, scans concatenated
  uint32_t n_points;
  const double * scans;       // [n_scans][4] {pose_x, pose_y, cos(theta), sin(theta)}
  const uint32_t * offsets;   // [n_scans + 1] first point of every scan
  uint32_t n_scans;
  double * world_xy;          // [n_points][2] scratch
  uint32_t * keys_in, * keys_out, * vals_in, * vals_out;  // [n_points] scratch
  void * sort_temp;
  size_t sort_temp_bytes;
  uint32_t * seg_begin;       // [2 * (ncell + 1)] scratch: begin | end
  double * cells6;            // [ncell][6] out: {mean_x, mean_y, i00, i01, i11, n}
  double * cells_lds_image;   // [ncell + 1][kCellDoubles] out
  double * cells_global;      // [ncell + 1][kCellStrideGlobal] out
  uint32_t * occ_bits;        // out
  uint8_t * cell_bytes;       // out: [size_y + 2][size_x + 2]
  int eigen_form;             // ndt2d_eigen2.h: 0 = Eigen's EigenSolver transcribed, 1 = closed form
};
size_t build_sort_temp_bytes(uint32_t n_points, uint32_t ncell);
hipError_t launch_build_grid(const BuildArgs & args, hipStream_t stream);
// The compacted records + cell -> record table of a device-built grid (ncell < 65,535), from the
// packed records and the occupancy bitmap launch_build_grid left: records[(n_occ + 1)][6] (at most
// ncell + 1 rows), ranks[ncell + 1], *n_occ_out = the cells that can score.
hipError_t launch_compact_grid(uint32_t ncell, const double * cells_lds_image, const uint32_t * occ_bits,
                               double * records, uint16_t * ranks, uint32_t * n_occ_out, hipStream_t stream);
// The scorer layouts of a grid uploaded as cells6 records (ndt2d_set_grid): packed
// records in both strides, occupancy bitmap, per-cell map bytes.  `geometry` carries
// size / cell size / origin only.
hipError_t launch_pack_grid(const GridDesc & geometry, const double * cells6,
                            double * cells_lds_image, double * cells_global, uint32_t * occ_bits,
                            uint8_t * cell_bytes, hipStream_t stream);
// A grid given as the list of its n cells that hold points (cells6[k] belongs to cell
// cell_index[k]; all others are empty; rank_of[k]: the listed cell's record among the compacted
// ones, the cell -> record table `ranks` is optional), as the host staged it:
// [cells6: 6 n doubles | cell indices: n u32 | ranks of the listed cells: n u16 | compacted
// records | occupancy words of the cells that can score], offsets in doubles.
// src: the image as the kernel addresses it -- the pinned staging buffer in place, or
// its device copy (then dst == nullptr); dst: where to leave the device copy.  The map bytes
// are left zeroed: launch_sparse_bytes.
struct SparseImage
{
  const double * src;
  double * dst;
  uint32_t n;            // listed cells
  size_t off_idx, off_rk, off_compact, n_compact, off_occ;
};
hipError_t launch_grid_install(const GridDesc & geometry, const SparseImage & image, double * cells_lds_image,
                               double * cells_global, uint32_t * occ_bits, uint8_t * cell_bytes,
                               uint16_t * ranks, uint32_t n_occ, hipStream_t stream);
// What is left of a list install afterwards: the map bytes around the n listed cells (device
// copies of the list; the geometry's cells_lds_image / occ_bits as the install left them).
// Only the small-lattice search reads the bytes, so the job need not run before a scoreScan:
// it rides along in that launch's spare blocks (FewOut::side) or runs ahead of the search.
struct SparseBytesJob
{
  const uint32_t * cell_index;
  const double * cells6;
  uint32_t n;
  uint8_t * bytes;
};
inline uint32_t sparse_bytes_blocks(const SparseBytesJob & job)
{
  return static_cast<uint32_t>((static_cast<uint64_t>(job.n) * 9 + 15) / 16);   // 16 lanes per (cell, neighbour)
}
hipError_t launch_sparse_bytes(const GridDesc & geometry, const SparseBytesJob & job, hipStream_t stream);
hipError_t launch_grid_sparse_to_dense(const uint32_t * cell_index, const double * cells6, uint32_t n,
                                       uint32_t ncell, double * dense6, hipStream_t stream);
hipError_t launch_grid_tail(const GridDesc & geometry, const double * cells_lds_image,
                            uint32_t * occ_bits, uint8_t * cell_bytes, hipStream_t stream);

struct LaunchInfo
{
  const char * variant;
  int n_kernels;
};

// Launch the match search + its final reduction.  record_out: device,
// 12 doubles.  workspace: device scratch for per-wave partials, at least
// match_workspace_doubles() doubles.
size_t match_workspace_doubles(const MatchArgs & args);
size_t match_workspace_head_doubles();   // counters + first-stage records in front of the record table
// ev_main_done (optional) is recorded right after the search kernel, before the
// tiny final reduction, so the caller can time the dominant kernel alone.
// outer (optional): scratch of match_lane_outer_doubles() doubles; without it the
// lane-per-candidate mapping is not available.
// host_record (optional, device-visible host-coherent memory): receives the record and
// then, at host_record[kHostFlagSlot], `seq` as a 64-bit integer -- a host may spin on
// it instead of synchronising the stream.
constexpr int kHostFlagSlot = 16;
// ... or seq | kHostFlagGaveUp: the launch's last block waited in vain for another block's
// result (its bounded poll, ndt2d_device_fn.h); the call has no result (NDT2D_ERR_HIP).
constexpr unsigned long long kHostFlagGaveUp = 1ull << 63;
// Whether launch_match with these arguments reads the search tables from device memory
// (false: the small-lattice search takes them as kernel arguments from args.host_tables).
bool match_needs_device_tables(const MatchArgs & args, bool outer_available, int force_variant);
// doubles of search tables that fit the small-lattice search's kernel arguments
constexpr uint32_t kArgTableDoubles = 416;
// ev_main_start / ev_main_done (optional): recorded right before and right after the search
// kernel itself (behind the lane mapping's table pre-kernel, before the reductions).
hipError_t launch_match(const MatchArgs & args, double * workspace, double * outer,
                        double * record_out, double * record_out2, double * host_record,
                        unsigned long long seq, int force_variant, hipStream_t stream,
                        hipEvent_t ev_main_start, hipEvent_t ev_main_done, LaunchInfo * info);

size_t poses_workspace_doubles(uint64_t n_poses);
// The candidates (slab-local index < hi) that scored below 0 and within rel * |best| + abs_tol of
// the search's best: scores[n] (slab-local order), record = the search's result record (record[0] =
// best score).  out[0] = how many there are, out[1 .. cap] = their slab-local indices (in no
// particular order; only the first cap to arrive are kept).
hipError_t launch_collect_near(const double * scores, uint64_t n, uint64_t hi, const double * record, double rel,
                               double abs_tol, unsigned long long * out, uint32_t cap, hipStream_t stream);
// host_sums (optional, with stats_out): host-coherent memory as the device addresses it; the
// reduction stores the eight sums there too and then `host_seq` at
// host_sums[kPoseSumsFlagOffset] for a host thread that spins on it.
constexpr int kPoseSumsFlagOffset = 8;
hipError_t launch_score_poses(const PosesArgs & args, double * workspace, double * stats_out,
                              int force_variant, hipStream_t stream, hipEvent_t ev_main_done,
                              LaunchInfo * info, double * host_sums = nullptr,
                              unsigned long long host_seq = 0);

// out[k] = ((rows[0][k] + rows[1][k]) + rows[2][k]) + ... for the 8 moment sums of n_rows pieces of
// a particle set scored piece by piece (the pipelined pose batches of ndt2d_device.hip).
hipError_t launch_sum_moment_rows(const double * rows, uint32_t n_rows, double * out, hipStream_t stream);

// The moment sums of updateStatistics as a kernel argument (by_value != 0) -- see launch_pf_finalize.
struct PoseTotals
{
  double v[8];
  int by_value;
};

// ParticleFilter::updateStatistics from the moment sums: normalises weights in
// place and writes out = {sum w, mean x, mean y, mean theta, cov xx, cov xy, cov yy,
// theta-variance increment}.  workspace: poses_workspace_doubles() doubles.
// The sums come from device memory (`stats`) or, totals_by_value != nullptr, as eight host
// values that travel as kernel arguments.  host_out (optional): `out` in host-coherent memory too.
hipError_t launch_pf_finalize(const double * poses_xyt, uint64_t n_poses, double * weights,
                              const double * stats, const double * totals_by_value, double * workspace,
                              double * out, double * host_out, hipStream_t stream);

// Particle-filter steps around `measure` (ndt2d_motion.hip).  The float fields are
// the parameters std::normal_distribution<float>(mean, sigma) holds
// (reference src/motion_model.cpp:70-72, src/particle_filter.cpp:56-58).
struct MotionParams
{
  float rot1, trans, rot2;
  float sigma_rot1, sigma_trans, sigma_rot2;
};
struct InitParams
{
  float x, y, theta;
  float sigma_x, sigma_y, sigma_theta;
};
// Grid-stride cap of the streaming particle kernels; poses_workspace_doubles()
// holds 8 doubles for each of these blocks.
constexpr uint32_t kMaxStreamBlocks = 4096;
// noise: [n][3] standard normals (float) or null = the Philox4x32-10 stream of
// (seed, step), particle i using counter first_index + i.
hipError_t launch_pf_noise(float * noise_out, uint64_t n, uint64_t seed, uint64_t first_index,
                           uint64_t step, hipStream_t stream);
hipError_t launch_pf_motion(double * poses_xyt, uint64_t n, const MotionParams & params,
                            const float * noise, uint64_t seed, uint64_t first_index,
                            uint64_t step, hipStream_t stream);
hipError_t launch_pf_init(double * poses_xyt, uint64_t n, const InitParams & params,
                          const float * noise, uint64_t seed, uint64_t first_index,
                          uint64_t step, hipStream_t stream);
// stats_out[8] = sums {w, w x, w y, w cos, w sin, w xx, w xy, w yy}; weights null =
// uniform 1/n.  workspace: poses_workspace_doubles() doubles.
hipError_t launch_pose_moments(const double * poses_xyt, uint64_t n, const double * weights,
                               double * workspace, double * stats_out, hipStream_t stream);

// LaserScan -> Scan conversion + beam subsampling (ndt2d_scan.hip).  cos_lt / sin_lt
// are the host libm values of reference src/ndt_mapper.cpp:403-404.
struct ScanDesc
{
  float angle_min, angle_increment;
  double range_max;
  int inverted;
  double laser_x, laser_y, cos_lt, sin_lt;
  double motion_x, motion_y, motion_theta;
};
// points_xy: [n_ranges][2]; info_out[2] = {points kept, upper bound of max |point|}
hipError_t launch_convert_scan(const float * ranges, uint32_t n_ranges, const ScanDesc & desc,
                               double * points_xy, double * info_out, hipStream_t stream);
// beams_xy: [max_beams][2]; scan_info = launch_convert_scan's record;
// info_out[3] = {points, beams used, upper bound of max |beam|}
hipError_t launch_subsample(const double * points_xy, const double * scan_info,
                            uint32_t max_beams, double * beams_xy, double * info_out,
                            hipStream_t stream);

// OccupancyGrid rendering (ndt2d_occupancy.hip).  Scans as in BuildArgs.
struct OccupancyArgs
{
  const double * points_xy;   // [n_points][2] robot frame, scans concatenated
  uint32_t n_points;
  const double * scans;       // [n_scans][4] {pose_x, pose_y, cos(theta), sin(theta)}
  const uint32_t * offsets;   // [n_scans + 1]
  uint32_t n_scans;
  double resolution, origin_x, origin_y;
  uint32_t width, height;
};
// bounds_out[4] = {min_x, max_x, min_y, max_y} of the map-frame points
// [first_point, n_points) (+inf / -inf if none); workspace: poses_workspace_doubles().
hipError_t launch_occupancy_bounds(const OccupancyArgs & args, uint32_t first_point,
                                   double * workspace, double * bounds_out, hipStream_t stream);
// counts: [width * height] scratch; data: [width * height] out, -1 / 0 / 100.
hipError_t launch_occupancy_render(const OccupancyArgs & args, double occ_thresh,
                                   unsigned long long * counts, signed char * data,
                                   hipStream_t stream);

// Lane-per-candidate search (ndt2d_match_lane.hip).  outer: device scratch of
// match_lane_outer_doubles() doubles for the rotated-beam table; workspace
// receives one partial record per wave (*n_workers_out of them).
size_t match_lane_outer_doubles(const MatchArgs & args);
bool match_lane_supported(const MatchArgs & args, size_t lds_per_block);
// How a launch's theta steps are cut into slabs of at most kLaneSlabItems work items (see
// there): *slab_th theta steps per slab, *n_slabs slabs; false if no cut within the limits
// exists.  Most searches are one slab.
bool lane_slabs(const MatchArgs & args, uint32_t * slab_th, uint32_t * n_slabs);
// no_skip: every beam of every candidate takes the exact path (no occupancy / bound
// / negligible-term skipping) -- the bit-exactness control for the skipping logic.
hipError_t launch_match_lane(const MatchArgs & args, double * outer, double * workspace,
                             uint32_t max_workers, int cus, size_t lds_per_block, bool no_skip,
                             hipStream_t stream, hipEvent_t ev_after_pre_kernel, uint32_t * n_workers_out,
                             int * records_mode_out, uint32_t * parts_out = nullptr);

// Small-lattice search (ndt2d_match_small.hip): a block per (theta, up to P tiles of 64
// candidates), its waves split the beams; needs grid.cell_bytes.  The launch includes the
// final reduction: workspace takes one record per (theta, tile), `done` one word per record
// (kSmallMaxItems of them, zero when allocated); the launch's extra, last block writes the
// result record(s) as launch_match does.
bool match_small_supported(const MatchArgs & args, size_t lds_per_block);
// The block size (log2) of the map the small-lattice search of `args` needs: 0 = the
// per-cell bytes, k > 0 = GridDesc::block_bytes at 2^k cells per byte (to be made with
// grid_block_bytes_launch before the launch), -1 = the small-lattice search cannot serve it.
int match_small_block_log2(const MatchArgs & args, size_t lds_per_block);
size_t grid_block_bytes_size(const GridDesc & g, uint32_t block_log2);
hipError_t grid_block_bytes_launch(const GridDesc & g, uint32_t block_log2, uint8_t * out, hipStream_t stream);
bool match_small_takes_arg_tables(const MatchArgs & args);
hipError_t launch_match_small(const MatchArgs & args, double * workspace, unsigned long long * done,
                              int cus, size_t lds_per_block, bool no_skip, double * record_out,
                              double * record_out2, double * host_record, unsigned long long seq,
                              hipStream_t stream);

// Particle scoring with per-wave compaction of the occupied (pose, beam) pairs
// (ndt2d_poses_compact.hip).
bool poses_compact_supported(const PosesArgs & args, size_t lds_per_block);
// Smallest coarse_log2 (0 = the cell-level bitmap) with which the kernel's LDS image fits;
// -1 if none up to 4 does.
int poses_coarse_log2(const PosesArgs & args, size_t lds_per_block);
size_t poses_coarse_words(const GridDesc & g, uint32_t coarse_log2);
size_t poses_lds_per_block();   // the LDS a block of the particle kernels may use on this device
hipError_t poses_coarse_bits_launch(const GridDesc & g, uint32_t coarse_log2, uint32_t * out, hipStream_t stream);
// screen: phase A decides in FP32 which pairs can contribute (exact FP64 follows for
// those); false = the exact phase A ("compact-exact", the control).
hipError_t launch_poses_compact(const PosesArgs & args, int cus, bool screen, hipStream_t stream,
                                uint32_t * blocks_out);

// Scoring of a small batch of poses, a block per pose and a thread per beam
// (ndt2d_poses_compact.hip, score_few_kernel); bit-identical to the batched kernel.
constexpr uint32_t kFewPoses = 8;        // poses that travel as kernel arguments
constexpr uint32_t kFewPosesMax = 2048;  // poses this kernel is used for
constexpr int kFewThreads = 256;
struct FewPoses
{
  double xyt[3 * kFewPoses];
};
constexpr uint32_t kArgBeams = 208;      // beams that travel as kernel arguments (3.3 KB)
struct FewBeams
{
  double xy[2 * kArgBeams];
};
// Where a score_few launch leaves its results.  args.scores: the scores -- or, with
// `stats`, the normalised weights -- device memory when flag == nullptr, else
// host-coherent memory with `seq` raised at *flag once everything is there.
struct FewOut
{
  unsigned long long * flag;
  unsigned long long seq;
  // one word per pose (kFewPosesMax of them, zero when allocated; needed with flag != nullptr):
  // a pose's block stores the launch's `seq` there once its score is in place, and the
  // launch's extra last block polls them (instead of a ticket that every block would draw)
  unsigned long long * done;
  double * beams_out;           // device: the beams, when they came as kernel arguments
  int stats;                    // run ParticleFilter::updateStatistics in the launch's last block
  double * dev_scores;          // device scratch [n_poses] (stats)
  double * host_out;            // host-coherent [NDT2D_PF_RESULT_DOUBLES] (stats)
  // device scratch [n_poses][3] (stats, optional): every block leaves its pose here, so that
  // the statistics pass does not read args.poses_xyt -- pinned host memory, over PCIe -- again
  double * dev_poses;
  // (optional, side.n > 0) the map bytes of a list install, computed by blocks of this launch
  // behind the poses' -- nobody waits for them here, the search that reads them is queued later
  SparseBytesJob side;
};
bool score_few_supported(const PosesArgs & args, size_t lds_per_block);
// args.poses_xyt == nullptr: the (<= kFewPoses) poses are few->xyt.  host_beams (optional,
// with out.beams_out): the args.n_beams <= kArgBeams beams in host memory; they travel as
// kernel arguments and the kernel leaves them in out.beams_out.
hipError_t launch_score_few(const PosesArgs & args, const FewPoses * few, const FewOut & out,
                            const double * host_beams, hipStream_t stream);

// force_variant: grid placement in the low bits, candidate mapping above them
enum { kVariantAuto = 0, kVariantLds = 1, kVariantGlobal = 2, kVariantGridMask = 3,
       kVariantWave = 4, kVariantLane = 8, kVariantDense = 16, kVariantNoSkip = 32,
       kVariantSmall = 64 };
// Work items of the lane-per-candidate search are handed out by kItemShards counters
// (one address takes ~88 atomics/us; a launch issues up to tens of thousands), each in
// a cache line of its own.
constexpr uint32_t kItemShards = 8;
constexpr uint32_t kItemShardStride = 64;   // in uint32: 256 bytes
// The lane-per-candidate search leaves one record per work item (theta x 8x8 patch).  A
// lattice of more than kLaneSlabItems items (192 MB of records) is searched in slabs of
// whole theta steps, launch after launch on the stream, each slab's records reduced to 256
// before the next one overwrites them; at most kMaxLaneSlabs slabs (larger ones if need
// be), none beyond kMaxLaneItems items -- what neither allows takes the wave mapping.
constexpr uint64_t kLaneSlabItems = 1ull << 21;
constexpr uint32_t kMaxLaneSlabs = 32;
constexpr uint64_t kMaxLaneItems = 1ull << 24;
// launch_match: lattices with fewer (theta, 8x8 patch) items than this take the
// small-lattice search (beams split across the waves of a block); it can hold
// kSmallMaxItems.  Above, the lane-per-candidate search; the wave-per-candidate mapping
// serves what neither can (windows beyond 1,024 cells, > 2^24 items).
// Round 5, whole matchScan calls with the event pairs off (experiments/small_crossover.py; small /
// large form, ms): 100 beams 2,704 items 0.068 / 0.088, 3,380 items 0.082 / 0.073; 360 beams 2,704
// 0.100 / 0.140, 4,056 0.136 / 0.159, 5,070 0.164 / 0.166; 720 beams 2,704 0.143 / 0.173, 3,380 0.168 /
// 0.193, 4,056 0.191 / 0.197, 5,070 0.220 / 0.209 -- the crossover lies near 3,000 items for the
// node's own 100-beam scans and past 4,000 for longer ones (2,560 for all until then: the small form
// has gained since -- centre-first blocks -- and the large form's extra launches cost every call 30 us).
constexpr uint64_t kSmallBelowItemsShortScan = 3072;   // scans of at most kShortScanBeams beams
constexpr uint64_t kSmallBelowItemsLongScan = 4096;
constexpr uint32_t kShortScanBeams = 128;
constexpr uint64_t kWaveBelowItems = 2560;             // (the wave mapping's range where the small form cannot run)
constexpr uint64_t kSmallMaxItems = 8192;
// Lane search: lattices (whole, not a launch's share) below these many work items have a
// candidate's beams cut into four / two parts (MatchArgs::beam_parts): one expensive item
// -- 720 exact evaluations in a dependent chain -- takes a wave 0.2 ms however few items
// the launch has.
// (experiments/mid_lattice_parts.py, 720 beams: 1,352 items 230 -> 93 us and 3,549 items 349 -> 155 us
// with four parts, 13,520 items 311 -> 284 us with two, 23,660 items 377 us uncut against 411 us)
constexpr uint64_t kPartsFourBelow = 10240;
constexpr uint64_t kPartsTwoBelow = 18432;

// The searches and the particle kernel address their LDS map ABSOLUTELY (lds_byte_at, ndt2d_lane_fn.h:
// the packed cell bytes are the LDS address), which is right only while the kernel's dynamic block
// starts at LDS offset 0, i.e. while the kernel has no static __shared__ of its own.  Checked on the
// host, once per kernel: hipFuncGetAttributes(...).sharedSizeBytes must be 0 (until round 6 the
// kernels checked it themselves and trapped -- which takes the whole context down with them).
// hipErrorInvalidDeviceFunction when it is not: the launch does not happen, the call returns an error.
// (also remembers the largest dynamic LDS size granted to the kernel, so that hipFuncSetAttribute
// is not called again on every launch)
hipError_t prepare_absolute_lds_kernel(const void * kernel, size_t dynamic_lds_bytes);

}  // namespace ndt2d

#endif  // NDT2D_KERNELS_H_
