/*
 * ndt2d_hip.h -- C-ABI of libndt2d_hip.so: hand-written HIP (gfx950 / MI355X)
 * kernels for ndt_2d's NDT scan-matching hot path, plus the host-side mirror of
 * the reference's ndt_2d::ScanMatcher plugin interface for that path.
 *
 * Plain C, no torch / Eigen / ROS types: pointers, sizes, doubles.  Every entry
 * point returns an int status (NDT2D_OK == 0) and never throws.  Citations are
 * file:line in the reference repository (mikeferguson/ndt_2d @ 2024-12-18).
 *
 * Two layers live in the one library:
 *
 *  (1) device layer  ndt2d_*         one opaque context per GPU and per plugin
 *      instance: resident NDT grid, resident beams, search tables, kernel
 *      launches.  This is what a cgo/JNI/ctypes/pluginlib binding calls.
 *
 *  (2) matcher layer ndt2d_matcher_* the reference's ScanMatcherNDT object
 *      (initialize / addScans / matchScan / scoreScan / scorePoints / reset,
 *      include/ndt_2d/scan_matcher.hpp:42-91) restated over layer (1), plus the
 *      additive batched particle path (ParticleFilter::measure,
 *      src/particle_filter.cpp:78-89).  The pluginlib shim
 *      (ndt_2d_amd/plugin/scan_matcher_ndt_hip.cpp) is a thin wrapper of it.
 *
 * There is NO CPU fallback: nothing in this library exists without a usable GPU
 * (ndt2d_create / ndt2d_matcher_create return NDT2D_ERR_NO_DEVICE / NDT2D_ERR_HIP),
 * and every search, every batch of poses and every scan of more than 256 beams is
 * evaluated on the GPU.  What runs on the HOST, by design and on a live matcher only
 * (DESIGN.md 3.6): ONE pose of a short scan -- ndt2d_matcher_score_points /
 * _score_scan as the unchanged ParticleFilter::measure calls them, once per particle
 * (src/particle_filter.cpp:81-87) -- is scored by the calling thread from the host
 * copy of the NDT in the reference's order (0.4 us instead of a 9 us launch + PCIe
 * round trip; ndt2d_matcher_set_single_pose_path(m, "device", 0) turns it off), and
 * the few candidates of a marked near-tie are rescored in the reference's arithmetic.
 */
#ifndef NDT2D_HIP_H_
#define NDT2D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDT2D_OK 0
#define NDT2D_ERR_INVALID 1    /* bad argument (NULL, zero size, bad range) */
#define NDT2D_ERR_NO_GRID 2    /* compute call before ndt2d_set_grid / addScans */
#define NDT2D_ERR_HIP 3        /* a HIP runtime call failed; see ndt2d_last_error */
#define NDT2D_ERR_NO_DEVICE 4  /* no GPU visible to this process */
#define NDT2D_ERR_STATE 5      /* call sequence error (e.g. fetch before launch) */
#define NDT2D_ERR_ALLOC 6      /* host memory could not be had (std::bad_alloc caught at the boundary) */
#define NDT2D_ERR_INTERNAL 7   /* any other C++ exception caught at the boundary; see ndt2d_last_error */

#define NDT2D_NO_INDEX UINT64_MAX

/* ------------------------------------------------------------------------ */
/* (1) device layer                                                         */
/* ------------------------------------------------------------------------ */

typedef struct ndt2d_context * ndt2d_handle;

/* ABI version of this header: bumped whenever an export is added, a signature changes or a
 * default changes results.  4 (round 6): + NDT2D_ERR_ALLOC / NDT2D_ERR_INTERNAL,
 * ndt2d_host_build_grid_ex, and the round-5 additions that had gone out under 3 (pose_sums_*,
 * pf_finalize_totals_launch, pipeline_pieces, multi_thresholds, eigenvalue_form, build_info).  A
 * consumer compares ndt2d_abi_version() with the NDT2D_ABI_VERSION it was compiled against
 * (the pluginlib shim does, in initialize). */
#define NDT2D_ABI_VERSION 4
int ndt2d_abi_version(void);
/* Which sources this library was compiled from: "NDT2D_SOURCE_SHA256=<64 hex digits> arch=...
 * compiler=...".  The hash is ndt_2d_amd/build.py's source_sha256() (csrc/ *.hip, *.cpp, *.h,
 * this header, the compiler flags) at build time: the tests, bench.py and tests/conftest.py
 * compare it with the tree they run from, so a stale binary cannot stand in for the sources. */
const char * ndt2d_build_info(void);

/* One context per (plugin instance, GPU).  The reference keeps all state per
 * ScanMatcherNDT instance (std::unique_ptr<NDT> ndt_,
 * include/ndt_2d/scan_matcher_ndt.hpp:102) and runs two instances concurrently
 * on two threads (src/ndt_mapper.cpp:141-142,508-515,634-643): a context owns
 * its own HIP stream and buffers and holds no process-global mutable state. */
int ndt2d_create(ndt2d_handle * out, int device_id);
int ndt2d_destroy(ndt2d_handle h);
/* Message of the last failure on this context ("" if none).  Never NULL. */
const char * ndt2d_last_error(ndt2d_handle h);
/* Launch on a caller-owned hipStream_t instead of the context's own stream
 * (NULL restores the own stream).  ndt2d_get_stream returns the active one. */
int ndt2d_set_stream(ndt2d_handle h, void * hip_stream);
void * ndt2d_get_stream(ndt2d_handle h);
int ndt2d_device_id(ndt2d_handle h);

/* Upload the NDT cell grid; replaces the NDT object built by
 * ScanMatcherNDT::addScans (src/scan_matcher_ndt.cpp:66-73).
 * cells6[i] = {mean_x, mean_y, information(0,0), information(0,1),
 * information(1,1), n} for cell i = grid_y * size_x + grid_x, i.e. the fields
 * Cell::score reads (src/ndt_model.cpp:105-116); size/origin/cell_size are
 * NDT::size_x_, size_y_, origin_x_, origin_y_, cell_size_
 * (include/ndt_2d/ndt_model.hpp:128-131). */
int ndt2d_set_grid(ndt2d_handle h, const double * cells6, uint32_t size_x, uint32_t size_y,
                   double cell_size, double origin_x, double origin_y);

/* The same grid given as the LIST of its cells that received points: cells6[k] (layout as
 * above) belongs to cell cell_index[k] = grid_y * size_x + grid_x, every cell not listed is
 * an empty one (n = 0, as NDT::NDT leaves it, src/ndt_model.cpp:118-126); a cell may be
 * listed once, in any order (a cell with n >= 5 listed twice: NDT2D_ERR_INVALID).  The mapper
 * rebuilds its local NDT for every scan (src/ndt_mapper.cpp:508-509); with a real lidar
 * that grid has tens of thousands of cells (scan poses +- range_max,
 * src/scan_matcher_ndt.cpp:52-66) of which the scans touch a thousand or two: the cost of
 * this call follows the list, not the grid. */
int ndt2d_set_grid_sparse(ndt2d_handle h, const uint32_t * cell_index, const double * cells6,
                          size_t n_listed, uint32_t size_x, uint32_t size_y, double cell_size,
                          double origin_x, double origin_y);

/* The same in two steps, for a host that can write its list straight into the library's pinned
 * staging buffer (no intermediate copy): begin hands out room for up to `capacity` listed cells
 * (cell_index_out[k], cells6_out[6 k .. 6 k + 5]), commit installs the first n_listed of them.
 * Nothing else may be called on the context in between. */
int ndt2d_grid_stage_begin(ndt2d_handle h, uint32_t size_x, uint32_t size_y, size_t capacity,
                           uint32_t ** cell_index_out, double ** cells6_out);
int ndt2d_grid_stage_commit(ndt2d_handle h, size_t n_listed, double cell_size, double origin_x,
                            double origin_y);

/* Build the NDT on the device from the scans themselves and install it: the
 * whole of ScanMatcherNDT::addScans (src/scan_matcher_ndt.cpp:49-74) --
 * bounding box of the scan poses +- range_max, NDT::addScan for every scan in
 * order (src/ndt_model.cpp:132-152), NDT::compute (:154-160).  Scan k has pose
 * poses_xyt[3k..3k+2] and robot-frame points points_xy[2*offsets[k] ..
 * 2*offsets[k+1]).  Every cell sees its points in the reference's order (stable
 * sort by cell), so the result is bit-identical to the host build.  Asynchronous
 * on the context's stream. */
int ndt2d_build_grid(ndt2d_handle h, double ndt_resolution, double range_max,
                     const double * poses_xyt, const double * points_xy, const size_t * offsets,
                     size_t n_scans);
/* How ndt2d_build_grid forms the eigenvalues of Cell::compute (src/ndt_model.cpp:84-85,
 * Eigen::EigenSolver<Eigen::Matrix2d>): "eigen" (default) = Eigen 3.4.0's RealSchur /
 * EigenSolver transcribed operation by operation for a 2 x 2 input (csrc/ndt2d_eigen2.h: the
 * input scaled by its largest entry, the Givens rotation of the 2 x 2 block, the diagonal read
 * back); "closed" = the closed form (a + d) / 2 +- sqrt(((a - d) / 2)^2 + b^2) of rounds 1-4.
 * The two differ in the last ulps; the difference reaches a cell's information matrix only
 * through the clamp branch (:88-96). */
int ndt2d_set_eigenvalue_form(ndt2d_handle h, const char * form);
/* Geometry and (optionally) the cells6 records of the installed grid, copied
 * back from the device.  Any output may be NULL. */
int ndt2d_get_grid(ndt2d_handle h, double * cells6_out, size_t capacity_cells, uint32_t * size_x,
                   uint32_t * size_y, double * cell_size, double * origin_x, double * origin_y);
/* ScanMatcherNDT::reset (src/scan_matcher_ndt.cpp:180-183). */
int ndt2d_clear_grid(ndt2d_handle h);
int ndt2d_has_grid(ndt2d_handle h);

/* Upload the beam endpoints of one scan, robot frame, already subsampled to
 * min(laser_max_beams, points.size()) points by the caller
 * (src/scan_matcher_ndt.cpp:95-96,110 / :165-166,171). */
int ndt2d_set_beams(ndt2d_handle h, const double * beams_xy, size_t n_beams);

/* Upload the search lattice of matchScan.  dth[n_th] / dlin[n_lin] are the
 * values the reference's floating-point loops visit
 * (`for (dth = -angular_size_; dth < angular_size_; dth += angular_res_)`,
 * src/scan_matcher_ndt.cpp:103,117,119); cos_th/sin_th[n_th] are
 * cos/sin(scan_pose.theta + dth[i]) evaluated by the host libm (:106-107);
 * pose_x/pose_y = scan_pose.x/.y (:112,114). */
int ndt2d_set_search(ndt2d_handle h, double pose_x, double pose_y, const double * dth,
                     const double * cos_th, const double * sin_th, size_t n_th,
                     const double * dlin, size_t n_lin);

/* ndt2d_set_beams + ndt2d_set_search in one call: both travel in ONE staged copy (a
 * small search is a few tens of microseconds; every copy command costs ~3 us).
 * beams_xy == NULL: the n_beams beams the context already holds stay (the caller knows
 * they are this scan's: ndt2d_matcher_* compares), and the tables are only staged --
 * a small-lattice search takes them as kernel arguments, so such a call copies nothing. */
int ndt2d_set_search_beams(ndt2d_handle h, const double * beams_xy, size_t n_beams, double pose_x,
                           double pose_y, const double * dth, const double * cos_th,
                           const double * sin_th, size_t n_th, const double * dlin, size_t n_lin);

/* Result of one (possibly sharded) matchScan search,
 * src/scan_matcher_ndt.cpp:103-143.  Candidate flat index =
 * (i_theta * n_lin + i_x) * n_lin + i_y, the reference's loop order. */
typedef struct ndt2d_match_result
{
  double best_score;      /* raw best score = -NDT::likelihood (:127); 0.0 if none < 0 */
  uint64_t best_index;    /* flat index of the winner, NDT2D_NO_INDEX if none (:128) */
  double acc[10];         /* k00,k01,k02,k11,k12,k22, u0,u1,u2, s  (:137-140) */
  uint64_t n_candidates;  /* candidates evaluated by this call */
  /* 1: another candidate scored within the near-tie tolerance of the winner --
   * |difference| <= NDT2D_NEAR_TIE_REL * (the larger magnitude); the
   * kernels' scores differ from the CPU reference's in the last bits (device exp vs libm: a
   * relative error below 1e-13), so the two could come out in the other order there;
   * ndt2d_match_near_best lists such candidates and ndt2d_matcher_match_scan settles them with
   * the reference's own arithmetic.  0 proves that no other candidate lies that close. */
  uint64_t near_tie;
} ndt2d_match_result;
#define NDT2D_NEAR_TIE_REL 1.4551915228366852e-11   /* 2^-36 */

/* Number of doubles of the device-resident result record
 * {best_score, best_index (exact double, -1 if none; + 0.5 = near_tie, truncate), acc[10]}. */
#define NDT2D_MATCH_RECORD_DOUBLES 12

/* Evaluate the theta slab [th_begin, th_end) of the lattice (all n_lin x n_lin
 * translations of each theta).  Asynchronous on the context's stream.
 * d_scores (device pointer, optional): receives the raw score of every
 * candidate of the slab, slab-local flat order.  d_record (device pointer,
 * optional): receives the NDT2D_MATCH_RECORD_DOUBLES-double result record
 * (for a device-side all-reduce); the context always keeps its own copy. */
int ndt2d_match_launch(ndt2d_handle h, size_t th_begin, size_t th_end, double * d_scores,
                       double * d_record);
/* The same for the theta steps th_first, th_first + th_stride, ... (th_count of
 * them): one rank's share of a lattice whose theta axis is dealt out round-robin
 * (the cost of a step varies across the angular range, so contiguous slabs leave
 * the ranks unevenly loaded).  best_index is the flat index in the WHOLE lattice
 * ((i_theta * n_lin + i_x) * n_lin + i_y, the reference's visiting order
 * src/scan_matcher_ndt.cpp:103-119), d_scores is local: step k of this call first. */
int ndt2d_match_launch_strided(ndt2d_handle h, size_t th_first, size_t th_stride,
                               size_t th_count, double * d_scores, double * d_record);
/* Wait for the last ndt2d_match_launch and return its result.  (The final reduction
 * writes the record into host-coherent pinned memory and then raises a flag there; the
 * host spins on the flag, which returns ~4 us sooner than a stream synchronisation.) */
int ndt2d_match_fetch(ndt2d_handle h, ndt2d_match_result * out);
/* Search the slab [th_begin, th_end) keeping every candidate's score on the device, and list the
 * candidates that scored below 0 and within rel * |best| of the slab's best
 * (flat indices in the whole lattice, ascending).  *n_out = how many there are; when that exceeds
 * `capacity` the list holds the first `capacity`-or-fewer of them in visiting order (further passes
 * over the scores).  result_out (optional) = the slab's result.  Synchronous.  This is the slow path
 * behind a near_tie result: one more search plus one pass over its scores. */
int ndt2d_match_near_best(ndt2d_handle h, size_t th_begin, size_t th_end, double rel, uint64_t * index_out,
                          size_t capacity, size_t * n_out, ndt2d_match_result * result_out);
/* Searches launched and results fetched on this context so far: a layer that leaves a search
 * pending across calls (ndt2d_matcher_score_scan launches the next matchScan's search) checks
 * with these that nobody else launched or fetched on the context in between. */
int ndt2d_match_status(ndt2d_handle h, uint64_t * n_launched, uint64_t * n_fetched);
/* launch + fetch; h_scores (host pointer, optional) receives the slab scores. */
int ndt2d_match(ndt2d_handle h, size_t th_begin, size_t th_end, double * h_scores,
                ndt2d_match_result * out);

/* Batched ScanMatcherNDT::scorePoints (src/scan_matcher_ndt.cpp:156-178) =
 * the body of ParticleFilter::measure's loop (src/particle_filter.cpp:81-87):
 * scores[i] = sum_k -likelihood(T(pose_i) * beam_k) / n_beams for the beams of
 * ndt2d_set_beams.  d_* are device pointers; poses are {x, y, theta} triples.
 * d_stats (optional) receives NDT2D_POSE_STATS_DOUBLES doubles:
 * {sum w, sum w*x, sum w*y, sum w*cos(theta), sum w*sin(theta), sum w*x*x,
 *  sum w*x*y, sum w*y*y} with w = scores[i] (un-normalised), the sums
 * ParticleFilter::updateStatistics needs (src/particle_filter.cpp:166-200). */
#define NDT2D_POSE_STATS_DOUBLES 8
int ndt2d_score_poses_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                             double * d_scores, double * d_stats);
/* Host-pointer convenience: H2D poses, launch, D2H scores (+ stats).  Up to 8 poses
 * (scorePoints / scoreScan call it with ONE) take a block-per-pose kernel whose poses are
 * kernel arguments and whose scores land in host-coherent memory: one launch, no copy,
 * bit-identical scores.  Buffers from ndt2d_host_alloc are read / written by the kernel
 * in place (no copy is queued). */
int ndt2d_score_poses(ndt2d_handle h, const double * h_poses_xyt, size_t n_poses,
                      double * h_scores, double * h_stats);

/* ndt2d_set_beams + ndt2d_score_poses in one call.  Up to 8 poses and up to 208 beams
 * (a scoreScan with the plugin's default laser_max_beams = 100) travel as kernel
 * arguments: one launch, no copy at all, and the kernel leaves the beams in the context's
 * beam buffer for the calls that follow on the same scan.  Anything larger falls back to
 * the two calls. */
int ndt2d_score_poses_beams(ndt2d_handle h, const double * beams_xy, size_t n_beams,
                            const double * h_poses_xyt, size_t n_poses, double * h_scores);

/* The kernel-argument launch of ndt2d_score_poses_beams without the wait (<= 8 poses; beams_xy
 * with <= 208 beams, or NULL: the beams the context holds), and the wait.  Between the two the
 * caller may queue the search of the same scan -- the mapper calls scoreScan(scan) and then
 * matchScan(scan, ...) (reference src/ndt_mapper.cpp:514-515, 552-553), and a search queued
 * behind the scoring kernel starts when that ends instead of a host round trip later
 * (ndt2d_matcher_score_scan does this once it has seen the pair).  NDT2D_ERR_STATE when the
 * request is not a kernel-argument launch.  One launch may be pending per context. */
int ndt2d_score_poses_beams_launch(ndt2d_handle h, const double * beams_xy, size_t n_beams,
                                   const double * h_poses_xyt, size_t n_poses);
int ndt2d_score_fetch(ndt2d_handle h, double * h_scores);

/* ParticleFilter::updateStatistics (src/particle_filter.cpp:163-218) on the
 * device, from the (all-reduced) moment sums d_stats of ndt2d_score_poses_launch:
 * d_weights[n] are divided by the total weight in place (:171-174) and d_out
 * receives NDT2D_PF_RESULT_DOUBLES doubles {sum w, mean x, mean y, mean theta
 * (circular, :205), cov xx, cov xy, cov yy (:208-215), sum_i w_i *
 * shortest_angular_distance(theta_i, mean theta)^2 (:213-217, to be ADDED to the
 * caller's cov(2,2), which the reference never zeroes)}.  Device pointers,
 * asynchronous. */
#define NDT2D_PF_RESULT_DOUBLES 8
int ndt2d_pf_finalize_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                             double * d_weights, const double * d_stats, double * d_out);
/* The same in the form one device of a SHARDED particle set takes (ndt2d_matcher_create_multi,
 * host exchange): no copy and no stream synchronisation between the two halves of
 * ParticleFilter::measure.
 *   ndt2d_pose_sums_launch   = ndt2d_score_poses_launch whose eight moment sums
 *       (NDT2D_POSE_STATS_DOUBLES) also go to the context's host-coherent result block, behind
 *       a flag;  ndt2d_pose_sums_fetch spins on that flag and returns them (this device's row
 *       of the [n_dev, 8] table: the "total particle weight" exchange, src/particle_filter.cpp:166-174);
 *   ndt2d_pf_finalize_totals_launch = ndt2d_pf_finalize_launch with the TOTAL sums given as
 *       eight host values that travel as kernel arguments; its result lands in the host-coherent
 *       block as well: ndt2d_pf_result_read returns it once the caller has synchronised the
 *       stream (it does, behind its copy of the weights).
 * One such pair may be in flight per context. */
int ndt2d_pose_sums_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses, double * d_scores);
int ndt2d_pose_sums_fetch(ndt2d_handle h, double * sums_out);
int ndt2d_pf_finalize_totals_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                                    double * d_weights, const double * totals);
int ndt2d_pf_result_read(ndt2d_handle h, double * out);
/* Host-pointer convenience = ParticleFilter::measure for the beams of
 * ndt2d_set_beams: H2D particles, score, statistics, D2H normalised weights and
 * the NDT2D_PF_RESULT_DOUBLES result.  (From 131,072 particles the call is pipelined in pieces:
 * see ndt2d_set_pipeline_pieces for what that means for the last bits of the statistics.) */
int ndt2d_pf_measure(ndt2d_handle h, const double * h_poses_xyt, size_t n_poses,
                     double * h_weights, double * h_out);

/* ---- particle-filter steps either side of `measure` (particles stay in HBM) ----
 *
 * The reference draws three std::normal_distribution<float> values per particle
 * from an mt19937 seeded by std::random_device (include/ndt_2d/motion_model.hpp:
 * 63-64, particle_filter.hpp), i.e. not reproducibly.  Every call below takes the
 * draws either as d_noise = [n][3] float standard normals (device pointer), or
 * with d_noise == NULL from a counter-based stream: Philox4x32-10 keyed by
 * `seed`, counter (first_index + i, step) for particle i, Box-Muller.  The stream
 * depends on (seed, step, global particle index) only, so shards of a particle set
 * on different GPUs draw what the whole set would.  ndt2d_pf_noise_launch writes
 * that stream out (the same numbers a fused call uses). */
int ndt2d_pf_noise_launch(ndt2d_handle h, uint64_t seed, uint64_t step, uint64_t first_index,
                          size_t n, float * d_noise_out);
/* MotionModel::sample (src/motion_model.cpp:45-83) applied in place to
 * d_poses_xyt[n][3]; alphas5 = the model's {a1..a5} (motion_model.cpp:39-43;
 * a5 is unused by the reference too).  Asynchronous. */
int ndt2d_pf_motion_launch(ndt2d_handle h, double * d_poses_xyt, size_t n, double dx, double dy,
                           double dth, const double * alphas5, const float * d_noise,
                           uint64_t seed, uint64_t step, uint64_t first_index);
/* ParticleFilter::init sampling loop (src/particle_filter.cpp:53-65): poses
 * written as (float draw) widened to double, theta through normalize_angle. */
int ndt2d_pf_init_launch(ndt2d_handle h, double * d_poses_xyt, size_t n, double x, double y,
                         double theta, double sigma_x, double sigma_y, double sigma_theta,
                         const float * d_noise, uint64_t seed, uint64_t step,
                         uint64_t first_index);
/* The moment sums of updateStatistics (src/particle_filter.cpp:166-200) for given
 * weights (d_weights == NULL: uniform 1/n, as init assigns at :67): d_stats gets
 * NDT2D_POSE_STATS_DOUBLES doubles in ndt2d_score_poses_launch's layout, ready for
 * ndt2d_pf_finalize_launch (after an all-reduce when the set is sharded). */
int ndt2d_pose_moments_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n,
                              const double * d_weights, double * d_stats);
/* Host-pointer convenience = ParticleFilter::update (src/particle_filter.cpp:
 * 71-76): motion model on h_poses_xyt in place (h_noise [n][3] or NULL = Philox),
 * then updateStatistics with h_weights (normalised in place) into h_out
 * (NDT2D_PF_RESULT_DOUBLES, see ndt2d_pf_finalize_launch). */
int ndt2d_pf_update(ndt2d_handle h, double * h_poses_xyt, size_t n, double dx, double dy,
                    double dth, const double * alphas5, const float * h_noise, uint64_t seed,
                    uint64_t step, double * h_weights, double * h_out);

/* ---- LaserScan -> Scan conversion on the device ----
 *
 * The loop of NdtMapper::laserCallback that turns a sensor_msgs/LaserScan into
 * the point list of an ndt_2d::Scan (src/ndt_mapper.cpp:385-453).  The message
 * fields keep their ROS types (float32); motion_* is `translation` (:386-389),
 * the odometry motion between the start and the end of the sweep; laser_* is
 * laser_transform_; range_max is the node's range_max_ (NaN and longer ranges are
 * dropped, :413,436); inverted is laser_inverted_ (descending visiting order,
 * index 0 never visited, :410). */
typedef struct ndt2d_laser_scan
{
  float angle_min, angle_increment;
  double range_max;
  int inverted;
  double laser_x, laser_y, laser_theta;
  double motion_x, motion_y, motion_theta;
} ndt2d_laser_scan;
/* d_ranges: device float[n_ranges]; d_points_xy_out: device double[n_ranges][2],
 * filled with the kept points in the reference's order; d_info_out: device
 * double[2] = {number of points kept, upper bound of max |point|}.  Asynchronous. */
int ndt2d_convert_scan_launch(ndt2d_handle h, const float * d_ranges, size_t n_ranges,
                              const ndt2d_laser_scan * scan, double * d_points_xy_out,
                              double * d_info_out);
/* Host-pointer convenience: H2D ranges, convert, D2H points (capacity n_ranges). */
int ndt2d_convert_scan(ndt2d_handle h, const float * h_ranges, size_t n_ranges,
                       const ndt2d_laser_scan * scan, double * h_points_xy_out,
                       size_t * n_points_out);
/* Conversion fused with matchScan's beam subsampling (src/scan_matcher_ndt.cpp:
 * 95-96,110): the ranges go H2D (4 B/beam), the points and the beams of
 * ndt2d_set_beams stay on the device.  Blocks for a 24-byte D2H of the counts.
 * *n_beams_out == 0 (no point kept) leaves the context without beams. */
int ndt2d_set_beams_from_ranges(ndt2d_handle h, const float * h_ranges, size_t n_ranges,
                                const ndt2d_laser_scan * scan, size_t laser_max_beams,
                                size_t * n_points_out, size_t * n_beams_out);
/* The points of the last ndt2d_set_beams_from_ranges (device pointer, valid until
 * the next conversion on this context) and their number. */
const double * ndt2d_scan_points(ndt2d_handle h, size_t * n_points_out);

/* ---- OccupancyGrid rendering on the device ----
 *
 * OccupancyGrid::getMsg (src/occupancy_grid.cpp:47-152) with its updateBounds
 * (:154-185): every beam of every scan is ray-traced from its scan pose through
 * the reference's simplified Bresenham line, cells are counted as hit / empty and
 * published as -1 (unknown) / 0 (free) / 100 (occupied, hit ratio > occ_thresh).
 * Scans are given as for ndt2d_build_grid.
 *
 * bounds_inout = {min_x_, max_x_, min_y_, max_y_} of the generator (all 0 when it
 * is new, :37-40) and n_scans_bounded = its num_scans_: as in the reference the
 * bounds are extended by the scans [n_scans_bounded, n_scans) only, and only when
 * the two counts differ (:51-54), then rounded to the resolution (:181-184).
 * info_out receives the message's meta data (:60-65).  With data_out == NULL the
 * call stops there (use it to size the buffer); otherwise data_out[width*height]
 * (row-major, y * width + x) receives the map.  A ray cell outside the grid is
 * skipped (the reference would write out of bounds: its bounds cover the scans'
 * points, not their poses).  Counts are integers, so the result is bit-identical
 * to the sequential loop. */
typedef struct ndt2d_occupancy_info
{
  double resolution;
  uint32_t width, height;
  double origin_x, origin_y;
} ndt2d_occupancy_info;
int ndt2d_occupancy_grid(ndt2d_handle h, double resolution, double occ_thresh,
                         const double * poses_xyt, const double * points_xy,
                         const size_t * offsets, size_t n_scans, size_t n_scans_bounded,
                         double * bounds_inout, ndt2d_occupancy_info * info_out,
                         signed char * data_out, size_t data_capacity);

/* ---- device memory for hosts without a GPU runtime of their own ----
 *
 * The *_launch entry points take device pointers.  A host that already manages
 * device memory (PyTorch, a HIP application) passes its own; the C++ mirrors in
 * ndt_2d_amd/plugin/ (particle_filter_hip.hpp) use these four.  Copies are
 * ordered on the context's stream and return when the data has arrived. */
int ndt2d_device_alloc(ndt2d_handle h, size_t bytes, void ** d_out);
int ndt2d_device_free(ndt2d_handle h, void * d_ptr);
int ndt2d_copy_to_device(ndt2d_handle h, void * d_dst, const void * h_src, size_t bytes);
int ndt2d_copy_to_host(ndt2d_handle h, void * h_dst, const void * d_src, size_t bytes);
/* The same copies queued on the context's stream without the wait: the host buffer must stay
 * valid (and, for a copy that really is asynchronous, be pinned: ndt2d_host_alloc) until
 * ndt2d_synchronize.  A multi-device matcher feeds its devices with these. */
int ndt2d_copy_to_device_async(ndt2d_handle h, void * d_dst, const void * h_src, size_t bytes);
int ndt2d_copy_to_host_async(ndt2d_handle h, void * h_dst, const void * d_src, size_t bytes);

/* Pinned, GPU-mapped host memory: poses / weights buffers allocated here are read and
 * written by the kernels of the host-pointer entry points directly over PCIe (no staging
 * copy, no copy command); the C++ ParticleFilter mirror keeps its particles in it. */
int ndt2d_host_alloc(ndt2d_handle h, size_t bytes, void ** out);
/* h may be NULL once the context that allocated `ptr` has been destroyed (nothing of its
 * stream can then be in flight): the buffer outlives its context if the caller wants so. */
int ndt2d_host_free(ndt2d_handle h, void * ptr);

/* HIP events around the dominant kernel of every launch (ndt2d_last_launch_ms /
 * ndt2d_launch_history_ms) are recorded by default; a latency-critical host (the
 * pluginlib shim) turns them off: a recorded event holds the stream up for ~5.5 us where
 * consecutive kernels otherwise start back to back, and the pair costs ~4.5 us of host
 * time per call. */
int ndt2d_set_timing(ndt2d_handle h, int enabled);

/* Block until everything launched on the context's stream has finished. */
int ndt2d_synchronize(ndt2d_handle h);
/* GPU time (HIP events on the launch stream) of the dominant kernel -- the
 * search / scoring kernel, without the few-microsecond final reduction -- of the
 * most recent ndt2d_match_launch / ndt2d_score_poses_launch.  Synchronises.
 * *n_kernels (optional) = kernels launched by that call. */
int ndt2d_last_launch_ms(ndt2d_handle h, float * ms, int * n_kernels);
/* The same figure for the last launches, oldest first (the context keeps the
 * event pairs of its last NDT2D_TIMING_HISTORY launches): lets a caller queue
 * launches back to back and read the kernel durations afterwards.  Writes at
 * most `capacity` values, *n_out = how many; blocks until the newest finished. */
#define NDT2D_TIMING_HISTORY 256
int ndt2d_launch_history_ms(ndt2d_handle h, float * ms_out, size_t capacity, size_t * n_out);
/* Tuning / introspection: name of the kernel variant the last launch used. */
const char * ndt2d_last_variant(ndt2d_handle h);
/* "auto" (the default) picks the candidate mapping of the match search by the size
 * of the lattice: lane-per-candidate with the beams split across the waves of a block
 * ("small") below 4,096 (theta, 8x8 patch) work items -- the plugin's default search is
 * 720 -- and lane-per-candidate with persistent waves ("lane") above; wave-per-candidate
 * ("wave") where neither applies (search windows beyond 1,024 cells, NaN beams).  All give
 * the oracle's result; they differ from each other in the last bits of a score
 * (summation order).
 * Force a kernel variant (testing / A-B measurement): "auto", "lds", "global"
 * (grid placement), "wave", "wave-lds", "wave-global", "lane", "small" (candidate mapping
 * of the match search), "lane-noskip" / "small-noskip" (the lane mappings with every term
 * evaluated: the bit-exactness controls of their skipping), "batched" (pose batches of
 * at most 2,048 on the batched particle kernel instead of block-per-pose), "dense" (particle
 * scoring without compaction), "compact-exact" (particle scoring with the exact
 * FP64 phase A: the bit-exactness control of the FP32 screen). */
int ndt2d_set_variant(ndt2d_handle h, const char * name);
/* A batch of 131,072 poses or more handed over in ordinary host memory (ndt2d_score_poses,
 * ndt2d_pf_measure: ParticleFilter::measure of a large filter, reference
 * src/particle_filter.cpp:78-89) is cut into pieces: piece k + 1 is uploaded on a stream of its own
 * while piece k is scored, raw scores travel back under the piece after.  The raw scores do not
 * depend on the cut (bit-identical), the pieces' moment sums are added in piece order.
 * CONTRACT on the statistics: the eight moment sums (total weight, sum w x, ...) are sums of a
 * million terms whose ORDER follows the cut -- pieces here, devices in a multi-device matcher
 * (ndt2d_matcher_set_multi_thresholds), blocks of the one-launch form -- so the normalised weights,
 * mean and covariance of ndt2d_pf_measure / ndt2d_score_poses(h_stats) agree between any two
 * settings of this knob (and between one and several devices) to within 64 ulps of the sums'
 * magnitude, not bit for bit; a caller that needs the same bits every time fixes the knob.
 * tests/test_gpu_pose_batch_pipeline.py holds every setting to that bound.
 * pieces: 0 = default (4), 1 = off (one upload, one launch, one download), at most 16. */
int ndt2d_set_pipeline_pieces(ndt2d_handle h, int pieces);
/* Pieces the last ndt2d_score_poses / ndt2d_pf_measure was cut into (1: not pipelined). */
int ndt2d_last_pipeline_pieces(ndt2d_handle h);

/* ------------------------------------------------------------------------ */
/* (2) matcher layer: ndt_2d::ScanMatcherNDT restated over the device layer  */
/* ------------------------------------------------------------------------ */

typedef struct ndt2d_matcher ndt2d_matcher;

int ndt2d_matcher_create(ndt2d_matcher ** out, int device_id);
/* One matcher over n_dev GPUs of this process (`ndt2d_create(handle*, const int*
 * device_ids, int n_dev)`): the plugin object the unchanged node holds, with the 8-GPU split of
 * the loop-closure search (reference src/ndt_mapper.cpp:634-643 calls a plain matchScan) behind
 * it.  One device context and stream per entry of device_ids, all driven by the calling thread;
 * grid, beams and search tables are replicated by host-to-device copies.
 *
 *   matchScan   theta steps dealt round-robin (device r takes r, r + n_dev, ...: the cost of a
 *               step varies across the angular range), one 12-double record per device, the
 *               records exchanged ONCE and combined with the reference's first-wins rule
 *               (strict `<` in visiting order, src/scan_matcher_ndt.cpp:128: the lower score,
 *               between equal scores the lower flat index), accumulators summed in device order;
 *   scorePoses / pf_measure   contiguous particle ranges, one exchange of the [n_dev, 8] moment
 *               sums (the "total particle weight" of src/particle_filter.cpp:166-174), and the
 *               theta variance of the reference's second pass (:213-217) with a second one.
 *
 * The exchange (ndt2d_matcher_set_exchange): "rccl" = ONE in-place ncclAllReduce(sum) of the
 * [n_dev, 12] (or [n_dev, 8]) table per device -- every device fills its own row, x + 0 is
 * exact -- in one ncclGroupStart / ncclGroupEnd over single-process communicators
 * (ncclCommInitAll; xGMI between the devices), the table then read back from the first device;
 * "host" = no collective: every device's final reduction writes its record into its context's
 * host-coherent result block and the host combines them (a device may then appear more than
 * once in device_ids: several contexts on one GPU, which RCCL refuses).  "auto" (default):
 * "rccl" when all devices differ and librccl.so.1 loads, "host" otherwise.  Both give the same
 * bits.
 *
 * Dealing: one persistent worker thread per device beyond the first (made here, parked on a
 * condition variable after ~200 us without work), the calling thread drives the first device:
 * all devices' uploads and launches go out side by side (ndt2d_matcher_last_fanout_us: when
 * each device's launch had been queued, from the call's start).  Sharded particle sets run
 * their whole share on the device's thread -- upload, scoring, the sums' meeting on the host
 * (a barrier between the threads, no stream synchronisation), updateStatistics with the totals
 * as kernel arguments, the weights' way back.
 *
 * Work smaller than the thresholds of ndt2d_matcher_set_multi_thresholds (candidates x beams
 * of a search, default 1e9 -- ~0.3 ms of one GPU; particles x beams of a batch, default 2e8
 * -- ~0.2 ms: BASELINE configs[4], 7.2e8, is sharded) and every single-pose call stay on the
 * first device.  ndt2d_matcher_set_multi_min_units sets both to one value.
 * n_dev == 1 behaves exactly as ndt2d_matcher_create. */
int ndt2d_matcher_create_multi(ndt2d_matcher ** out, const int * device_ids, int n_dev);
int ndt2d_matcher_destroy(ndt2d_matcher * m);
const char * ndt2d_matcher_last_error(ndt2d_matcher * m);
int ndt2d_matcher_device_count(ndt2d_matcher * m);
/* The device context of rank `rank` (0 <= rank < device_count), NULL otherwise. */
ndt2d_handle ndt2d_matcher_device_at(ndt2d_matcher * m, int rank);
int ndt2d_matcher_set_exchange(ndt2d_matcher * m, const char * mode);
int ndt2d_matcher_set_multi_min_units(ndt2d_matcher * m, double units);
int ndt2d_matcher_set_multi_thresholds(ndt2d_matcher * m, double min_search_units, double min_pose_units);
int ndt2d_matcher_get_multi_thresholds(ndt2d_matcher * m, double * min_search_units, double * min_pose_units);
/* Of the last call that was dealt to the devices: for device r, microseconds from the call's
 * start until its (first) launch had been queued -- out_us[r], r < min(capacity, n_dev);
 * *n_out = n_dev (0 if no call was dealt yet). */
int ndt2d_matcher_last_fanout_us(ndt2d_matcher * m, double * out_us, size_t capacity, size_t * n_out);
/* What the last matchScan / scorePoses / pf_measure ran as: the kernel variant of the first
 * device (ndt2d_last_variant), prefixed "multi[n]/rccl/" or "multi[n]/host/" when the call was
 * dealt to n devices. */
const char * ndt2d_matcher_last_variant(ndt2d_matcher * m);
/* ndt2d_set_timing on every device of the matcher. */
int ndt2d_matcher_set_timing(ndt2d_matcher * m, int enabled);
/* The (first) device context the matcher drives (for sharded launches / streams).  The matcher
 * remembers which beams it put there (a scan that arrives again is not uploaded again):
 * it must remain the only writer of this context's beams. */
ndt2d_handle ndt2d_matcher_device(ndt2d_matcher * m);

/* ScanMatcherNDT::initialize (src/scan_matcher_ndt.cpp:35-47): the six
 * declared parameters (defaults 0.25, 0.0025, 0.1, 0.005, 0.05, 100) and
 * range_max. */
int ndt2d_matcher_initialize(ndt2d_matcher * m, double ndt_resolution,
                             double search_angular_resolution, double search_angular_size,
                             double search_linear_resolution, double search_linear_size,
                             size_t laser_max_beams, double range_max);
/* ScanMatcherNDT::addScans (src/scan_matcher_ndt.cpp:49-74): scan k has pose
 * poses_xyt[3k..3k+2] and robot-frame points
 * points_xy[2*offsets[k] .. 2*offsets[k+1]).  Builds the NDT on the host with
 * the reference's incremental formulas (src/ndt_model.cpp:50-103,132-160) and
 * uploads it. */
int ndt2d_matcher_add_scans(ndt2d_matcher * m, const double * poses_xyt,
                            const double * points_xy, const size_t * offsets, size_t n_scans);
/* Where addScans builds the NDT: "host" (C++ on the host, then upload), "device"
 * (ndt2d_build_grid) or "auto" (device from 73,728 map points up: ~100 scans of 720 beams).  Both give
 * bit-identical grids. */
int ndt2d_matcher_set_build_mode(ndt2d_matcher * m, const char * mode);
/* ndt2d_set_eigenvalue_form for the host build and every device of the matcher. */
int ndt2d_matcher_set_eigenvalue_form(ndt2d_matcher * m, const char * form);
/* ScanMatcherNDT::matchScan (src/scan_matcher_ndt.cpp:76-149).  *score_out =
 * the function's return value (best_score / scan_points_to_use; 0.0 and
 * outputs untouched when no NDT, :80).  pose_inout[3] is written only when a
 * candidate scores < 0 (:128-134); covariance_out[9] row-major (:146). */
int ndt2d_matcher_match_scan(ndt2d_matcher * m, const double * scan_pose_xyt,
                             const double * points_xy, size_t n_points, double * pose_inout,
                             double * covariance_out, double * score_out);
/* Same search, additionally returning the raw score of every candidate
 * (all_scores, host, capacity all_scores_cap), the candidate count and the
 * winner's flat index; any of the extra outputs may be NULL. */
int ndt2d_matcher_match_scan_ex(ndt2d_matcher * m, const double * scan_pose_xyt,
                                const double * points_xy, size_t n_points,
                                double * pose_inout, double * covariance_out,
                                double * score_out, double * all_scores,
                                size_t all_scores_cap, size_t * n_candidates_out,
                                uint64_t * best_index_out);
/* The two halves of matchScan, for sharded (multi-GPU) searches:
 * prepare_search subsamples the scan (:95-96,110), builds the offset and
 * cos/sin tables (:103-107,117,119) and uploads them -- after it,
 * ndt2d_match_launch(ndt2d_matcher_device(m), th_begin, th_end, ...) evaluates
 * any theta slab; finish_match turns a (combined) NDT2D_MATCH_RECORD_DOUBLES
 * record into matchScan's outputs (:128-134,146,148). */
int ndt2d_matcher_prepare_search(ndt2d_matcher * m, const double * scan_pose_xyt,
                                 const double * points_xy, size_t n_points, size_t * n_th_out,
                                 size_t * n_lin_out, size_t * n_beams_out);
int ndt2d_matcher_finish_match(ndt2d_matcher * m, const double * record, double * pose_inout,
                               double * covariance_out, double * score_out);
/* Subsample + upload the beams only (particle path; then
 * ndt2d_score_poses_launch on ndt2d_matcher_device(m)). */
int ndt2d_matcher_prepare_beams(ndt2d_matcher * m, const double * points_xy, size_t n_points,
                                size_t * n_beams_out);
/* ScanMatcherNDT::scoreScan (:151-154) and scorePoints (:156-178).
 * The mapper calls scoreScan(scan) and then matchScan(scan, ...) (src/ndt_mapper.cpp:514-515,
 * 552-553).  Once a matcher has seen that pair -- a matchScan of the scan and pose of the
 * scoreScan just before it -- its scoreScan queues the scan's search behind the scoring kernel
 * before it waits for the score, and the matchScan that follows collects that search (same
 * kernels, same results; the search starts when the scoring kernel ends instead of a host
 * round trip later).  Any other call waits such a search out, drops it, and the matcher stops
 * launching ahead until it sees the pair again.  ndt2d_matcher_set_search_ahead(m, 0) turns
 * it off (default: on). */
int ndt2d_matcher_score_scan(ndt2d_matcher * m, const double * scan_pose_xyt,
                             const double * points_xy, size_t n_points, double * score_out);
int ndt2d_matcher_set_search_ahead(ndt2d_matcher * m, int enabled);
/* Near-tie adjudication (default: on).  When a search's winner comes back with near_tie set,
 * matchScan lists the candidates within the tolerance of the best (ndt2d_match_near_best with
 * NDT2D_NEAR_TIE_REL, the first 256 in visiting order), rescores each on the host exactly as the reference does (points_outer / points_inner,
 * NDT::likelihood in beam order, libm's exp; src/scan_matcher_ndt.cpp:106-127) and applies the
 * reference's rule -- strict `<` in visiting order (:128-134): the lowest host score, between equal
 * ones the lowest flat index.  The returned pose / score are then that candidate's (the score
 * the reference's own bits); the covariance sums are unaffected.  Needs the host copy of the
 * NDT and of the beams (not available to ndt2d_matcher_match_laser_scan, which skips it).
 * stats: searches that came back marked, and how many of those changed the winner. */
int ndt2d_matcher_set_adjudication(ndt2d_matcher * m, int enabled);
/* The same for a search that was sharded from outside (one process per GPU, ndt_2d_amd/dist.py):
 * record_inout[NDT2D_MATCH_RECORD_DOUBLES] is the COMBINED record of all shards; if its winner
 * is marked (index + 0.5) it is settled as above -- on this matcher's own device, over the whole
 * lattice of the last ndt2d_matcher_prepare_search, which every rank holds: all ranks reach the
 * same verdict without talking to each other -- and leaves with a plain index either way. */
int ndt2d_matcher_settle_near_tie(ndt2d_matcher * m, const double * scan_pose_xyt, double * record_inout);
int ndt2d_matcher_adjudication_stats(ndt2d_matcher * m, uint64_t * marked, uint64_t * changed, uint64_t * truncated);
/* Where ONE pose is scored (scorePoints, scoreScan).  "host" (default): a scan of at most
 * max_beams subsampled beams (default 256; 0 keeps the current value) is scored by the calling
 * thread from the host copy of the NDT, in the reference's order with libm's exp
 * (src/scan_matcher_ndt.cpp:156-178, src/ndt_model.cpp:105-116,162-170,203-218) -- the unchanged
 * ParticleFilter::measure calls scorePoints once per particle (src/particle_filter.cpp:81-87),
 * and a kernel launch plus a PCIe round trip per call costs several times the ~100 cell
 * evaluations it is for (SURVEY.md 8b foresees exactly this path for the per-pose virtual
 * call).  It needs a live matcher -- a GPU -- all the same: searches, batches
 * (ndt2d_matcher_score_poses / _pf_measure) and longer scans always run on the device, and
 * "device" sends the single poses there as well (the parity tests run both and compare). */
int ndt2d_matcher_set_single_pose_path(ndt2d_matcher * m, const char * where, size_t max_beams);
/* Searches launched ahead by scoreScan, and how many of them a matchScan collected. */
int ndt2d_matcher_search_ahead_stats(ndt2d_matcher * m, uint64_t * launched, uint64_t * collected);
int ndt2d_matcher_score_points(ndt2d_matcher * m, const double * points_xy, size_t n_points,
                               const double * pose_xyt, double * score_out);
/* ScanMatcherNDT::reset (:180-183). */
int ndt2d_matcher_reset(ndt2d_matcher * m);
int ndt2d_matcher_has_ndt(ndt2d_matcher * m);

/* laserCallback's LaserScan -> Scan conversion (src/ndt_mapper.cpp:385-453) fused
 * with matchScan: the raw ranges go to the device, where they are converted,
 * de-skewed, subsampled and searched; *n_points_out (optional) = points the
 * conversion kept.  Same outputs and conventions as ndt2d_matcher_match_scan on
 * the converted points. */
int ndt2d_matcher_match_laser_scan(ndt2d_matcher * m, const double * scan_pose_xyt,
                                   const float * ranges, size_t n_ranges,
                                   const ndt2d_laser_scan * scan, double * pose_inout,
                                   double * covariance_out, double * score_out,
                                   size_t * n_points_out);

/* Additive batched interface (not in the reference's ScanMatcher): scores
 * n_poses poses in one launch; scores_out[i] == scorePoints(points, pose_i). */
int ndt2d_matcher_score_poses(ndt2d_matcher * m, const double * points_xy, size_t n_points,
                              const double * poses_xyt, size_t n_poses, double * scores_out);
/* ParticleFilter::measure (src/particle_filter.cpp:78-89) incl. its
 * updateStatistics (:163-218): weights_out[n] = normalised weights,
 * mean_out[3], cov_inout[9] row-major ((2,2) accumulates onto the previous
 * value, :216). */
int ndt2d_matcher_pf_measure(ndt2d_matcher * m, const double * particles_xyt,
                             size_t n_particles, const double * points_xy, size_t n_points,
                             double * weights_out, double * mean_out, double * cov_inout);

/* Host NDT introspection (tests compare it bit-for-bit with the oracle). */
int ndt2d_matcher_grid_info(ndt2d_matcher * m, uint32_t * size_x, uint32_t * size_y,
                            double * cell_size, double * origin_x, double * origin_y);
int ndt2d_matcher_grid_cells6(ndt2d_matcher * m, double * cells6_out, size_t capacity_cells);
/* The search lattice the matcher visits (the reference's FP-accumulated
 * loops): writes up to cap values, returns the count through *n_out. */
int ndt2d_search_offsets(double size, double res, double * out, size_t cap, size_t * n_out);
/* The draw-and-stop loop of ParticleFilter::resample (src/particle_filter.cpp:
 * 94-134), host code: KLD sampling is a sequential stopping rule and stays on the
 * CPU.  Draw i picks the particle whose cumulative weight
 * first exceeds uniforms[i] * sum(weights) (what std::discrete_distribution does
 * with its own generator, :94,110; the caller supplies the uniforms in [0, 1), so
 * any generator can drive it), inserts its KD-tree key
 * static_cast<int>(value / leaf_size3[d]) (kd_tree.hpp:95-98; the leaf count is the
 * number of distinct keys) and recomputes Mx (:117-126); the loop ends when the
 * count reaches max(min_particles, Mx) or max_particles (:107,129-132).
 * indices_out[max_particles] receives the chosen particle of every draw kept,
 * *n_out their number.  n_uniforms must be at least max_particles. */
int ndt2d_kld_resample(const double * particles_xyt, const double * weights, size_t n,
                       size_t min_particles, size_t max_particles, double kld_err, double kld_z,
                       const double * leaf_size3, const double * uniforms, size_t n_uniforms,
                       uint32_t * indices_out, size_t * n_out);
/* Host-only (no GPU needed) NDT build: the arithmetic of addScans without the
 * upload, for hosts that only want the packed grid. */
int ndt2d_host_build_grid(double ndt_resolution, double range_max, const double * poses_xyt,
                          const double * points_xy, const size_t * offsets, size_t n_scans,
                          double * cells6_out, size_t capacity_cells, uint32_t * size_x,
                          uint32_t * size_y, double * origin_x, double * origin_y);
/* ... with flags: NDT2D_BUILD_SEQUENTIAL adds every scan's points one after the other, the
 * reference's loop as it stands (src/ndt_model.cpp:132-152), instead of the four quarters of a
 * scan side by side (csrc/ndt2d_host.cpp HostNdt::add_scan) -- the two give the same bits, and
 * tests/test_host_logic.py holds them to it; NDT2D_BUILD_CLOSED_FORM takes the closed-form
 * eigenvalues (ndt2d_matcher_set_eigenvalue_form "closed"). */
#define NDT2D_BUILD_SEQUENTIAL 1u
#define NDT2D_BUILD_CLOSED_FORM 2u
int ndt2d_host_build_grid_ex(double ndt_resolution, double range_max, const double * poses_xyt,
                             const double * points_xy, const size_t * offsets, size_t n_scans, unsigned flags,
                             double * cells6_out, size_t capacity_cells, uint32_t * size_x,
                             uint32_t * size_y, double * origin_x, double * origin_y);

/* ------------------------------------------------------------------------ */
/* Synthetic workload generator (BASELINE.md section 3 / SURVEY.md 8d)       */
/* ------------------------------------------------------------------------ */

/* Closed square room [-half, half]^2 with 0.5 m square pillars centred on the
 * lattice (pitch*i + pitch/2, pitch*j + pitch/2) that fit inside the room. */
typedef struct ndt2d_world
{
  double room_half;
  double pillar_pitch;
  double pillar_half;
} ndt2d_world;

/* Ray-cast one n_beams scan over 2*pi (angle_min = -pi) from pose, range noise
 * N(0, noise_sigma^2) from splitmix64(seed) + Box-Muller; writes robot-frame
 * points_xy[2*n_beams]. */
int ndt2d_synth_scan(const ndt2d_world * world, const double * pose_xyt, size_t n_beams,
                     double noise_sigma, uint64_t seed, double * points_xy);
/* 1 if the pose is inside or within `margin` (Chebyshev) of a pillar. */
int ndt2d_synth_pose_blocked(const ndt2d_world * world, double x, double y, double margin);
/* n uniforms in [0,1) from splitmix64(seed). */
int ndt2d_synth_uniform(uint64_t seed, size_t n, double * out);

#ifdef __cplusplus
}
#endif

#endif  /* NDT2D_HIP_H_ */
