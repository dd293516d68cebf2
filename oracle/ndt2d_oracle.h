/*
 * ndt2d_oracle.h -- CPU restatement of ndt_2d's NDT scan-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C, dependency
 * free restatement of the reference arithmetic, same operation order as the
 * cited reference lines.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under ndt_2d_amd/ links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - Cell::addPoint/compute/score, NDT::addScan/compute/likelihood/getIndex
 *     are PINNED against the reference's own known-answer tests
 *     (reference test/ndt_model_tests.cpp:32-230) in
 *     tests/test_oracle_reference_vectors.py.
 *   - ScanMatcherNDT::matchScan/scorePoints/scoreScan/addScans and
 *     ParticleFilter::measure/updateStatistics: PARITY UNPINNED -- the
 *     reference holds no test, fixture or golden vector for them, and the
 *     reference cannot be compiled here (Eigen3, rclcpp, pluginlib, tf2,
 *     angles are absent; no network).
 *   - MotionModel::sample: pinned statistically only -- the reference's
 *     scenario (test/particle_tests.cpp:74-140, 50 poses, tolerances 0.3/0.5)
 *     is run in tests/test_particle_host.py; the reference's RNG is seeded by
 *     std::random_device, so no bit-level vector can exist.  ParticleFilter::
 *     init: PARITY UNPINNED (no reference test).
 *   - LaserScan conversion (NdtMapper::laserCallback) and OccupancyGrid:
 *     PARITY UNPINNED (no reference test); each is cross-checked against an
 *     independent numpy / Python statement of the same lines in tests/.
 *
 * Third-party arithmetic restated (not under /root/reference):
 *   Eigen3 (unpinned; 3.4.0 on ROS 2 Humble): fixed-size 2x2 inverse,
 *   EigenSolver<Matrix2d> eigenvalues (restated in closed form, see
 *   orc_cell_compute), Isometry3d * Vector3d, AngleAxisd -> matrix.
 *   ROS `angles` (unpinned): normalize_angle / shortest_angular_distance.
 *   libm exp/cos/sin/atan2/fmod come from the host glibc.
 *
 * All matrices are row-major double[4] / double[9].
 */
#ifndef NDT2D_ORACLE_H_
#define NDT2D_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* struct Cell, reference include/ndt_2d/ndt_model.hpp:43-65 */
typedef struct orc_cell
{
  int valid;
  double n;
  double mean[2];
  double covariance[4];
  double correlation[4];
  double information[4];
} orc_cell;

void orc_cell_init(orc_cell * c);                                /* ndt_model.cpp:40-48 */
void orc_cell_add_point(orc_cell * c, double x, double y);       /* ndt_model.cpp:50-63 */
void orc_cell_compute(orc_cell * c);                             /* ndt_model.cpp:65-103 */
/* How orc_cell_compute forms the eigenvalues of :84-85: 0 (default) = Eigen 3.4.0's
 * EigenSolver<Matrix2d> transcribed, 1 = the closed form d + p +- z (see ndt2d_oracle.c). */
void orc_set_eigen_form(int form);
int orc_get_eigen_form(void);
double orc_cell_score(const orc_cell * c, double x, double y);   /* ndt_model.cpp:105-116 */

/* class NDT, reference include/ndt_2d/ndt_model.hpp:67-134 */
typedef struct orc_ndt orc_ndt;

orc_ndt * orc_ndt_create(double cell_size, double size_x, double size_y,
                         double origin_x, double origin_y);      /* ndt_model.cpp:118-126 */
void orc_ndt_destroy(orc_ndt * ndt);
void orc_ndt_add_scan(orc_ndt * ndt, double pose_x, double pose_y, double pose_theta,
                      const double * points_xy, size_t n_points); /* ndt_model.cpp:132-152 */
void orc_ndt_compute(orc_ndt * ndt);                             /* ndt_model.cpp:154-160 */
double orc_ndt_likelihood_point(const orc_ndt * ndt, double x, double y);      /* :162-170 */
double orc_ndt_likelihood_points(const orc_ndt * ndt, const double * points_xy,
                                 size_t n_points);               /* :178-187 */
double orc_ndt_likelihood_scan(const orc_ndt * ndt, double pose_x, double pose_y,
                               double pose_theta, const double * points_xy,
                               size_t n_points);                 /* :189-201 */
int orc_ndt_get_index(const orc_ndt * ndt, double x, double y);  /* :203-218 */
size_t orc_ndt_size_x(const orc_ndt * ndt);
size_t orc_ndt_size_y(const orc_ndt * ndt);
double orc_ndt_cell_size(const orc_ndt * ndt);
double orc_ndt_origin_x(const orc_ndt * ndt);
double orc_ndt_origin_y(const orc_ndt * ndt);
const orc_cell * orc_ndt_cells(const orc_ndt * ndt);
/* Pack every cell as {mean_x, mean_y, info00, info01, info11, n} (6 doubles). */
void orc_ndt_export_cells6(const orc_ndt * ndt, double * out);

/* The reference's `for (v = -size; v < size; v += res)` loop
 * (scan_matcher_ndt.cpp:103,117,119).  Writes at most cap values, returns the
 * number of iterations the loop makes. */
size_t orc_search_offsets(double size, double res, double * out, size_t cap);

/* class ScanMatcherNDT, reference include/ndt_2d/scan_matcher_ndt.hpp:42-105 */
typedef struct orc_matcher orc_matcher;

orc_matcher * orc_matcher_create(void);
void orc_matcher_destroy(orc_matcher * m);
/* initialize(): the six declared parameters + range_max (scan_matcher_ndt.cpp:35-47) */
void orc_matcher_initialize(orc_matcher * m, double ndt_resolution,
                            double search_angular_resolution, double search_angular_size,
                            double search_linear_resolution, double search_linear_size,
                            size_t laser_max_beams, double range_max);
/* addScans(begin, end) (scan_matcher_ndt.cpp:49-74).  Scan k has pose
 * poses_xyt[3k..3k+2] and points points_xy[2*offsets[k] .. 2*offsets[k+1]). */
void orc_matcher_add_scans(orc_matcher * m, const double * poses_xyt,
                           const double * points_xy, const size_t * offsets,
                           size_t n_scans);
/* matchScan (scan_matcher_ndt.cpp:76-149).  pose_inout is written only when a
 * candidate scores < best (caller pre-initialises it, as the reference's
 * callers do).  covariance_out row-major 3x3.  all_scores (optional, may be
 * NULL) receives the raw per-candidate score (= -likelihood sum) in loop
 * order, at most all_scores_cap values; *n_candidates_out (optional) the
 * number of candidates visited; *best_index_out (optional) the flat loop
 * index of the winner or UINT64_MAX. */
double orc_matcher_match_scan(const orc_matcher * m, const double * scan_pose_xyt,
                              const double * points_xy, size_t n_points,
                              double * pose_inout, double * covariance_out,
                              double * all_scores, size_t all_scores_cap,
                              size_t * n_candidates_out, uint64_t * best_index_out);
/* Same arithmetic per candidate, the (theta, dx) strips of the lattice dealt to OpenMP
 * threads (CPU baseline on all host cores; full-lattice winners for the golden
 * fixtures).  Strips are combined in lattice order: the winner (score, pose, flat
 * index) is the sequential loop's; the covariance accumulators are per-strip partial
 * sums added in lattice order.  *threads_used_out (optional) = threads of the team. */
/* ... and every candidate's raw score (= -likelihood sum, src/scan_matcher_ndt.cpp:127)
 * into all_scores[flat index] (each strip writes its own range). */
double orc_matcher_match_scan_omp_scores(const orc_matcher * m, const double * scan_pose_xyt,
                                         const double * points_xy, size_t n_points,
                                         double * pose_inout, double * covariance_out,
                                         int n_threads, uint64_t * best_index_out,
                                         int * threads_used_out, double * all_scores,
                                         size_t all_scores_cap);
double orc_matcher_match_scan_omp_ex(const orc_matcher * m, const double * scan_pose_xyt,
                                     const double * points_xy, size_t n_points,
                                     double * pose_inout, double * covariance_out,
                                     int n_threads, uint64_t * best_index_out,
                                     int * threads_used_out);
double orc_matcher_match_scan_omp(const orc_matcher * m, const double * scan_pose_xyt,
                                  const double * points_xy, size_t n_points,
                                  double * pose_inout, double * covariance_out,
                                  int n_threads);
/* scorePoints (scan_matcher_ndt.cpp:156-178) and scoreScan (:151-154). */
double orc_matcher_score_points(const orc_matcher * m, const double * points_xy,
                                size_t n_points, const double * pose_xyt);
double orc_matcher_score_scan(const orc_matcher * m, const double * scan_pose_xyt,
                              const double * points_xy, size_t n_points);
void orc_matcher_reset(orc_matcher * m);                         /* :180-183 */
int orc_matcher_has_ndt(const orc_matcher * m);
const orc_ndt * orc_matcher_ndt(const orc_matcher * m);

/* ParticleFilter::measure, the per-particle loop (particle_filter.cpp:81-87):
 * weights[i] = scorePoints(points, particle_i).  The per-particle copy of the
 * point vector the reference makes (scan.cpp:67-70) is reproduced when
 * copy_points != 0 (timing faithfulness only; no effect on values). */
void orc_pf_measure(const orc_matcher * m, const double * particles_xyt, size_t n_particles,
                    const double * points_xy, size_t n_points, double * weights_out,
                    int copy_points);
void orc_pf_measure_omp(const orc_matcher * m, const double * particles_xyt,
                        size_t n_particles, const double * points_xy, size_t n_points,
                        double * weights_out, int n_threads);
/* ParticleFilter::updateStatistics (particle_filter.cpp:163-218).  weights are
 * normalised in place; mean_out[3]; cov_inout[9] row-major -- (2,2) accumulates
 * onto its previous value as in the reference (:216). */
void orc_pf_update_statistics(const double * particles_xyt, double * weights,
                              size_t n_particles, double * mean_out, double * cov_inout);

/* MotionModel::sample (src/motion_model.cpp:45-83) with the three
 * std::normal_distribution<float> draws per pose replaced by given standard
 * normals z[3i..3i+2] (the reference seeds its mt19937 from random_device, so
 * its draws are not reproducible): r = z * (float)sigma + (float)mean in float,
 * as libstdc++'s normal_distribution<float> computes it.  params_out (optional)
 * receives {rot1, trans, rot2, sigma_rot1, sigma_trans, sigma_rot2}. */
void orc_motion_sample(double dx, double dy, double dth, const double * alphas5,
                       double * poses_xyt, size_t n, const float * z, double * params_out);

/* ParticleFilter::init sampling loop (src/particle_filter.cpp:53-69), same
 * treatment of the draws (x, y, theta per particle, in that order). */
void orc_pf_init(double x, double y, double theta, double sigma_x, double sigma_y,
                 double sigma_theta, double * poses_xyt, size_t n, const float * z);

/* LaserScan -> Scan points, the conversion loop of NdtMapper::laserCallback
 * (src/ndt_mapper.cpp:385-453).  The message fields keep their ROS types
 * (sensor_msgs/LaserScan: float32 ranges, angle_min, angle_increment);
 * motion_* is `translation` (:386-389), the odometry motion between the start
 * and the end of the sweep; laser_* is laser_transform_.  Writes the kept
 * points (NaN / beyond range_max dropped, :413,436) in the reference's order
 * (descending index, index 0 never visited, when inverted: :410) and returns
 * their number. */
typedef struct orc_laser_scan
{
  float angle_min, angle_increment;
  double range_max;
  int inverted;
  double laser_x, laser_y, laser_theta;
  double motion_x, motion_y, motion_theta;
} orc_laser_scan;
size_t orc_convert_scan(const float * ranges, size_t n_ranges, const orc_laser_scan * scan,
                        double * points_xy_out);

/* OccupancyGrid (src/occupancy_grid.cpp).  Scans are given as for
 * orc_matcher_add_scans.  bounds = {min_x_, max_x_, min_y_, max_y_}, all 0 in a
 * fresh generator (:37-40).
 * orc_occupancy_update_bounds = updateBounds (:154-185) over scans
 * [first_scan, n_scans), including the final floor/ceil to the resolution.
 * orc_occupancy_render = the rest of getMsg (:56-151): info = {width, height} and
 * origin = {origin_x, origin_y} (:61-65), data[width*height] = -1 / 0 / 100.
 * With data == NULL only info and origin are produced.  A ray cell outside the
 * grid (the reference would write out of bounds: the bounds cover the scans'
 * points, not their poses) is skipped. */
void orc_occupancy_update_bounds(double * bounds, double resolution, const double * poses_xyt,
                                 const double * points_xy, const size_t * offsets,
                                 size_t first_scan, size_t n_scans);
void orc_occupancy_render(const double * bounds, double resolution, double occ_thresh,
                          const double * poses_xyt, const double * points_xy,
                          const size_t * offsets, size_t n_scans, uint32_t * info_wh,
                          double * origin_xy, signed char * data);

/* ROS angles (restated): used by updateStatistics and the motion model. */
double orc_normalize_angle(double a);
double orc_shortest_angular_distance(double from, double to);

#ifdef __cplusplus
}
#endif

#endif  /* NDT2D_ORACLE_H_ */
