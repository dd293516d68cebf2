// ndt_2d::ParticleFilter with the particle set resident in GPU memory.
//
// Same constructor arguments and methods as the reference's filter
// (include/ndt_2d/particle_filter.hpp:49-115, src/particle_filter.cpp), over the
// C-ABI of libndt2d_hip.so (include/ndt2d_hip.h): init / update / measure and
// every updateStatistics run as HIP kernels on one device array that lives as
// long as the filter; resample keeps the reference's sequential KLD stopping rule
// on the host (src/particle_filter.cpp:91-140).
//
// Differences a maintainer should know:
//  * The reference draws its noise from an mt19937 seeded by std::random_device
//    (motion_model.hpp:63-64); here the draws come from the device's counter-based
//    stream, keyed by an explicit `seed` and one step per init / update call, so
//    runs are reproducible.  resample() keeps an mt19937 (std::discrete_distribution
//    over the weights, :94), seeded from the same `seed`.
//  * The KD-tree of the reference (kd_tree.hpp) is only asked for its leaf count
//    (:118); the library's ndt2d_kld_resample counts the distinct discrete keys
//    (kd_tree.hpp:95-98), which is the same number.
//  * No Eigen in this header: getMean / getCovariance fill plain arrays (row-major
//    3 x 3), so that it builds without ROS; scan_matcher_ndt_hip.cpp style adaptors
//    to Eigen::Vector3d / Matrix3d are two lines.
//
// Error behaviour follows the reference: no exceptions; ok() turns false on the
// first failed device call and last_error() tells why.
#ifndef NDT_2D_HIP__PARTICLE_FILTER_HIP_HPP_
#define NDT_2D_HIP__PARTICLE_FILTER_HIP_HPP_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <random>
#include <string>
#include <vector>

#include "ndt2d_hip.h"

namespace ndt_2d_hip
{

class ParticleFilterHip
{
public:
  // MotionModel(a1..a5) (include/ndt_2d/motion_model.hpp:52) folded in as alphas5.
  ParticleFilterHip(std::size_t min_particles, std::size_t max_particles, const double * alphas5,
                    ndt2d_matcher * matcher, std::uint64_t seed = 0)
  : matcher_(matcher),
    dev_(ndt2d_matcher_device(matcher)),
    min_particles_(min_particles),
    max_particles_(max_particles),
    seed_(seed),
    gen_(static_cast<std::mt19937::result_type>(seed))
  {
    std::copy(alphas5, alphas5 + 5, alphas_);
    // particles_.assign(min_particles_, (0, 0, 0)); weights_ = 1 / min (src/particle_filter.cpp:48-50)
    resize(min_particles_);
    std::vector<double> zeros(3 * n_, 0.0), w(n_, n_ ? 1.0 / static_cast<double>(n_) : 0.0);
    check(ndt2d_copy_to_device(dev_, d_particles_, zeros.data(), zeros.size() * sizeof(double)));
    check(ndt2d_copy_to_device(dev_, d_weights_, w.data(), w.size() * sizeof(double)));
    updateStatistics(false);
  }

  ~ParticleFilterHip()
  {
    ndt2d_device_free(dev_, d_particles_);
    ndt2d_device_free(dev_, d_weights_);
    ndt2d_device_free(dev_, d_stats_);
  }

  ParticleFilterHip(const ParticleFilterHip &) = delete;
  ParticleFilterHip & operator=(const ParticleFilterHip &) = delete;

  // src/particle_filter.cpp:53-69
  void init(double x, double y, double theta, double sigma_x, double sigma_y, double sigma_theta)
  {
    check(ndt2d_pf_init_launch(dev_, d_particles_, n_, x, y, theta, sigma_x, sigma_y, sigma_theta,
                               nullptr, seed_, ++step_, 0));
    std::vector<double> w(n_, 1.0 / static_cast<double>(n_));
    check(ndt2d_copy_to_device(dev_, d_weights_, w.data(), w.size() * sizeof(double)));
    updateStatistics(false);
  }

  // src/particle_filter.cpp:71-76
  void update(double dx, double dy, double dth)
  {
    check(ndt2d_pf_motion_launch(dev_, d_particles_, n_, dx, dy, dth, alphas_, nullptr, seed_,
                                 ++step_, 0));
    updateStatistics(false);
  }

  // src/particle_filter.cpp:78-89; points = scan->getPoints(), interleaved x, y
  void measure(const double * points_xy, std::size_t n_points)
  {
    std::size_t n_beams = 0;
    check(ndt2d_matcher_prepare_beams(matcher_, points_xy, n_points, &n_beams));
    if (!ok_) return;
    if (!ndt2d_matcher_has_ndt(matcher_) || n_beams == 0)
    {
      // scorePoints returns 0.0 without a map (src/scan_matcher_ndt.cpp:159), 0.0 / 0 without points
      std::vector<double> w(n_, n_beams == 0 && ndt2d_matcher_has_ndt(matcher_) ? std::nan("") : 0.0);
      check(ndt2d_copy_to_device(dev_, d_weights_, w.data(), w.size() * sizeof(double)));
      updateStatistics(false);
      return;
    }
    check(ndt2d_score_poses_launch(dev_, d_particles_, n_, d_weights_, d_stats_));
    updateStatistics(true);
  }

  // src/particle_filter.cpp:91-140
  void resample(double kld_err, double kld_z)
  {
    std::vector<double> particles(3 * n_), weights(n_);
    check(ndt2d_copy_to_host(dev_, particles.data(), d_particles_, particles.size() * sizeof(double)));
    check(ndt2d_copy_to_host(dev_, weights.data(), d_weights_, weights.size() * sizeof(double)));
    if (!ok_) return;
    // the draws of std::discrete_distribution (:94,110) as uniforms from the same
    // mt19937; the draw-and-stop loop (:107-133) is the library's host code
    std::vector<double> uniforms(max_particles_);
    std::uniform_real_distribution<double> unit(0.0, 1.0);
    for (double & u : uniforms) u = unit(gen_);
    // KD-tree leaf size 0.5 x 0.5 x 0.2671 (particle_filter.cpp:44)
    const double leaf[3] = {0.5, 0.5, 0.2671};
    std::vector<std::uint32_t> chosen(std::max<std::size_t>(max_particles_, 1));
    std::size_t n_kept = 0;
    check(ndt2d_kld_resample(particles.data(), weights.data(), n_, min_particles_, max_particles_,
                             kld_err, kld_z, leaf, uniforms.data(), uniforms.size(), chosen.data(),
                             &n_kept));
    if (!ok_) return;
    std::vector<double> resampled, resampled_weights;
    resampled.reserve(3 * n_kept);
    resampled_weights.reserve(n_kept);
    for (std::size_t i = 0; i < n_kept; ++i)
    {
      const std::size_t p = chosen[i];
      resampled.insert(resampled.end(), particles.begin() + 3 * p, particles.begin() + 3 * p + 3);
      resampled_weights.push_back(weights[p]);
    }
    resize(resampled_weights.size());
    check(ndt2d_copy_to_device(dev_, d_particles_, resampled.data(), resampled.size() * sizeof(double)));
    check(ndt2d_copy_to_device(dev_, d_weights_, resampled_weights.data(),
                               resampled_weights.size() * sizeof(double)));
    updateStatistics(false);
  }

  void getMean(double * mean3) const { std::copy(mean_, mean_ + 3, mean3); }
  void getCovariance(double * cov9) const { std::copy(cov_, cov_ + 9, cov9); }

  std::size_t size() const { return n_; }
  // Host copy of the particles ({x, y, theta} triples), e.g. for getMsg (:152-161).
  std::vector<double> particles()
  {
    std::vector<double> out(3 * n_);
    check(ndt2d_copy_to_host(dev_, out.data(), d_particles_, out.size() * sizeof(double)));
    return out;
  }
  std::vector<double> weights()
  {
    std::vector<double> out(n_);
    check(ndt2d_copy_to_host(dev_, out.data(), d_weights_, out.size() * sizeof(double)));
    return out;
  }
  std::uint64_t step() const { return step_; }

  bool ok() const { return ok_; }
  const std::string & last_error() const { return error_; }

private:
  void check(int rc)
  {
    if (rc != NDT2D_OK && ok_)
    {
      ok_ = false;
      error_ = std::string("ndt2d error ") + std::to_string(rc) + ": " + ndt2d_last_error(dev_);
    }
  }

  void resize(std::size_t n)
  {
    if (n > capacity_)
    {
      ndt2d_device_free(dev_, d_particles_);
      ndt2d_device_free(dev_, d_weights_);
      d_particles_ = d_weights_ = nullptr;
      capacity_ = std::max(n, max_particles_);
      void * p = nullptr;
      check(ndt2d_device_alloc(dev_, 3 * capacity_ * sizeof(double), &p));
      d_particles_ = static_cast<double *>(p);
      check(ndt2d_device_alloc(dev_, capacity_ * sizeof(double), &p));
      d_weights_ = static_cast<double *>(p);
    }
    if (d_stats_ == nullptr)
    {
      void * p = nullptr;
      check(ndt2d_device_alloc(dev_, (NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES) * sizeof(double), &p));
      d_stats_ = static_cast<double *>(p);
    }
    n_ = n;
  }

  // src/particle_filter.cpp:163-218 on the device; have_moments: the sums were just
  // written by ndt2d_score_poses_launch
  void updateStatistics(bool have_moments)
  {
    if (n_ == 0 || !ok_) return;
    if (!have_moments) check(ndt2d_pose_moments_launch(dev_, d_particles_, n_, d_weights_, d_stats_));
    double * d_out = d_stats_ + NDT2D_POSE_STATS_DOUBLES;
    check(ndt2d_pf_finalize_launch(dev_, d_particles_, n_, d_weights_, d_stats_, d_out));
    double out[NDT2D_PF_RESULT_DOUBLES];
    check(ndt2d_copy_to_host(dev_, out, d_out, sizeof(out)));
    if (!ok_) return;
    mean_[0] = out[1];
    mean_[1] = out[2];
    mean_[2] = out[3];
    cov_[0] = out[4];
    cov_[1] = cov_[3] = out[5];
    cov_[4] = out[6];
    cov_[8] += out[7];   // never zeroed by the reference either (:216)
  }

  ndt2d_matcher * matcher_;
  ndt2d_handle dev_;
  std::size_t min_particles_, max_particles_;
  double alphas_[5];
  std::uint64_t seed_, step_ = 0;
  std::mt19937 gen_;
  double * d_particles_ = nullptr, * d_weights_ = nullptr, * d_stats_ = nullptr;
  std::size_t n_ = 0, capacity_ = 0;
  double mean_[3] = {0.0, 0.0, 0.0};
  double cov_[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  bool ok_ = true;
  std::string error_;
};

}  // namespace ndt_2d_hip

#endif  // NDT_2D_HIP__PARTICLE_FILTER_HIP_HPP_
