// Additive batched-scoring interface for ndt_2d scan matchers.
//
// The reference's ParticleFilter::measure (src/particle_filter.cpp:78-89) calls
// ScanMatcher::scorePoints once per particle through a virtual call and copies
// the whole point vector per particle (src/scan.cpp:67-70).  A GPU launch per
// particle is hopeless, so matchers that can score many poses at once implement
// this mixin next to ndt_2d::ScanMatcher (include/ndt_2d/scan_matcher.hpp:42-91,
// which stays untouched), and measure_batched() below is the drop-in body for
// ParticleFilter::measure's loop.
#ifndef NDT_2D_HIP__BATCH_POSE_SCORER_HPP_
#define NDT_2D_HIP__BATCH_POSE_SCORER_HPP_

#include <cstddef>
#include <vector>

#include <ndt_2d/point.hpp>
#include <ndt_2d/pose_2d.hpp>
#include <ndt_2d/scan_matcher.hpp>

namespace ndt_2d_hip
{

class BatchPoseScorer
{
public:
  virtual ~BatchPoseScorer() = default;

  /**
   * @brief scores[i] = scorePoints(points, Pose2d(xyt[3i], xyt[3i+1], xyt[3i+2])).
   * @param points Points to score, robot frame.
   * @param poses_xyt n poses as {x, y, theta} triples (the memory layout of
   *        std::vector<Eigen::Vector3d>, i.e. ParticleFilter::particles_).
   * @param n Number of poses.
   * @param scores Output, n values.
   * @returns false if the batch could not be scored (scores untouched).
   */
  virtual bool scorePoses(const std::vector<ndt_2d::Point> & points, const double * poses_xyt,
                          std::size_t n, double * scores) const = 0;
};

/**
 * @brief Body of ParticleFilter::measure's loop (src/particle_filter.cpp:81-87):
 *        one launch when the matcher implements BatchPoseScorer, the
 *        reference's per-particle loop otherwise.
 */
template<typename ParticleVector>
void measure_batched(const ndt_2d::ScanMatcherPtr & matcher, const ndt_2d::ScanPtr & scan,
                     const ParticleVector & particles, std::vector<double> & weights)
{
  const std::vector<ndt_2d::Point> points = scan->getPoints();
  weights.resize(particles.size());
  const auto * batch = dynamic_cast<const BatchPoseScorer *>(matcher.get());
  if (batch && !particles.empty() &&
      batch->scorePoses(points, particles[0].data(), particles.size(), weights.data()))
  {
    return;
  }
  for (std::size_t i = 0; i < particles.size(); ++i)
  {
    ndt_2d::Pose2d pose(particles[i](0), particles[i](1), particles[i](2));
    weights[i] = matcher->scorePoints(points, pose);
  }
}

}  // namespace ndt_2d_hip

#endif  // NDT_2D_HIP__BATCH_POSE_SCORER_HPP_
