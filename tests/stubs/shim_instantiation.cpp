// Syntax check of the shim against the reference's real interface headers.
#include <Eigen/Core>
#include <memory>
#include <vector>
#include "scan_matcher_ndt_hip.hpp"

void instantiate(const ndt_2d::ScanMatcherPtr & matcher, const ndt_2d::ScanPtr & scan)
{
  std::vector<Eigen::Vector3d> particles(4);
  std::vector<double> weights;
  ndt_2d_hip::measure_batched(matcher, scan, particles, weights);
  std::shared_ptr<ndt_2d::ScanMatcher> m = std::make_shared<ndt_2d_hip::ScanMatcherNDTHip>();
  (void)m;
}
