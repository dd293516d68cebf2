// pluginlib shim: ndt_2d::ScanMatcher (include/ndt_2d/scan_matcher.hpp:42-91)
// implemented over libndt2d_hip.so's matcher layer (include/ndt2d_hip.h).
// Select it in the unchanged node with
//   scan_matcher_type: "ndt_2d_hip::ScanMatcherNDTHip"   (src/ndt_mapper.cpp:91-92)
#ifndef NDT_2D_HIP__SCAN_MATCHER_NDT_HIP_HPP_
#define NDT_2D_HIP__SCAN_MATCHER_NDT_HIP_HPP_

#include <string>
#include <vector>

#include <rclcpp/rclcpp.hpp>
#include <ndt_2d/scan_matcher.hpp>

#include "batch_pose_scorer.hpp"
#include "ndt2d_hip.h"

namespace ndt_2d_hip
{

class ScanMatcherNDTHip : public ndt_2d::ScanMatcher, public BatchPoseScorer
{
public:
  ScanMatcherNDTHip() = default;
  virtual ~ScanMatcherNDTHip();

  // the six pure virtuals of ndt_2d::ScanMatcher
  void initialize(const std::string & name, rclcpp::Node * node, double range_max) override;
  void addScans(const std::vector<ndt_2d::ScanPtr>::const_iterator & begin,
                const std::vector<ndt_2d::ScanPtr>::const_iterator & end) override;
  double matchScan(const ndt_2d::ScanPtr & scan, ndt_2d::Pose2d & pose,
                   Eigen::Matrix3d & covariance) const override;
  double scoreScan(const ndt_2d::ScanPtr & scan) const override;
  double scorePoints(const std::vector<ndt_2d::Point> & points,
                     const ndt_2d::Pose2d & pose) const override;
  void reset() override;

  // BatchPoseScorer
  bool scorePoses(const std::vector<ndt_2d::Point> & points, const double * poses_xyt,
                  std::size_t n, double * scores) const override;

private:
  bool ok(int rc, const char * what) const;

  ndt2d_matcher * matcher_ = nullptr;
  rclcpp::Node * node_ = nullptr;
  std::string name_;
};

}  // namespace ndt_2d_hip

#endif  // NDT_2D_HIP__SCAN_MATCHER_NDT_HIP_HPP_
