// pluginlib shim over libndt2d_hip.so; see scan_matcher_ndt_hip.hpp.
//
// ndt_2d::Point is {double x, y} (include/ndt_2d/point.hpp:35-51), so a
// std::vector<Point> is already the interleaved xy array the C-ABI takes.
// There is no CPU fallback: if the device call fails the method logs the error
// and returns the reference's "no map" value 0.0 with the outputs untouched
// (src/scan_matcher_ndt.cpp:80,159); nothing is thrown across the boundary.
#include "scan_matcher_ndt_hip.hpp"

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ndt_2d_hip
{

static_assert(sizeof(ndt_2d::Point) == 2 * sizeof(double), "Point must be {double x, y}");

ScanMatcherNDTHip::~ScanMatcherNDTHip()
{
  if (matcher_) ndt2d_matcher_destroy(matcher_);
}

bool ScanMatcherNDTHip::ok(int rc, const char * what) const
{
  if (rc == NDT2D_OK) return true;
  if (node_)
  {
    RCLCPP_ERROR(node_->get_logger(), "%s: %s failed (%d): %s", name_.c_str(), what, rc,
                 matcher_ ? ndt2d_matcher_last_error(matcher_) : "no device context");
  }
  return false;
}

void ScanMatcherNDTHip::initialize(const std::string & name, rclcpp::Node * node,
                                   double range_max)
{
  node_ = node;
  name_ = name;
  // the reference's six parameters, same names and defaults (src/scan_matcher_ndt.cpp:37-44)
  const double resolution = node->declare_parameter<double>(name + ".ndt_resolution", 0.25);
  const double angular_res =
    node->declare_parameter<double>(name + ".search_angular_resolution", 0.0025);
  const double angular_size = node->declare_parameter<double>(name + ".search_angular_size", 0.1);
  const double linear_res =
    node->declare_parameter<double>(name + ".search_linear_resolution", 0.005);
  const double linear_size = node->declare_parameter<double>(name + ".search_linear_size", 0.05);
  const int laser_max_beams = node->declare_parameter<int>(name + ".laser_max_beams", 100);
  // new, additive: which GPU this instance runs on ...
  const int device_id = node->declare_parameter<int>(name + ".device_id", 0);
  // ... or which GPUs: the instance then deals its searches (matchScan's theta steps) and pose
  // batches to all of them and exchanges the per-device records once per call
  // (ndt2d_matcher_create_multi).  The loop-closure matcher of an 8-GPU node:
  //   global_scan_matcher.device_ids: [0, 1, 2, 3, 4, 5, 6, 7]
  // -- reached by the unchanged `global_scan_matcher_->matchScan(...)` of src/ndt_mapper.cpp:643.
  const std::vector<int64_t> device_ids =
    node->declare_parameter<std::vector<int64_t>>(name + ".device_ids", std::vector<int64_t>());
  // "auto" | "rccl" (one all-reduce of the record table) | "host" (no collective)
  const std::string exchange = node->declare_parameter<std::string>(name + ".exchange", "auto");
  // Work below these stays on the first device (candidates x beams of a search; particles x
  // beams of a batch -- BatchPoseScorer::scorePoses / measurePoses, the batched counterpart of
  // src/particle_filter.cpp:78-89).  The defaults are the library's (1e9 / 2e8: the
  // 1,000,000-particle global localisation of BASELINE configs[4], 7.2e8, is sharded);
  // negative = keep the library's default.
  const double multi_min_units = node->declare_parameter<double>(name + ".multi_min_units", -1.0);
  const double multi_min_pose_units = node->declare_parameter<double>(name + ".multi_min_pose_units", -1.0);

  // the library found at run time must be the one this shim was compiled against: an older
  // libndt2d_hip.so on the loader's path would otherwise fail at some later symbol, not here
  if (ndt2d_abi_version() != NDT2D_ABI_VERSION)
  {
    RCLCPP_ERROR(node_->get_logger(), "%s: libndt2d_hip.so has ABI %d, this plugin was built against %d (%s)",
                 name_.c_str(), ndt2d_abi_version(), NDT2D_ABI_VERSION, ndt2d_build_info());
    matcher_ = nullptr;
    return;
  }
  std::vector<int> ids(device_ids.begin(), device_ids.end());
  if (ids.empty()) ids.push_back(device_id);
  if (!ok(ndt2d_matcher_create_multi(&matcher_, ids.data(), static_cast<int>(ids.size())),
          "ndt2d_matcher_create_multi"))
  {
    matcher_ = nullptr;
    return;
  }
  ok(ndt2d_matcher_set_exchange(matcher_, exchange.c_str()), "ndt2d_matcher_set_exchange");
  {
    double search_units = 0.0, pose_units = 0.0;
    ndt2d_matcher_get_multi_thresholds(matcher_, &search_units, &pose_units);
    if (multi_min_units >= 0.0) search_units = multi_min_units;
    if (multi_min_pose_units >= 0.0) pose_units = multi_min_pose_units;
    ok(ndt2d_matcher_set_multi_thresholds(matcher_, search_units, pose_units), "ndt2d_matcher_set_multi_thresholds");
  }
  // no per-launch timing events in production: they cost ~4.5 us of every call
  ndt2d_matcher_set_timing(matcher_, 0);
  ok(ndt2d_matcher_initialize(matcher_, resolution, angular_res, angular_size, linear_res,
                              linear_size, static_cast<std::size_t>(laser_max_beams), range_max),
     "ndt2d_matcher_initialize");
}

void ScanMatcherNDTHip::addScans(const std::vector<ndt_2d::ScanPtr>::const_iterator & begin,
                                 const std::vector<ndt_2d::ScanPtr>::const_iterator & end)
{
  if (!matcher_) return;
  std::vector<double> poses;
  std::vector<double> points;
  std::vector<std::size_t> offsets(1, 0);
  for (auto scan = begin; scan != end; ++scan)
  {
    const ndt_2d::Pose2d pose = (*scan)->getPose();
    poses.push_back(pose.x);
    poses.push_back(pose.y);
    poses.push_back(pose.theta);
    for (const auto & p : (*scan)->getPoints())
    {
      points.push_back(p.x);
      points.push_back(p.y);
    }
    offsets.push_back(points.size() / 2);
  }
  ok(ndt2d_matcher_add_scans(matcher_, poses.data(), points.data(), offsets.data(),
                             offsets.size() - 1),
     "ndt2d_matcher_add_scans");
}

double ScanMatcherNDTHip::matchScan(const ndt_2d::ScanPtr & scan, ndt_2d::Pose2d & pose,
                                    Eigen::Matrix3d & covariance) const
{
  if (!matcher_ || !ndt2d_matcher_has_ndt(matcher_)) return 0.0;
  const ndt_2d::Pose2d scan_pose = scan->getPose();
  const std::vector<ndt_2d::Point> points = scan->getPoints();
  const double sp[3] = {scan_pose.x, scan_pose.y, scan_pose.theta};
  double pose_io[3] = {pose.x, pose.y, pose.theta};
  double cov[9];
  double score = 0.0;
  if (!ok(ndt2d_matcher_match_scan(matcher_, sp, reinterpret_cast<const double *>(points.data()),
                                   points.size(), pose_io, cov, &score),
          "ndt2d_matcher_match_scan"))
  {
    return 0.0;
  }
  pose.x = pose_io[0];
  pose.y = pose_io[1];
  pose.theta = pose_io[2];
  for (int r = 0; r < 3; ++r)
  {
    for (int c = 0; c < 3; ++c) covariance(r, c) = cov[r * 3 + c];
  }
  return score;
}

double ScanMatcherNDTHip::scoreScan(const ndt_2d::ScanPtr & scan) const
{
  // scorePoints(scan->getPoints(), scan->getPose()) (reference src/scan_matcher_ndt.cpp:151-154)
  // through its own entry point: the mapper's next call is matchScan(scan, ...)
  // (src/ndt_mapper.cpp:514-515), and the library queues that search behind this call's kernel
  // once it has seen the pair (include/ndt2d_hip.h, ndt2d_matcher_score_scan)
  if (!matcher_) return 0.0;
  const ndt_2d::Pose2d scan_pose = scan->getPose();
  const std::vector<ndt_2d::Point> points = scan->getPoints();
  const double p[3] = {scan_pose.x, scan_pose.y, scan_pose.theta};
  double score = 0.0;
  if (!ok(ndt2d_matcher_score_scan(matcher_, p, reinterpret_cast<const double *>(points.data()),
                                   points.size(), &score),
          "ndt2d_matcher_score_scan"))
  {
    return 0.0;
  }
  return score;
}

double ScanMatcherNDTHip::scorePoints(const std::vector<ndt_2d::Point> & points,
                                      const ndt_2d::Pose2d & pose) const
{
  if (!matcher_) return 0.0;
  const double p[3] = {pose.x, pose.y, pose.theta};
  double score = 0.0;
  if (!ok(ndt2d_matcher_score_points(matcher_, reinterpret_cast<const double *>(points.data()),
                                     points.size(), p, &score),
          "ndt2d_matcher_score_points"))
  {
    return 0.0;
  }
  return score;
}

void ScanMatcherNDTHip::reset()
{
  if (matcher_) ok(ndt2d_matcher_reset(matcher_), "ndt2d_matcher_reset");
}

bool ScanMatcherNDTHip::scorePoses(const std::vector<ndt_2d::Point> & points,
                                   const double * poses_xyt, std::size_t n,
                                   double * scores) const
{
  if (!matcher_) return false;
  return ok(ndt2d_matcher_score_poses(matcher_, reinterpret_cast<const double *>(points.data()),
                                      points.size(), poses_xyt, n, scores),
            "ndt2d_matcher_score_poses");
}

}  // namespace ndt_2d_hip

#include <pluginlib/class_list_macros.hpp>
PLUGINLIB_EXPORT_CLASS(ndt_2d_hip::ScanMatcherNDTHip, ndt_2d::ScanMatcher)
