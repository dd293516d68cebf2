// ndt_2d::OccupancyGrid with the ray tracing on the GPU.
//
// Same constructor arguments and the same persistent state as the reference's
// generator (include/ndt_2d/occupancy_grid.hpp:44-74, src/occupancy_grid.cpp:
// 34-42): resolution, occupancy threshold, the bounds min_x_/max_x_/min_y_/max_y_
// and num_scans_.  getMsg takes the scans as plain arrays (pose triple + point
// list per scan) and fills a plain struct with the fields of the
// nav_msgs/OccupancyGrid the reference publishes (:60-66, 134-150); a ROS build
// copies them into the message in four lines.  All map cells are written by
// ndt2d_occupancy_grid (include/ndt2d_hip.h); nothing here touches one.
#ifndef NDT_2D_HIP__OCCUPANCY_GRID_HIP_HPP_
#define NDT_2D_HIP__OCCUPANCY_GRID_HIP_HPP_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "ndt2d_hip.h"

namespace ndt_2d_hip
{

// One ndt_2d::Scan as the renderer needs it: getPose() and getPoints().
struct ScanView
{
  double pose[3];              // x, y, theta
  const double * points_xy;    // interleaved x, y in the scan's own frame
  std::size_t n_points;
};

struct OccupancyGridMsg
{
  double resolution = 0.0;
  std::uint32_t width = 0, height = 0;
  double origin_x = 0.0, origin_y = 0.0;
  std::vector<signed char> data;   // row-major, y * width + x; -1 / 0 / 100
};

class OccupancyGridHip
{
public:
  OccupancyGridHip(double resolution, double occ_thresh, ndt2d_handle device)
  : resolution_(resolution), occ_thresh_(occ_thresh), dev_(device)
  {
  }

  // src/occupancy_grid.cpp:44-152.  Returns false (and keeps last_error()) when a
  // device call fails; the reference has no failure path here.
  bool getMsg(const std::vector<ScanView> & scans, OccupancyGridMsg & grid)
  {
    poses_.clear();
    points_.clear();
    offsets_.assign(1, 0);
    for (const ScanView & s : scans)
    {
      poses_.insert(poses_.end(), s.pose, s.pose + 3);
      points_.insert(points_.end(), s.points_xy, s.points_xy + 2 * s.n_points);
      offsets_.push_back(offsets_.back() + s.n_points);
    }
    ndt2d_occupancy_info info;
    // first call: bounds (only when the scan count changed, :51-54) and meta data
    if (!call(scans.size(), num_scans_, &info, nullptr, 0)) return false;
    num_scans_ = scans.size();
    grid.resolution = info.resolution;
    grid.width = info.width;
    grid.height = info.height;
    grid.origin_x = info.origin_x;
    grid.origin_y = info.origin_y;
    grid.data.assign(static_cast<std::size_t>(info.width) * info.height, 0);
    if (grid.data.empty()) return true;
    return call(scans.size(), num_scans_, &info, grid.data.data(), grid.data.size());
  }

  const double * bounds() const { return bounds_; }   // min_x_, max_x_, min_y_, max_y_
  std::size_t num_scans() const { return num_scans_; }
  const std::string & last_error() const { return error_; }

private:
  bool call(std::size_t n_scans, std::size_t n_bounded, ndt2d_occupancy_info * info,
            signed char * data, std::size_t capacity)
  {
    const int rc = ndt2d_occupancy_grid(dev_, resolution_, occ_thresh_, poses_.data(),
                                        points_.data(), offsets_.data(), n_scans, n_bounded,
                                        bounds_, info, data, capacity);
    if (rc == NDT2D_OK) return true;
    error_ = std::string("ndt2d error ") + std::to_string(rc) + ": " + ndt2d_last_error(dev_);
    return false;
  }

  double resolution_, occ_thresh_;
  ndt2d_handle dev_;
  double bounds_[4] = {0.0, 0.0, 0.0, 0.0};
  std::size_t num_scans_ = 0;
  std::vector<double> poses_, points_;
  std::vector<std::size_t> offsets_;
  std::string error_;
};

}  // namespace ndt_2d_hip

#endif  // NDT_2D_HIP__OCCUPANCY_GRID_HIP_HPP_
