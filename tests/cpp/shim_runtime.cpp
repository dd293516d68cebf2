// The pluginlib shim (ndt_2d_amd/plugin/scan_matcher_ndt_hip.cpp) RUN, not only parsed: a
// C++ program that holds the plugin object as the node does -- through ndt_2d::ScanMatcherPtr,
// with the reference's own Scan / Pose2d / Point classes (include/ndt_2d/*.hpp, src/scan.cpp
// compiled from where they lie) -- and calls the six virtuals in the node's order
// (src/ndt_mapper.cpp:299-312,508-515,634-643; src/particle_filter.cpp:81-87).  Eigen3, rclcpp and
// pluginlib are absent in this image: tests/stubs/ stands in for those three (a Matrix3d with
// operator(), a Node that hands out parameters, a no-op registration macro).
//
//   shim_runtime <input file> <output file> [name.param=value ...]
//
// input  : u64 n_scans | poses f64[n][3] | offsets u64[n+1] | points f64[total][2] |
//          query pose f64[3] | u64 n_query | query points f64[n_query][2] | u64 n_poses | poses f64[n][3]
// output : f64 scoreScan | f64 matchScan | pose f64[3] | cov f64[9] (row-major) | f64 scorePoints[n_poses]
//          | f64 batch[n_poses] | f64 matchScan after reset (0.0) | u64 device count
#include <Eigen/Core>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "scan_matcher_ndt_hip.hpp"

namespace
{
template<typename T>
std::vector<T> read_n(std::ifstream & f, size_t n)
{
  std::vector<T> v(n);
  f.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
  return v;
}
}  // namespace

int main(int argc, char ** argv)
{
  if (argc < 3) return 64;
  std::ifstream in(argv[1], std::ios::binary);
  if (!in) return 65;
  const uint64_t n_scans = read_n<uint64_t>(in, 1)[0];
  const auto poses = read_n<double>(in, 3 * n_scans);
  const auto offsets = read_n<uint64_t>(in, n_scans + 1);
  const auto points = read_n<double>(in, 2 * offsets[n_scans]);
  const auto qpose = read_n<double>(in, 3);
  const uint64_t n_query = read_n<uint64_t>(in, 1)[0];
  const auto qpts = read_n<double>(in, 2 * n_query);
  const uint64_t n_poses = read_n<uint64_t>(in, 1)[0];
  const auto pf = read_n<double>(in, 3 * n_poses);
  if (!in) return 66;

  rclcpp::Node node;
  std::string name = "global_scan_matcher";
  for (int i = 3; i < argc; ++i)
  {
    const std::string kv = argv[i];
    const size_t eq = kv.find('=');
    if (eq == std::string::npos) return 67;
    node.overrides[kv.substr(0, eq)] = kv.substr(eq + 1);
  }
  const double range_max = node.overrides.count("range_max") ? std::atof(node.overrides["range_max"].c_str()) : 4.75;

  // as NdtMapper does: create, initialize, addScans (src/ndt_mapper.cpp:299-312)
  ndt_2d::ScanMatcherPtr matcher = std::make_shared<ndt_2d_hip::ScanMatcherNDTHip>();
  matcher->initialize(name, &node, range_max);
  std::vector<ndt_2d::ScanPtr> scans;
  for (uint64_t k = 0; k < n_scans; ++k)
  {
    auto scan = std::make_shared<ndt_2d::Scan>(k);
    scan->setPose(ndt_2d::Pose2d(poses[3 * k], poses[3 * k + 1], poses[3 * k + 2]));
    std::vector<ndt_2d::Point> pts;
    for (uint64_t j = offsets[k]; j < offsets[k + 1]; ++j) pts.emplace_back(points[2 * j], points[2 * j + 1]);
    scan->setPoints(pts);
    scans.push_back(scan);
  }
  auto query = std::make_shared<ndt_2d::Scan>(n_scans);
  query->setPose(ndt_2d::Pose2d(qpose[0], qpose[1], qpose[2]));
  {
    std::vector<ndt_2d::Point> pts;
    for (uint64_t j = 0; j < n_query; ++j) pts.emplace_back(qpts[2 * j], qpts[2 * j + 1]);
    query->setPoints(pts);
  }

  std::vector<double> out;
  // no map yet: 0.0, outputs untouched (src/scan_matcher_ndt.cpp:80)
  {
    ndt_2d::Pose2d p;
    Eigen::Matrix3d c;
    c(0, 0) = 42.0;
    if (matcher->matchScan(query, p, c) != 0.0 || c(0, 0) != 42.0 || matcher->scoreScan(query) != 0.0) return 70;
  }
  matcher->reset();                                       // :508
  matcher->addScans(scans.begin(), scans.end());          // :509
  out.push_back(matcher->scoreScan(query));               // :514
  ndt_2d::Pose2d correction;                              // :512 (0, 0, 0)
  Eigen::Matrix3d covariance;
  out.push_back(matcher->matchScan(query, correction, covariance));   // :515
  out.push_back(correction.x);
  out.push_back(correction.y);
  out.push_back(correction.theta);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out.push_back(covariance(r, c));
  // ParticleFilter::measure's loop, unchanged (src/particle_filter.cpp:81-87)
  const std::vector<ndt_2d::Point> qp = query->getPoints();
  for (uint64_t i = 0; i < n_poses; ++i)
  {
    out.push_back(matcher->scorePoints(qp, ndt_2d::Pose2d(pf[3 * i], pf[3 * i + 1], pf[3 * i + 2])));
  }
  // ... and through the additive batched interface
  {
    std::vector<Eigen::Vector3d> particles(n_poses);
    for (uint64_t i = 0; i < n_poses; ++i) for (int d = 0; d < 3; ++d) particles[i](d) = pf[3 * i + d];
    std::vector<double> weights;
    ndt_2d_hip::measure_batched(matcher, query, particles, weights);
    out.insert(out.end(), weights.begin(), weights.end());
  }
  matcher->reset();
  {
    ndt_2d::Pose2d p;
    Eigen::Matrix3d c;
    out.push_back(matcher->matchScan(query, p, c));
  }
  std::ofstream of(argv[2], std::ios::binary);
  of.write(reinterpret_cast<const char *>(out.data()), static_cast<std::streamsize>(out.size() * sizeof(double)));
  matcher.reset();   // the node resets its instances before the loader dies (src/ndt_mapper.cpp:150-152)
  std::printf("ok %zu values\n", out.size());
  return 0;
}
