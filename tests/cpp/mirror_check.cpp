// C++ consumer of the header-only mirrors in ndt_2d_amd/plugin/ (ParticleFilterHip,
// OccupancyGridHip) over the C-ABI -- no torch, no Python in the process.  Driven by
// tests/test_gpu_cpp_mirrors.py, which writes the input file, runs this program on
// the GPU box and compares the output file with the oracle and the Python mirrors.
//
//   mirror_check <input.bin> <output.bin>
//
// input  (8-byte little-endian words): n_scans, poses[3 n], offsets[n + 1],
//        points[2 total]; then min_particles, max_particles, seed (u64); alphas[5];
//        init[6] = x y theta sx sy sth; motion[3]; kld_err, kld_z; resolution,
//        occ_thresh (doubles).  The LAST scan is the one measured.
// output: n (u64), particles[3 n], weights[n], mean[3], cov[9] after measure;
//         m (u64), particles[3 m], weights[m], mean[3], cov[9] after resample;
//         width, height (u64), resolution, origin_x, origin_y, bounds[4], data[w h]
//         (one int8 per cell, padded to 8 bytes).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ndt2d_hip.h"
#include "occupancy_grid_hip.hpp"
#include "particle_filter_hip.hpp"

namespace
{

struct Reader
{
  std::FILE * f;
  std::uint64_t u64()
  {
    std::uint64_t v = 0;
    if (std::fread(&v, 8, 1, f) != 1) std::perror("read");
    return v;
  }
  double f64()
  {
    double v = 0;
    if (std::fread(&v, 8, 1, f) != 1) std::perror("read");
    return v;
  }
  std::vector<double> f64s(std::size_t n)
  {
    std::vector<double> v(n);
    if (n && std::fread(v.data(), 8, n, f) != n) std::perror("read");
    return v;
  }
};

struct Writer
{
  std::FILE * f;
  void u64(std::uint64_t v) { std::fwrite(&v, 8, 1, f); }
  void f64s(const double * p, std::size_t n) { std::fwrite(p, 8, n, f); }
  void f64s(const std::vector<double> & v) { f64s(v.data(), v.size()); }
};

void dump_filter(Writer & w, ndt_2d_hip::ParticleFilterHip & pf)
{
  double mean[3], cov[9];
  pf.getMean(mean);
  pf.getCovariance(cov);
  w.u64(pf.size());
  w.f64s(pf.particles());
  w.f64s(pf.weights());
  w.f64s(mean, 3);
  w.f64s(cov, 9);
}

}  // namespace

int main(int argc, char ** argv)
{
  if (argc != 3) return 2;
  Reader in{std::fopen(argv[1], "rb")};
  Writer out{std::fopen(argv[2], "wb")};
  if (!in.f || !out.f) return 2;

  const std::size_t n_scans = in.u64();
  std::vector<double> poses = in.f64s(3 * n_scans);
  std::vector<std::size_t> offsets(n_scans + 1);
  for (std::size_t & o : offsets) o = in.u64();
  std::vector<double> points = in.f64s(2 * offsets.back());
  const std::size_t min_particles = in.u64(), max_particles = in.u64();
  const std::uint64_t seed = in.u64();
  std::vector<double> alphas = in.f64s(5), init = in.f64s(6), motion = in.f64s(3);
  const double kld_err = in.f64(), kld_z = in.f64();
  const double resolution = in.f64(), occ_thresh = in.f64();

  ndt2d_matcher * matcher = nullptr;
  if (ndt2d_matcher_create(&matcher, 0) != NDT2D_OK)
  {
    std::fprintf(stderr, "no device\n");
    return 3;
  }
  int rc = ndt2d_matcher_initialize(matcher, 0.25, 0.0025, 0.1, 0.005, 0.05, 100, 30.0);
  // the map: every scan but the last
  if (rc == NDT2D_OK)
    rc = ndt2d_matcher_add_scans(matcher, poses.data(), points.data(), offsets.data(), n_scans - 1);
  if (rc != NDT2D_OK)
  {
    std::fprintf(stderr, "matcher: %s\n", ndt2d_matcher_last_error(matcher));
    return 4;
  }

  {
    ndt_2d_hip::ParticleFilterHip pf(min_particles, max_particles, alphas.data(), matcher, seed);
    pf.init(init[0], init[1], init[2], init[3], init[4], init[5]);
    pf.update(motion[0], motion[1], motion[2]);
    const std::size_t last = n_scans - 1;
    pf.measure(points.data() + 2 * offsets[last], offsets[last + 1] - offsets[last]);
    dump_filter(out, pf);
    pf.resample(kld_err, kld_z);
    dump_filter(out, pf);
    if (!pf.ok())
    {
      std::fprintf(stderr, "filter: %s\n", pf.last_error().c_str());
      return 5;
    }
  }

  {
    ndt_2d_hip::OccupancyGridHip grid(resolution, occ_thresh, ndt2d_matcher_device(matcher));
    std::vector<ndt_2d_hip::ScanView> scans(n_scans);
    for (std::size_t k = 0; k < n_scans; ++k)
    {
      std::memcpy(scans[k].pose, &poses[3 * k], sizeof(scans[k].pose));
      scans[k].points_xy = points.data() + 2 * offsets[k];
      scans[k].n_points = offsets[k + 1] - offsets[k];
    }
    ndt_2d_hip::OccupancyGridMsg msg;
    // twice, as the mapper's publish timer does: the second call keeps the bounds
    if (!grid.getMsg(scans, msg) || !grid.getMsg(scans, msg))
    {
      std::fprintf(stderr, "grid: %s\n", grid.last_error().c_str());
      return 6;
    }
    out.u64(msg.width);
    out.u64(msg.height);
    const double meta[3] = {msg.resolution, msg.origin_x, msg.origin_y};
    out.f64s(meta, 3);
    out.f64s(grid.bounds(), 4);
    std::vector<signed char> padded(msg.data);
    padded.resize((padded.size() + 7) / 8 * 8, 0);
    std::fwrite(padded.data(), 1, padded.size(), out.f);
  }

  ndt2d_matcher_destroy(matcher);
  std::fclose(in.f);
  std::fclose(out.f);
  return 0;
}
