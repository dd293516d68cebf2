// Where the host side of addScans goes, phase by phase (the matcher layer compiled into this program):
//   g++ -O3 -std=c++17 -ffp-contract=off -I include -I ndt_2d_amd/csrc experiments/host_build_phases.cpp -L ndt_2d_amd -lndt2d_hip -Wl,-rpath,$PWD/ndt_2d_amd -o experiments/bin/host_build_phases
//   experiments/bin/host_build_phases 1   (toy map)   |   0   (245 x 245 grid)
#include "../ndt_2d_amd/csrc/ndt2d_host.cpp"
#include <chrono>
#include <cstdio>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char ** argv)
{
  const int toy = argc > 1 ? atoi(argv[1]) : 1;
  const int eform = argc > 2 ? atoi(argv[2]) : ndt2d::kEigenFormSchur;
  const int N = 9, B = 720;
  static double poses[3 * N], pts[2 * B * N];
  static size_t off[N + 1];
  const ndt2d_world big = {95.0, 5.0, 0.25}, room = {4.0, 4.0, 0.25};
  const ndt2d_world * w = toy ? &room : &big;
  const double pitch = toy ? 0.25 : 0.5;
  const double cx = toy ? 0.0 : 1.0, cy = toy ? 0.0 : 0.5;
  size_t n = 0;
  for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i)
  {
    double * p = poses + 3 * n;
    p[0] = cx + pitch * i; p[1] = cy + pitch * j; p[2] = 0.0;
    ndt2d_synth_scan(w, p, B, 0.01, 77u + (unsigned)(10 * (j + 1) + (i + 1)), pts + 2 * B * n);
    off[n] = (size_t)B * n; ++n;
  }
  off[n] = (size_t)B * n;
  const double rmax = toy ? 4.75 : 30.0;
  std::unique_ptr<HostNdt> ndt;
  std::vector<uint32_t> idx; std::vector<double> c6;
  double t[4] = {0, 0, 0, 0};
  const int reps = 3000;
  for (int r = -100; r < reps; ++r)
  {
    const double a = now_us();
    double min_x = 1e300, max_x = -1e300, min_y = 1e300, max_y = -1e300;
    for (size_t k = 0; k < n; ++k) { min_x = std::min(poses[3*k] - rmax, min_x); max_x = std::max(poses[3*k] + rmax, max_x); min_y = std::min(poses[3*k+1] - rmax, min_y); max_y = std::max(poses[3*k+1] + rmax, max_y); }
    if (ndt) ndt->reset(0.25, max_x - min_x, max_y - min_y, min_x, min_y); else ndt.reset(new HostNdt(0.25, max_x - min_x, max_y - min_y, min_x, min_y));
    if (getenv("NOIL")) ndt->set_interleave(false);
    if (getenv("ILALL")) ndt->set_side_by_side_max_bytes(~size_t(0));
    const double b = now_us();
    for (size_t k = 0; k < n; ++k) ndt->add_scan(poses[3*k], poses[3*k+1], poses[3*k+2], pts + 2 * off[k], off[k+1] - off[k]);
    const double c = now_us();
    ndt->compute(eform);
    const double d = now_us();
    idx.resize(ndt->n_touched()); c6.resize(6 * ndt->n_touched()); ndt->sparse6(idx.data(), c6.data());
    const double e = now_us();
    if (r >= 0) { t[0] += b - a; t[1] += c - b; t[2] += d - c; t[3] += e - d; }
  }
  { unsigned long long hsh = 1469598103934665603ull; for (double d : c6) { unsigned long long b; memcpy(&b, &d, 8); hsh = (hsh ^ b) * 1099511628211ull; } printf("cells hash %016llx  ", hsh); }
  printf("toy=%d cells touched %zu of %zu: reset %.2f us, add_scan x%zu %.2f us, compute %.2f us, sparse6 %.2f us\n", toy, idx.size(), ndt->ncell(), t[0]/reps, n, t[1]/reps, t[2]/reps, t[3]/reps);
  return 0;
}
