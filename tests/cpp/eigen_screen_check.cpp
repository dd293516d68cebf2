// Property check of ndt2d::clamp_test_surely_false (csrc/ndt2d_eigen2.h): wherever it says the
// test `small < 0.001 * large` of Cell::compute (reference src/ndt_model.cpp:88) is false, BOTH
// eigenvalue forms agree -- on covariances spread over 12 decades of conditioning, with the mass
// around the threshold, on exactly singular / negative / NaN / huge / tiny inputs.
//   g++ -O2 -std=c++17 -ffp-contract=off -I ndt_2d_amd/csrc tests/cpp/eigen_screen_check.cpp -o check && ./check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "ndt2d_eigen2.h"

static uint64_t state = 0x9e3779b97f4a7c15ull;
static double uniform()
{
  state += 0x9e3779b97f4a7c15ull;
  uint64_t z = state;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char ** argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 4000000;
  long screened = 0, clamped = 0, bad = 0, near = 0;
  double least_ratio_screened = 1.0;
  for (long i = 0; i < n; ++i)
  {
    // eigenvalues l2 >= l1 with ratio r, rotated by phi
    const double l2 = std::pow(10.0, -8.0 + 12.0 * uniform());
    double r;
    const double pick = uniform();
    if (pick < 0.35) r = 0.001 * std::pow(10.0, -1.0 + 2.0 * uniform());        // a decade either side of the threshold
    else if (pick < 0.55) r = 0.0011024 * (1.0 + 0.02 * (uniform() - 0.5));      // around the screen's own limit (q = 0.0011)
    else if (pick < 0.75) r = 0.001 + 0.00012 * uniform();                       // the band between threshold and limit, and just above
    else r = std::pow(10.0, -12.0 * uniform());
    const double l1 = l2 * r, phi = 6.283185307179586 * uniform();
    const double c = std::cos(phi), s = std::sin(phi);
    double a = c * c * l2 + s * s * l1, d = s * s * l2 + c * c * l1, b = c * s * (l2 - l1);
    if (i % 1000 == 0) b = 0.0;
    if (i % 1000 == 1) { a = -a; }
    if (i % 1000 == 2) { d = std::nan(""); }
    if (i % 1000 == 3) { a *= 1e200; d *= 1e200; b *= 1e200; }
    if (i % 1000 == 4) { a *= 1e-200; d *= 1e-200; b *= 1e-200; }
    if (i % 1000 == 5) { b = std::sqrt(a * d); }
    const bool sure = ndt2d::clamp_test_surely_false(a, b, d);
    for (int form = 0; form < 2; ++form)
    {
      double small, large;
      ndt2d::covariance_eigenvalues(form, a, b, d, &small, &large);
      if (small > large) std::swap(small, large);
      const bool clamp = small < 0.001 * large;
      if (form == 0) clamped += clamp;
      if (sure && clamp) ++bad;
      if (sure && large > 0.0) least_ratio_screened = std::min(least_ratio_screened, small / large);
    }
    screened += sure;
    near += (r > 0.0005 && r < 0.002);
  }
  std::printf("{\"cases\": %ld, \"screened\": %ld, \"clamp_branch\": %ld, \"near_threshold\": %ld, "
              "\"violations\": %ld, \"least_ratio_screened\": %.6g}\n", n, screened, clamped, near, bad,
              least_ratio_screened);
  return bad == 0 ? 0 : 1;
}
