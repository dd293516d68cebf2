// Cell::addPoint's recurrence on the host: which instruction form the GPU box's CPU runs fastest.
//   g++ -O3 -march=native -ffp-contract=off experiments/host_addpoint_forms.cpp -o experiments/bin/host_addpoint_forms
// forms: ymm + scalar (the product's), zmm masked (five moments in one 512-bit divide), each as one chain
// per cell visited in the order of a scan (consecutive beams share cells) and as four chains side by side.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <immintrin.h>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
typedef double v4d __attribute__((vector_size(32)));
struct Cell { double valid = 0, n = 0, mx = 0, my = 0, cxx = 0, cxy = 0, cyy = 0, cov[3] = {0, 0, 0}, info[3] = {0, 0, 0}; };
static inline void add_ymm(Cell & c, double x, double y)
{
  const double n1 = c.n + 1;
  v4d v; std::memcpy(&v, &c.mx, 32);
  const v4d t = {x, y, x * x, x * y};
  v = (v * c.n + t) / n1;
  std::memcpy(&c.mx, &v, 32);
  c.cyy = (c.cyy * c.n + y * y) / n1;
  c.n = n1;
}
#ifdef __AVX512F__
static inline void add_zmm(Cell & c, double x, double y)
{
  const __m512d n = _mm512_set1_pd(c.n), n1 = _mm512_set1_pd(c.n + 1);
  __m512d v = _mm512_maskz_loadu_pd(0x1f, &c.mx);
  const __m512d t = _mm512_set_pd(0, 0, 0, y * y, x * y, x * x, y, x);
  v = _mm512_div_pd(_mm512_add_pd(_mm512_mul_pd(v, n), t), n1);
  _mm512_mask_storeu_pd(&c.mx, 0x1f, v);
  c.n = c.n + 1;
}
#endif
int main()
{
  const int N = 720, S = 9, G = 41;
  std::vector<double> xy(2 * N * S); std::vector<int> idx(N * S);
  for (int s = 0; s < S; ++s) for (int k = 0; k < N; ++k)
  {
    const double a = -M_PI + k * 2 * M_PI / N, c = cos(a), sn = sin(a);
    double r = std::min(4.0 / fabs(c), 4.0 / fabs(sn)) + 0.01 * ((rand() % 200) - 100) / 100.0;
    const double x = r * c + 0.05 * s, y = r * sn;
    xy[2 * (s * N + k)] = x; xy[2 * (s * N + k) + 1] = y;
    int gx = (int)((x + 5) * 4), gy = (int)((y + 5) * 4); if (gx < 0) gx = 0; if (gx >= G) gx = G - 1; if (gy < 0) gy = 0; if (gy >= G) gy = G - 1;
    idx[s * N + k] = gy * G + gx;
  }
  std::vector<Cell> cells(G * G + 4);
  const int reps = 4000;
  for (int form = 0; form < 4; ++form)
  {
#ifndef __AVX512F__
    if (form & 1) continue;
#endif
    double tot = 0, h = 0;
    for (int r = -200; r < reps; ++r)
    {
      for (auto & c : cells) c = Cell();
      const double t0 = now_us();
      for (int s = 0; s < S; ++s)
      {
        const double * p = &xy[2 * s * N]; const int * I = &idx[s * N];
        if (form < 2)
        {
          for (int k = 0; k < N; ++k)
          {
#ifdef __AVX512F__
            if (form == 1) add_zmm(cells[I[k]], p[2 * k], p[2 * k + 1]); else
#endif
            add_ymm(cells[I[k]], p[2 * k], p[2 * k + 1]);
          }
        }
        else
        {
          const int q = N / 4;   // (quarters side by side; boundary cells ignored here: timing only)
          for (int j = 0; j < q; ++j)
          {
            for (int u = 0; u < 4; ++u)
            {
              const int k = j + u * q;
#ifdef __AVX512F__
              if (form == 3) add_zmm(cells[I[k]], p[2 * k], p[2 * k + 1]); else
#endif
              add_ymm(cells[I[k]], p[2 * k], p[2 * k + 1]);
            }
          }
        }
      }
      const double t1 = now_us();
      if (r >= 0) tot += t1 - t0;
    }
    for (auto & c : cells) h += c.mx + c.cyy;
    const char * name[4] = {"ymm+scalar, in scan order", "zmm masked, in scan order", "ymm+scalar, four quarters side by side", "zmm masked, four quarters side by side"};
    printf("%-42s %7.2f us per 6480 points (%.3f ns per point)  checksum %.17g\n", name[form], tot / reps, tot / reps / 6.48, h);
  }
  return 0;
}
