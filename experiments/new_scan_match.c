/* matchScan on a scan the matcher has not seen (loop closure: reset, addScans, matchScan of a new
 * scan, reference src/ndt_mapper.cpp:634-643 -- no scoreScan before it): the beams travel with the
 * call.  Alternates two scans so that every call brings new beams; beside it the same call on the
 * scan of the call before (tables only).
 *   gcc -O2 -std=c99 -I include experiments/new_scan_match.c -L ndt_2d_amd -lndt2d_hip -lm -Wl,-rpath,$PWD/ndt_2d_amd -o experiments/bin/new_scan_match */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "ndt2d_hip.h"
static double now_us(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
static int cmp(const void * a, const void * b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
#define N 9
#define B 720
#define REPS 3000
int main(int argc, char ** argv)
{
  const size_t max_beams = argc > 1 ? (size_t)atoi(argv[1]) : 100;
  static double poses[3 * N], pts[2 * B * N], scan[2][2 * B];
  static size_t off[N + 1];
  const ndt2d_world world = {4.0, 4.0, 0.25};
  int k = 0;
  for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i, ++k)
  {
    poses[3 * k] = 0.25 * i; poses[3 * k + 1] = 0.25 * j; poses[3 * k + 2] = 0.0;
    ndt2d_synth_scan(&world, poses + 3 * k, B, 0.01, 1000003u + (unsigned)k, pts + 2 * B * k);
    off[k] = (size_t)B * k;
  }
  off[N] = (size_t)B * N;
  const double truth[3] = {0.13, -0.07, 0.031}, guess[3] = {0.11, -0.05, 0.02};
  ndt2d_synth_scan(&world, truth, B, 0.01, 101u, scan[0]);
  ndt2d_synth_scan(&world, truth, B, 0.01, 102u, scan[1]);
  ndt2d_matcher * m;
  if (ndt2d_matcher_create(&m, 0) != NDT2D_OK) return 2;
  ndt2d_matcher_initialize(m, 0.25, 0.0025, 0.1, 0.005, 0.05, max_beams, 4.75);
  ndt2d_set_timing(ndt2d_matcher_device(m), 0);
  ndt2d_matcher_add_scans(m, poses, pts, off, N);
  static double t[2][REPS];
  for (int mode = 0; mode < 2; ++mode)
  {
    for (int r = -200; r < REPS; ++r)
    {
      double pose[3] = {0, 0, 0}, cov[9], score;
      const double * s = scan[mode == 0 ? (r & 1) : 0];
      const double a = now_us();
      if (ndt2d_matcher_match_scan(m, guess, s, B, pose, cov, &score) != NDT2D_OK) return 3;
      if (r >= 0) t[mode][r] = now_us() - a;
    }
    qsort(t[mode], REPS, sizeof(double), cmp);
    printf("%zu beams, %s: matchScan median %.2f us  p90 %.2f  (%s)\n", max_beams,
           mode == 0 ? "a new scan every call" : "the scan of the call before", t[mode][REPS / 2],
           t[mode][REPS * 9 / 10], ndt2d_last_variant(ndt2d_matcher_device(m)));
  }
  ndt2d_matcher_destroy(m);
  return 0;
}
