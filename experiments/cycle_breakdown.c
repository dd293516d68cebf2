/* The mapper's per-scan cycle (reset + addScans + scoreScan + matchScan, reference
 * src/ndt_mapper.cpp:508-515) stage by stage from a C host, on the 245 x 245 grid of a real
 * lidar (the latency probe's second scenario).  Inside a cycle the first call that needs a
 * result waits for everything queued before it, so the stages are timed in place.
 *   gcc -O2 -std=c99 -I include experiments/cycle_breakdown.c -L ndt_2d_amd -lndt2d_hip -lm -Wl,-rpath,$PWD/ndt_2d_amd -o experiments/bin/cycle_breakdown
 *   argv[1] = cycles (default 2000) */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "ndt2d_hip.h"
static double now_us(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
static int cmp(const void * a, const void * b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
#define N 9
#define B 720
#define MAXR 4000
int main(int argc, char ** argv)
{
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  const int toy = argc > 2 ? atoi(argv[2]) : 0;
  static double poses[3 * N], pts[2 * B * N], scan[2 * B];
  static size_t off[N + 1];
  const ndt2d_world big = {95.0, 5.0, 0.25}, room = {4.0, 4.0, 0.25};
  const ndt2d_world * w = toy ? &room : &big;
  const double truth_big[3] = {1.0, 0.5, 0.3}, truth_toy[3] = {0.13, -0.07, 0.031};
  const double * truth = toy ? truth_toy : truth_big;
  const double pitch = toy ? 0.25 : 0.5;
  size_t n = 0;
  for (int j = -1; j <= 1; ++j) for (int i = -1; i <= 1; ++i)
  {
    double * p = poses + 3 * n;
    p[0] = (toy ? 0.0 : truth[0]) + pitch * i; p[1] = (toy ? 0.0 : truth[1]) + pitch * j; p[2] = 0.0;
    if (ndt2d_synth_pose_blocked(w, p[0], p[1], 0.25)) continue;
    ndt2d_synth_scan(w, p, B, 0.01, 77u + (unsigned)(10 * (j + 1) + (i + 1)), pts + 2 * B * n);
    off[n] = (size_t)B * n; ++n;
  }
  off[n] = (size_t)B * n;
  ndt2d_synth_scan(w, truth, B, 0.01, 501u, scan);
  const double guess[3] = {truth[0] + 0.02, truth[1] - 0.02, truth[2] + 0.01};
  ndt2d_matcher * m;
  if (ndt2d_matcher_create(&m, 0) != NDT2D_OK) return 2;
  ndt2d_matcher_initialize(m, 0.25, 0.0025, 0.1, 0.005, 0.05, 100, toy ? 4.75 : 30.0);
  ndt2d_set_timing(ndt2d_matcher_device(m), 0);
  static double t[5][MAXR];
  for (int r = -200; r < reps && r < MAXR; ++r)
  {
    double pose[3] = {0, 0, 0}, cov[9], score, s2;
    const double a = now_us();
    if (ndt2d_matcher_reset(m) != NDT2D_OK) return 3;
    const double b = now_us();
    if (ndt2d_matcher_add_scans(m, poses, pts, off, n) != NDT2D_OK) return 4;
    const double c = now_us();
    if (ndt2d_matcher_score_scan(m, guess, scan, B, &s2) != NDT2D_OK) return 5;
    const double d = now_us();
    if (ndt2d_matcher_match_scan(m, guess, scan, B, pose, cov, &score) != NDT2D_OK) return 6;
    const double e = now_us();
    if (r >= 0) { t[0][r] = b - a; t[1][r] = c - b; t[2][r] = d - c; t[3][r] = e - d; t[4][r] = e - a; }
  }
  const char * name[5] = {"reset", "addScans (host build + submit)", "scoreScan (waits for the grid)", "matchScan", "cycle"};
  for (int s = 0; s < 5; ++s) { qsort(t[s], reps, sizeof(double), cmp); printf("%-34s median %6.2f us  p90 %6.2f\n", name[s], t[s][reps / 2], t[s][reps * 9 / 10]); }
  ndt2d_matcher_destroy(m);
  return 0;
}
