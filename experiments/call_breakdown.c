/* Where a default matchScan call's ~31 us go, stage by stage, from a C host:
 *   prepare_search (subsample, tables, staging) | match_launch (host submit) | match_fetch (wait) | finish.
 *   gcc -O2 -std=c99 -I include experiments/call_breakdown.c -L ndt_2d_amd -lndt2d_hip -lm -Wl,-rpath,$PWD/ndt_2d_amd -o experiments/bin/call_breakdown */
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
  const double pause_us = argc > 1 ? atof(argv[1]) : 0.0;
  static double poses[3 * N], pts[2 * B * N], scan[2 * B];
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
  ndt2d_synth_scan(&world, truth, B, 0.01, 101u, scan);
  ndt2d_matcher * m;
  if (ndt2d_matcher_create(&m, 0) != NDT2D_OK) return 2;
  ndt2d_matcher_initialize(m, 0.25, 0.0025, 0.1, 0.005, 0.05, 100, 4.75);
  ndt2d_handle h = ndt2d_matcher_device(m);
  ndt2d_set_timing(h, 0);
  ndt2d_matcher_add_scans(m, poses, pts, off, N);
  static double t[4][REPS];
  for (int r = -200; r < REPS; ++r)
  {
    size_t n_th, n_lin, use;
    ndt2d_match_result res;
    double pose[3] = {0, 0, 0}, cov[9], score, rec[12];
    const double a = now_us();
    if (ndt2d_matcher_prepare_search(m, guess, scan, B, &n_th, &n_lin, &use) != NDT2D_OK) return 3;
    const double b = now_us();
    if (ndt2d_match_launch(h, 0, n_th, NULL, NULL) != NDT2D_OK) return 4;
    const double c = now_us();
    if (ndt2d_match_fetch(h, &res) != NDT2D_OK) return 5;
    const double d = now_us();
    rec[0] = res.best_score; rec[1] = (double)res.best_index; for (int i = 0; i < 10; ++i) rec[2 + i] = res.acc[i];
    ndt2d_matcher_finish_match(m, rec, pose, cov, &score);
    const double e = now_us();
    if (r >= 0) { t[0][r] = b - a; t[1][r] = c - b; t[2][r] = d - c; t[3][r] = e - d; }
    /* argv[1] = microseconds of host-side pause between calls: the flag is raised BEFORE the
     * kernel has ended for the command processor, and a call issued at once queues behind that */
    if (pause_us > 0.0) { const double until = now_us() + pause_us; while (now_us() < until) {} }
  }
  const char * name[4] = {"prepare_search", "match_launch (host submit)", "match_fetch (wait for the flag)", "finish_match"};
  double total = 0;
  for (int s = 0; s < 4; ++s) { qsort(t[s], REPS, sizeof(double), cmp); printf("%-34s median %6.2f us  p90 %6.2f\n", name[s], t[s][REPS / 2], t[s][REPS * 9 / 10]); total += t[s][REPS / 2]; }
  printf("sum of medians %.2f us (%s)\n", total, ndt2d_last_variant(h));
  ndt2d_matcher_destroy(m);
  return 0;
}
