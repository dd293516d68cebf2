/* Call latencies of the node's default workload through the C-ABI, without an
 * interpreter in the way: what the pluginlib shim (ndt_2d_amd/plugin/) pays per call.
 *
 * The plugin's default parameters (reference src/scan_matcher_ndt.cpp:37-44): 100 of 720
 * beams, 21 x 21 x 80 = 35,280 candidates, against the NDT of 9 scans of the synthetic
 * 8 x 8 m room (SURVEY.md 8d, cfg-1's map).  Per accepted scan the mapper runs
 * reset + addScans + scoreScan + matchScan (src/ndt_mapper.cpp:508-515); the unchanged
 * ParticleFilter::measure calls scorePoints once per particle (src/particle_filter.cpp:
 * 81-87).  Then the mapper's calls once more on a grid of a real lidar's size (245 x 245
 * cells).  Prints one JSON object with medians in microseconds.
 *
 *   gcc -O2 -std=c99 -I include ndt_2d_amd/tools/latency_probe.c -L ndt_2d_amd -lndt2d_hip -lm
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ndt2d_hip.h"

static double now_us(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static int cmp(const void * a, const void * b)
{
  const double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

#define N_SCANS 9
#define N_BEAMS 720
#define REPS 2000

static ndt2d_matcher * m;
static double map_poses[3 * N_SCANS], map_pts[2 * N_BEAMS * N_SCANS];
static size_t map_off[N_SCANS + 1];
static size_t n_map_scans = N_SCANS;
static double scan_pts[2 * N_BEAMS];
static double guess[3] = {0.11, -0.05, 0.02};
static double pf_poses[3 * 500];

static void op_match(void)
{
  double pose[3] = {0, 0, 0}, cov[9], score;
  if (ndt2d_matcher_match_scan(m, guess, scan_pts, N_BEAMS, pose, cov, &score) != NDT2D_OK) exit(3);
}
static void op_score_scan(void)
{
  double score;
  if (ndt2d_matcher_score_scan(m, guess, scan_pts, N_BEAMS, &score) != NDT2D_OK) exit(4);
}
static void op_add(void)
{
  if (ndt2d_matcher_reset(m) != NDT2D_OK) exit(5);
  if (ndt2d_matcher_add_scans(m, map_poses, map_pts, map_off, n_map_scans) != NDT2D_OK) exit(6);
}
static void op_cycle(void)
{
  op_add();
  op_score_scan();
  op_match();
}
static void op_pf_loop(void)
{
  for (int i = 0; i < 500; ++i)
  {
    double w;
    if (ndt2d_matcher_score_points(m, scan_pts, N_BEAMS, pf_poses + 3 * i, &w) != NDT2D_OK) exit(7);
  }
}
static void op_pf_measure(void)
{
  static double w[500], mean[3], cov[9];
  if (ndt2d_matcher_pf_measure(m, pf_poses, 500, scan_pts, N_BEAMS, w, mean, cov) != NDT2D_OK) exit(9);
}
static void op_pf_batch(void)
{
  static double w[500];
  if (ndt2d_matcher_score_poses(m, scan_pts, N_BEAMS, pf_poses, 500, w) != NDT2D_OK) exit(8);
}

/* between (optional): an untimed call after every timed one */
static void measure_between(void (*op)(void), void (*between)(void), int reps, double * median, double * p99)
{
  double * t = (double *)malloc(sizeof(double) * (size_t)reps);
  for (int i = 0; i < reps / 10 + 5; ++i)
  {
    op();
    if (between != NULL) between();
  }
  for (int i = 0; i < reps; ++i)
  {
    const double t0 = now_us();
    op();
    t[i] = now_us() - t0;
    if (between != NULL) between();
  }
  qsort(t, (size_t)reps, sizeof(double), cmp);
  *median = t[reps / 2];
  *p99 = t[(int)(reps * 0.99)];
  free(t);
}

static void measure(void (*op)(void), int reps, double * median, double * p99)
{
  measure_between(op, NULL, reps, median, p99);
}

/* --devices 0,1,..: the loop-closure search of cfg-4 (+-5 m / 0.02 m x +-pi /
 * 0.005 rad = 315,508,257 candidates x 720 beams) and cfg-2's (2,000,000 candidates) through ONE
 * multi-device matcher (ndt2d_matcher_create_multi) -- what the unchanged node's
 * global_scan_matcher_->matchScan() (reference src/ndt_mapper.cpp:634-643) costs with the
 * plugin's device_ids parameter set.  [--exchange auto|host|rccl].  One JSON object. */
static int multi_mode(const char * id_list, const char * exchange)
{
  int ids[64], n_dev = 0;
  char buf[256];
  snprintf(buf, sizeof(buf), "%s", id_list);
  for (char * tok = strtok(buf, ","); tok != NULL && n_dev < 64; tok = strtok(NULL, ",")) ids[n_dev++] = atoi(tok);
  if (n_dev == 0) return 64;
  const ndt2d_world world = {4.0, 4.0, 0.25};
  int k = 0;
  for (int j = -1; j <= 1; ++j)
  {
    for (int i = -1; i <= 1; ++i, ++k)
    {
      map_poses[3 * k] = 0.25 * i;
      map_poses[3 * k + 1] = 0.25 * j;
      map_poses[3 * k + 2] = 0.0;
      if (ndt2d_synth_scan(&world, map_poses + 3 * k, N_BEAMS, 0.01, 1000003u + (unsigned)k, map_pts + 2 * N_BEAMS * k) !=
          NDT2D_OK)
        return 1;
      map_off[k] = (size_t)N_BEAMS * (size_t)k;
    }
  }
  map_off[N_SCANS] = (size_t)N_BEAMS * N_SCANS;
  const double truth[3] = {0.13, -0.07, 0.031};
  if (ndt2d_synth_scan(&world, truth, N_BEAMS, 0.01, 101u, scan_pts) != NDT2D_OK) return 1;
  int rc = ndt2d_matcher_create_multi(&m, ids, n_dev);
  if (rc != NDT2D_OK)
  {
    fprintf(stderr, "ndt2d_matcher_create_multi -> %d (no GPU: there is no CPU fallback)\n", rc);
    return 2;
  }
  if (ndt2d_matcher_set_exchange(m, exchange) != NDT2D_OK) return 3;
  ndt2d_matcher_set_timing(m, 0);
  const double zero[3] = {0.0, 0.0, 0.0};
  const double cfgs[2][4] = {{1.0, 0.02, 0.5, 0.005}, {5.0, 0.02, 3.14159265358979323846, 0.005}};
  const char * names[2] = {"cfg2", "cfg4"};
  /* (one line at the end: RCCL prints its version banner to stdout when it is first used) */
  char out[2048];
  size_t len = (size_t)snprintf(out, sizeof(out), "{\"devices\": %d, \"exchange_requested\": \"%s\"", n_dev, exchange);
  for (int c = 0; c < 2; ++c)
  {
    if (ndt2d_matcher_initialize(m, 0.25, cfgs[c][3], cfgs[c][2], cfgs[c][1], cfgs[c][0], N_BEAMS, 4.75) != NDT2D_OK) return 4;
    if (ndt2d_matcher_reset(m) != NDT2D_OK || ndt2d_matcher_add_scans(m, map_poses, map_pts, map_off, N_SCANS) != NDT2D_OK) return 5;
    const int reps = c == 0 ? 40 : 7;
    double t[40], pose[3], cov[9], score = 0.0;
    size_t n_cand = 0;
    uint64_t best = 0;
    for (int r = -2; r < reps; ++r)
    {
      pose[0] = pose[1] = pose[2] = 0.0;
      const double t0 = now_us();
      if (ndt2d_matcher_match_scan_ex(m, zero, scan_pts, N_BEAMS, pose, cov, &score, NULL, 0, &n_cand, &best) != NDT2D_OK)
      {
        fprintf(stderr, "match_scan: %s\n", ndt2d_matcher_last_error(m));
        return 6;
      }
      if (r >= 0) t[r] = now_us() - t0;
    }
    qsort(t, (size_t)reps, sizeof(double), cmp);
    const double ms = t[reps / 2] * 1e-3;
    len += (size_t)snprintf(out + len, sizeof(out) - len,
                            ", \"%s\": {\"n_candidates\": %zu, \"best_index\": %llu, \"score\": %.17g, \"step_ms\": %.4f, "
                            "\"units_per_s\": %.4e, \"variant\": \"%s\"}",
                            names[c], n_cand, (unsigned long long)best, score, ms,
                            (double)n_cand * N_BEAMS / (ms * 1e-3), ndt2d_matcher_last_variant(m));
    if (len >= sizeof(out)) return 7;
  }
  printf("\n%s}\n", out);
  ndt2d_matcher_destroy(m);
  return 0;
}

int main(int argc, char ** argv)
{
  {
    const char * devices = NULL;
    const char * exchange = "auto";
    for (int i = 1; i + 1 < argc; ++i)
    {
      if (strcmp(argv[i], "--devices") == 0) devices = argv[i + 1];
      if (strcmp(argv[i], "--exchange") == 0) exchange = argv[i + 1];
    }
    if (devices != NULL) return multi_mode(devices, exchange);
  }
  /* cfg-1's world and map (ndt_2d_amd/synth.py): room 8 x 8 m, pillars at (+-2, +-2),
   * 9 map scans on a 3 x 3 lattice of pitch 0.25 m, seeds 1000003 + k; query scan from
   * (0.13, -0.07, 0.031), seed 101 */
  const ndt2d_world world = {4.0, 4.0, 0.25};
  int k = 0;
  for (int j = -1; j <= 1; ++j)       /* y outer, x inner, seed 1 * 1000003 + index */
  {
    for (int i = -1; i <= 1; ++i, ++k)
    {
      map_poses[3 * k] = 0.25 * i;
      map_poses[3 * k + 1] = 0.25 * j;
      map_poses[3 * k + 2] = 0.0;
      if (ndt2d_synth_scan(&world, map_poses + 3 * k, N_BEAMS, 0.01, 1000003u + (unsigned)k,
                           map_pts + 2 * N_BEAMS * k) != NDT2D_OK)
        return 1;
      map_off[k] = (size_t)N_BEAMS * (size_t)k;
    }
  }
  map_off[N_SCANS] = (size_t)N_BEAMS * N_SCANS;
  const double truth[3] = {0.13, -0.07, 0.031};
  if (ndt2d_synth_scan(&world, truth, N_BEAMS, 0.01, 101u, scan_pts) != NDT2D_OK) return 1;
  double u[1500];
  ndt2d_synth_uniform(303u, 1500, u);
  for (int i = 0; i < 500; ++i)
  {
    pf_poses[3 * i] = (u[3 * i] - 0.5) * 7.0;
    pf_poses[3 * i + 1] = (u[3 * i + 1] - 0.5) * 7.0;
    pf_poses[3 * i + 2] = (u[3 * i + 2] - 0.5) * 6.28;
  }

  int rc = ndt2d_matcher_create(&m, 0);
  if (rc != NDT2D_OK)
  {
    fprintf(stderr, "ndt2d_matcher_create -> %d (no GPU: there is no CPU fallback)\n", rc);
    return 2;
  }
  /* the plugin's defaults; range_max 4.75 as in cfg-1 */
  ndt2d_matcher_initialize(m, 0.25, 0.0025, 0.1, 0.005, 0.05, 100, 4.75);
  ndt2d_set_timing(ndt2d_matcher_device(m), 0);   /* as the shim does */
  op_add();

  double med[8], p99[8];
  measure(op_match, REPS, &med[0], &p99[0]);
  measure(op_score_scan, REPS, &med[1], &p99[1]);
  /* reset + addScans as the node issues it: a call that returns a result (scoreScan) follows
   * every one (untimed here).  Back to back -- nothing fetched in between -- each install first
   * has to ask the stream whether the one before has read the staging buffer. */
  measure_between(op_add, op_score_scan, REPS / 4, &med[2], &p99[2]);
  double add_b2b, add_b2b_p99;
  measure(op_add, REPS / 4, &add_b2b, &add_b2b_p99);
  measure(op_cycle, REPS / 4, &med[3], &p99[3]);
  measure(op_pf_loop, 20, &med[4], &p99[4]);
  measure(op_pf_batch, REPS / 4, &med[5], &p99[5]);
  measure(op_pf_measure, REPS / 4, &med[6], &p99[6]);
  /* the cycle once more with scoreScan NOT launching the scan's search ahead (the library does that
   * once it has seen the mapper's scoreScan / matchScan pair; ndt2d_matcher_set_search_ahead) */
  ndt2d_matcher_set_search_ahead(m, 0);
  measure(op_cycle, REPS / 4, &med[7], &p99[7]);
  ndt2d_matcher_set_search_ahead(m, 1);
  uint64_t ahead_launched = 0, ahead_collected = 0;
  ndt2d_matcher_search_ahead_stats(m, &ahead_launched, &ahead_collected);
  double pose[3] = {0, 0, 0}, cov[9], score;
  ndt2d_matcher_match_scan(m, guess, scan_pts, N_BEAMS, pose, cov, &score);
  char variant[128];
  snprintf(variant, sizeof(variant), "%s", ndt2d_last_variant(ndt2d_matcher_device(m)));
  ndt2d_matcher_destroy(m);

  /* The same calls on a grid of a real lidar's size: the local NDT spans the scan poses
   * +- range_max (reference src/scan_matcher_ndt.cpp:52-66), 245 x 245 cells for 30 m at
   * 0.25 m.  cfg-5's world (190 x 190 m room, pillars every 5 m), nine scans on a 3 x 3
   * lattice of pitch 0.5 m around the query pose (1.0, 0.5, 0.3). */
  const ndt2d_world big = {95.0, 5.0, 0.25};
  const double truth5[3] = {1.0, 0.5, 0.3};
  n_map_scans = 0;
  for (int j = -1; j <= 1; ++j)
  {
    for (int i = -1; i <= 1; ++i)
    {
      double * p = map_poses + 3 * n_map_scans;
      p[0] = truth5[0] + 0.5 * i;
      p[1] = truth5[1] + 0.5 * j;
      p[2] = 0.0;
      if (ndt2d_synth_pose_blocked(&big, p[0], p[1], 0.25)) continue;
      if (ndt2d_synth_scan(&big, p, N_BEAMS, 0.01, 77u + (unsigned)(10 * (j + 1) + (i + 1)),
                           map_pts + 2 * N_BEAMS * n_map_scans) != NDT2D_OK)
        return 1;
      map_off[n_map_scans] = (size_t)N_BEAMS * n_map_scans;
      ++n_map_scans;
    }
  }
  map_off[n_map_scans] = (size_t)N_BEAMS * n_map_scans;
  if (ndt2d_synth_scan(&big, truth5, N_BEAMS, 0.01, 501u, scan_pts) != NDT2D_OK) return 1;
  guess[0] = truth5[0] + 0.02;
  guess[1] = truth5[1] - 0.02;
  guess[2] = truth5[2] + 0.01;
  if (ndt2d_matcher_create(&m, 0) != NDT2D_OK) return 2;
  ndt2d_matcher_initialize(m, 0.25, 0.0025, 0.1, 0.005, 0.05, 100, 30.0);
  ndt2d_set_timing(ndt2d_matcher_device(m), 0);
  op_add();
  double rmed[5], rp99[5];
  measure(op_match, REPS, &rmed[0], &rp99[0]);
  measure(op_score_scan, REPS, &rmed[1], &rp99[1]);
  measure_between(op_add, op_score_scan, REPS / 4, &rmed[2], &rp99[2]);
  double radd_b2b, radd_b2b_p99;
  measure(op_add, REPS / 4, &radd_b2b, &radd_b2b_p99);
  measure(op_cycle, REPS / 4, &rmed[3], &rp99[3]);
  ndt2d_matcher_set_search_ahead(m, 0);
  measure(op_cycle, REPS / 4, &rmed[4], &rp99[4]);
  ndt2d_matcher_set_search_ahead(m, 1);
  uint32_t gsx = 0, gsy = 0;
  ndt2d_matcher_grid_info(m, &gsx, &gsy, NULL, NULL, NULL);
  double rpose[3] = {0, 0, 0}, rscore;
  ndt2d_matcher_match_scan(m, guess, scan_pts, N_BEAMS, rpose, cov, &rscore);
  char rvariant[128];
  snprintf(rvariant, sizeof(rvariant), "%s", ndt2d_last_variant(ndt2d_matcher_device(m)));
  char real[900];
  snprintf(real, sizeof(real),
           "{\"grid\": [%u, %u], \"map_scans\": %zu, \"range_max_m\": 30.0, \"match_scan_us\": %.2f, "
           "\"match_scan_p99_us\": %.2f, \"score_scan_us\": %.2f, \"add_scans_us\": %.2f, "
           "\"add_scans_back_to_back_us\": %.2f, "
           "\"mapper_cycle_us\": %.2f, \"mapper_cycle_p99_us\": %.2f, "
           "\"mapper_cycle_no_search_ahead_us\": %.2f, \"variant\": \"%s\", "
           "\"check_pose\": [%.17g, %.17g, %.17g], \"check_score\": %.17g}",
           gsx, gsy, n_map_scans, rmed[0], rp99[0], rmed[1], rmed[2], radd_b2b, rmed[3], rp99[3], rmed[4], rvariant,
           rpose[0], rpose[1], rpose[2], rscore);
  printf("{\"match_scan_us\": %.2f, \"match_scan_p99_us\": %.2f, \"score_scan_us\": %.2f, "
         "\"add_scans_us\": %.2f, \"add_scans_back_to_back_us\": %.2f, "
         "\"mapper_cycle_us\": %.2f, \"mapper_cycle_p99_us\": %.2f, "
         "\"mapper_cycle_no_search_ahead_us\": %.2f, \"search_ahead_launched\": %llu, "
         "\"search_ahead_collected\": %llu, "
         "\"measure_500_particles_unchanged_loop_us\": %.1f, \"score_points_call_us\": %.2f, "
         "\"measure_500_particles_batched_us\": %.2f, \"pf_measure_500_particles_us\": %.2f, "
         "\"variant\": \"%s\", "
         "\"check_pose\": [%.17g, %.17g, %.17g], \"check_score\": %.17g, \"real_lidar_map\": %s}\n",
         med[0], p99[0], med[1], med[2], add_b2b, med[3], p99[3], med[7], (unsigned long long)ahead_launched,
         (unsigned long long)ahead_collected, med[4], med[4] / 500.0, med[5], med[6],
         variant, pose[0], pose[1], pose[2], score, real);
  ndt2d_matcher_destroy(m);
  return 0;
}
