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
/* what follows an addScans in the node: scoreScan and matchScan of the new scan (untimed filler
 * between timed addScans calls -- a scoreScan alone would leave its search-ahead in flight and the
 * next addScans waiting for it, which is not the node's sequence) */
static void op_score_and_match(void)
{
  op_score_scan();
  op_match();
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


/* ", \"fanout_us\": [...], \"fanout_last_us\": x, \"fanout_skew_us\": y" of the matcher's last dealt call */
static size_t fanout_json(char * out, size_t cap)
{
  double f[64];
  size_t n = 0, len = 0;
  if (ndt2d_matcher_last_fanout_us(m, f, 64, &n) != NDT2D_OK || n == 0) return 0;
  if (n > 64) n = 64;
  double lo = f[0], hi = f[0];
  len += (size_t)snprintf(out + len, cap - len, ", \"fanout_us\": [");
  for (size_t r = 0; r < n; ++r)
  {
    len += (size_t)snprintf(out + len, cap - len, "%s%.1f", r ? ", " : "", f[r]);
    if (f[r] < lo) lo = f[r];
    if (f[r] > hi) hi = f[r];
  }
  len += (size_t)snprintf(out + len, cap - len, "], \"fanout_last_us\": %.1f, \"fanout_skew_us\": %.1f", hi, hi - lo);
  return len;
}

/* --workload cfg5: BASELINE.json configs[4] through ONE multi-device matcher with its DEFAULT
 * thresholds -- the unchanged node's filter_->measure(global_scan_matcher_, scan) (reference
 * src/ndt_mapper.cpp:474 -> src/particle_filter.cpp:78-89) with the plugin's device_ids set:
 * 1,000,000 particles x 720 beams against the 801 x 801 NDT of the 190 m room (SURVEY.md 8d;
 * the same map scans, query scan and particles as ndt_2d_amd/synth.py).  The whole
 * ndt2d_matcher_pf_measure call (host particles in, host weights + statistics out), the
 * fan-out stamps, and every device's particle range measured ALONE on a one-device matcher. */
static int cfg5_mode(const int * ids, int n_dev, const char * exchange, char ** json_out)
{
  enum { K = 40, NP = 1000000 };
  const ndt2d_world world = {95.0, 5.0, 0.25};
  double * poses = (double *)malloc(sizeof(double) * 3 * K * K);
  double * pts = (double *)malloc(sizeof(double) * 2 * N_BEAMS * K * K);
  size_t * off = (size_t *)malloc(sizeof(size_t) * (K * K + 1));
  double * parts = (double *)malloc(sizeof(double) * 3 * NP);
  double * u = (double *)malloc(sizeof(double) * 3 * NP);
  double * w = (double *)malloc(sizeof(double) * NP);
  double * w1 = (double *)malloc(sizeof(double) * NP);
  if (!poses || !pts || !off || !parts || !u || !w || !w1) return 1;
  size_t n_scans = 0;
  int idx = 0;
  for (int j = 0; j < K; ++j)
  {
    for (int i = 0; i < K; ++i, ++idx)
    {
      const double x = (i - (K - 1) / 2.0) * 4.5, y = (j - (K - 1) / 2.0) * 4.5;
      if (ndt2d_synth_pose_blocked(&world, x, y, 0.25)) continue;
      double * p = poses + 3 * n_scans;
      p[0] = x;
      p[1] = y;
      p[2] = 0.0;
      if (ndt2d_synth_scan(&world, p, N_BEAMS, 0.01, 5u * 1000003u + (unsigned)idx, pts + 2 * N_BEAMS * n_scans) != NDT2D_OK)
        return 1;
      off[n_scans] = (size_t)N_BEAMS * n_scans;
      ++n_scans;
    }
  }
  off[n_scans] = (size_t)N_BEAMS * n_scans;
  const double truth[3] = {1.0, 0.5, 0.3};
  if (ndt2d_synth_scan(&world, truth, N_BEAMS, 0.01, 501u, scan_pts) != NDT2D_OK) return 1;
  ndt2d_synth_uniform(505u, 3 * (size_t)NP, u);
  for (int i = 0; i < NP; ++i)
  {
    parts[3 * i] = (2.0 * u[3 * i] - 1.0) * 95.0;
    parts[3 * i + 1] = (2.0 * u[3 * i + 1] - 1.0) * 95.0;
    parts[3 * i + 2] = (2.0 * u[3 * i + 2] - 1.0) * 3.14159265358979323846;
  }
  ndt2d_matcher * single = NULL;
  if (ndt2d_matcher_create_multi(&m, ids, n_dev) != NDT2D_OK || ndt2d_matcher_create(&single, ids[0]) != NDT2D_OK)
  {
    fprintf(stderr, "ndt2d_matcher_create_multi failed (no GPU: there is no CPU fallback)\n");
    return 2;
  }
  if (ndt2d_matcher_set_exchange(m, exchange) != NDT2D_OK) return 3;
  ndt2d_matcher_set_timing(m, 0);
  ndt2d_matcher_set_timing(single, 0);
  ndt2d_matcher * both[2] = {m, single};
  for (int q = 0; q < 2; ++q)
  {
    if (ndt2d_matcher_initialize(both[q], 0.25, 0.0025, 0.1, 0.005, 0.05, N_BEAMS, 12.25) != NDT2D_OK) return 4;
    if (ndt2d_matcher_add_scans(both[q], poses, pts, off, n_scans) != NDT2D_OK)
    {
      fprintf(stderr, "add_scans: %s\n", ndt2d_matcher_last_error(both[q]));
      return 5;
    }
  }
  uint32_t gsx = 0, gsy = 0;
  ndt2d_matcher_grid_info(m, &gsx, &gsy, NULL, NULL, NULL);
  double mean[3], cov[9] = {0}, t[9];
  const int reps = 9;
  for (int r = -2; r < reps; ++r)
  {
    const double t0 = now_us();
    if (ndt2d_matcher_pf_measure(m, parts, NP, scan_pts, N_BEAMS, w, mean, cov) != NDT2D_OK)
    {
      fprintf(stderr, "pf_measure: %s\n", ndt2d_matcher_last_error(m));
      return 6;
    }
    if (r >= 0) t[r] = now_us() - t0;
  }
  qsort(t, (size_t)reps, sizeof(double), cmp);
  const double ms = t[reps / 2] * 1e-3;
  char variant[160];
  snprintf(variant, sizeof(variant), "%s", ndt2d_matcher_last_variant(m));
  char * out = (char *)malloc(8192);
  size_t len = 0, cap = 8192;
  double thr_s = 0.0, thr_p = 0.0;
  ndt2d_matcher_get_multi_thresholds(m, &thr_s, &thr_p);
  len += (size_t)snprintf(out + len, cap - len,
                          "{\"particles\": %d, \"beams\": %d, \"grid\": [%u, %u], \"map_scans\": %zu, \"units\": %.4e, "
                          "\"multi_min_pose_units\": %.3e, \"call_ms\": %.4f, \"units_per_s\": %.4e, \"variant\": \"%s\", "
                          "\"mean\": [%.17g, %.17g, %.17g]",
                          (int)NP, N_BEAMS, gsx, gsy, n_scans, (double)NP * N_BEAMS, thr_p, ms,
                          (double)NP * N_BEAMS / (ms * 1e-3), variant, mean[0], mean[1], mean[2]);
  if (strncmp(variant, "multi[", 6) == 0) len += fanout_json(out + len, cap - len);
  /* the same call with the particles and the weights in pinned host memory (ndt2d_host_alloc:
   * what the C++ mirror's particle store uses) -- the uploads are then DMA the call does not wait for */
  double pinned_ms = -1.0, single_pinned_ms = -1.0;
  {
    void *pp = NULL, *pw = NULL;
    if (ndt2d_host_alloc(ndt2d_matcher_device(m), sizeof(double) * 3 * NP, &pp) == NDT2D_OK &&
        ndt2d_host_alloc(ndt2d_matcher_device(m), sizeof(double) * NP, &pw) == NDT2D_OK)
    {
      memcpy(pp, parts, sizeof(double) * 3 * NP);
      ndt2d_matcher * who[2] = {m, single};
      for (int q = 0; q < 2; ++q)
      {
        double c9[9] = {0}, mean_p[3], tp[9];
        int bad = 0;
        for (int r = -2; r < reps && !bad; ++r)
        {
          const double t0 = now_us();
          bad = ndt2d_matcher_pf_measure(who[q], (const double *)pp, NP, scan_pts, N_BEAMS, (double *)pw, mean_p, c9) != NDT2D_OK;
          if (r >= 0) tp[r] = now_us() - t0;
        }
        if (!bad)
        {
          qsort(tp, (size_t)reps, sizeof(double), cmp);
          if (q == 0) pinned_ms = tp[reps / 2] * 1e-3;
          else single_pinned_ms = tp[reps / 2] * 1e-3;
        }
      }
    }
    if (pp != NULL) ndt2d_host_free(ndt2d_matcher_device(m), pp);
    if (pw != NULL) ndt2d_host_free(ndt2d_matcher_device(m), pw);
  }
  len += (size_t)snprintf(out + len, cap - len, ", \"call_pinned_ms\": %.4f, \"single_device_call_pinned_ms\": %.4f", pinned_ms,
                          single_pinned_ms);
  /* the whole set on ONE device, and every device's range alone on one device */
  double mean1[3], cov1[9] = {0};
  for (int r = -2; r < reps; ++r)
  {
    const double t0 = now_us();
    if (ndt2d_matcher_pf_measure(single, parts, NP, scan_pts, N_BEAMS, w1, mean1, cov1) != NDT2D_OK) return 7;
    if (r >= 0) t[r] = now_us() - t0;
  }
  qsort(t, (size_t)reps, sizeof(double), cmp);
  const double single_ms = t[reps / 2] * 1e-3;
  double w_rel = 0.0;
  for (int i = 0; i < NP; ++i)
  {
    const double d = fabs(w[i] - w1[i]) / (fabs(w1[i]) > 0.0 ? fabs(w1[i]) : 1.0);
    if (d > w_rel) w_rel = d;
  }
  double alone[64], slowest = 0.0, sum = 0.0;
  for (int r = 0; r < n_dev; ++r)
  {
    const size_t base = (size_t)NP / (size_t)n_dev, rem = (size_t)NP % (size_t)n_dev;
    const size_t begin = (size_t)r * base + ((size_t)r < rem ? (size_t)r : rem);
    const size_t nr = base + ((size_t)r < rem ? 1 : 0);
    double ts[5];
    for (int k = -1; k < 5; ++k)
    {
      const double t0 = now_us();
      if (ndt2d_matcher_pf_measure(single, parts + 3 * begin, nr, scan_pts, N_BEAMS, w1, mean1, cov1) != NDT2D_OK) return 8;
      if (k >= 0) ts[k] = now_us() - t0;
    }
    qsort(ts, 5, sizeof(double), cmp);
    alone[r] = ts[2] * 1e-3;
    if (alone[r] > slowest) slowest = alone[r];
    sum += alone[r];
  }
  len += (size_t)snprintf(out + len, cap - len, ", \"single_device_call_ms\": %.4f, \"max_rel_weight_diff_vs_single\": %.3e, "
                          "\"shares_alone_ms\": [", single_ms, w_rel);
  for (int r = 0; r < n_dev; ++r) len += (size_t)snprintf(out + len, cap - len, "%s%.4f", r ? ", " : "", alone[r]);
  len += (size_t)snprintf(out + len, cap - len,
                          "], \"slowest_share_ms\": %.4f, \"sum_of_shares_ms\": %.4f, \"call_minus_slowest_share_us\": %.1f, "
                          "\"call_minus_sum_of_shares_us\": %.1f}",
                          slowest, sum, (ms - slowest) * 1e3, (ms - sum) * 1e3);
  ndt2d_matcher_destroy(single);
  ndt2d_matcher_destroy(m);
  m = NULL;
  free(poses); free(pts); free(off); free(parts); free(u); free(w); free(w1);
  *json_out = out;
  return 0;
}

/* --devices 0,1,..: the loop-closure search of cfg-4 (+-5 m / 0.02 m x +-pi /
 * 0.005 rad = 315,508,257 candidates x 720 beams) and cfg-2's (2,000,000 candidates) through ONE
 * multi-device matcher (ndt2d_matcher_create_multi) -- what the unchanged node's
 * global_scan_matcher_->matchScan() (reference src/ndt_mapper.cpp:634-643) costs with the
 * plugin's device_ids parameter set.  [--exchange auto|host|rccl].  One JSON object. */
static int multi_mode(const char * id_list, const char * exchange, const char * workload)
{
  int ids[64], n_dev = 0;
  char buf[256];
  snprintf(buf, sizeof(buf), "%s", id_list);
  for (char * tok = strtok(buf, ","); tok != NULL && n_dev < 64; tok = strtok(NULL, ",")) ids[n_dev++] = atoi(tok);
  if (n_dev == 0) return 64;
  if (strcmp(workload, "cfg5") == 0)
  {
    char * p5 = NULL;
    const int rc5 = cfg5_mode(ids, n_dev, exchange, &p5);
    if (rc5 != 0) return rc5;
    printf("\n{\"devices\": %d, \"exchange_requested\": \"%s\", \"cfg5\": %s}\n", n_dev, exchange, p5);
    free(p5);
    return 0;
  }
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
  static char out[16384];
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
    char variant[160];
    snprintf(variant, sizeof(variant), "%s", ndt2d_matcher_last_variant(m));
    len += (size_t)snprintf(out + len, sizeof(out) - len,
                            ", \"%s\": {\"n_candidates\": %zu, \"best_index\": %llu, \"score\": %.17g, \"step_ms\": %.4f, "
                            "\"units_per_s\": %.4e, \"variant\": \"%s\"",
                            names[c], n_cand, (unsigned long long)best, score, ms,
                            (double)n_cand * N_BEAMS / (ms * 1e-3), variant);
    if (strncmp(variant, "multi[", 6) == 0)
    {
      /* when each device's search had been queued, from the call's start (the last call's) */
      len += fanout_json(out + len, sizeof(out) - len);
      /* every device's share searched ALONE on its context (the tables and beams of the last call
       * are in place): launch to result, median of 5 -- the call cannot be faster than its slowest
       * share; contexts that share one GPU run their shares one after the other (the sum) */
      const size_t n_th = c == 0 ? 200 : 1257;   /* T1 of SURVEY.md 8: the visited theta offsets of cfg-2 / cfg-4 */
      double alone[64], slowest = 0.0, sum = 0.0;
      for (int r = 0; r < n_dev; ++r)
      {
        ndt2d_handle h = ndt2d_matcher_device_at(m, r);
        const size_t count = (size_t)r < n_th ? (n_th - (size_t)r + (size_t)n_dev - 1) / (size_t)n_dev : 0;
        double ts[5];
        for (int k = -1; k < 5 && count > 0; ++k)
        {
          ndt2d_match_result res;
          const double t0 = now_us();
          if (ndt2d_match_launch_strided(h, (size_t)r, (size_t)n_dev, count, NULL, NULL) != NDT2D_OK ||
              ndt2d_match_fetch(h, &res) != NDT2D_OK)
          {
            fprintf(stderr, "share %d alone: %s\n", r, ndt2d_last_error(h));
            return 8;
          }
          if (k >= 0) ts[k] = now_us() - t0;
        }
        if (count > 0) qsort(ts, 5, sizeof(double), cmp);
        alone[r] = count > 0 ? ts[2] * 1e-3 : 0.0;
        if (alone[r] > slowest) slowest = alone[r];
        sum += alone[r];
      }
      len += (size_t)snprintf(out + len, sizeof(out) - len, ", \"n_th\": %zu, \"shares_alone_ms\": [", n_th);
      for (int r = 0; r < n_dev; ++r) len += (size_t)snprintf(out + len, sizeof(out) - len, "%s%.4f", r ? ", " : "", alone[r]);
      len += (size_t)snprintf(out + len, sizeof(out) - len,
                              "], \"slowest_share_ms\": %.4f, \"sum_of_shares_ms\": %.4f, "
                              "\"call_minus_slowest_share_us\": %.1f, \"call_minus_sum_of_shares_us\": %.1f",
                              slowest, sum, (ms - slowest) * 1e3, (ms - sum) * 1e3);
    }
    len += (size_t)snprintf(out + len, sizeof(out) - len, "}");
    if (len >= sizeof(out)) return 7;
  }
  if (strcmp(workload, "all") == 0 || strcmp(workload, "overhead") == 0)
  {
    /* What dealing a call costs by itself: workloads too small to matter (cfg-1's lattice of
     * 17,640 candidates x 720 beams: ~50 us of one GPU; 16,384 particles: ~30 us), the
     * thresholds at zero, the multi-device call against the same call on ONE device.  The N
     * small kernels of the dealt call run side by side even when the contexts share a GPU,
     * so the difference is host time: waking the device threads, N uploads and launches in
     * parallel, the exchange. */
    ndt2d_matcher * one = NULL;
    if (ndt2d_matcher_create(&one, ids[0]) != NDT2D_OK) return 2;
    ndt2d_matcher_set_timing(one, 0);
    if (ndt2d_matcher_set_multi_min_units(m, 0.0) != NDT2D_OK) return 3;
    ndt2d_matcher * pair[2] = {one, m};
    double call_us[2][2] = {{0, 0}, {0, 0}};
    char ov_variant[2][160] = {"", ""};
    enum { NSMALL = 16384 };
    static double sp[3 * NSMALL], sw[NSMALL], su[3 * NSMALL];
    ndt2d_synth_uniform(909u, 3 * NSMALL, su);
    for (int i = 0; i < NSMALL; ++i)
    {
      sp[3 * i] = (2.0 * su[3 * i] - 1.0) * 3.9;
      sp[3 * i + 1] = (2.0 * su[3 * i + 1] - 1.0) * 3.9;
      sp[3 * i + 2] = (2.0 * su[3 * i + 2] - 1.0) * 3.14159265358979323846;
    }
    for (int q = 0; q < 2; ++q)
    {
      if (ndt2d_matcher_initialize(pair[q], 0.25, 0.01, 0.2, 0.05, 0.5, N_BEAMS, 4.75) != NDT2D_OK) return 4;
      if (ndt2d_matcher_reset(pair[q]) != NDT2D_OK || ndt2d_matcher_add_scans(pair[q], map_poses, map_pts, map_off, N_SCANS) != NDT2D_OK) return 5;
      enum { R = 300 };
      static double ts[R];
      double pose[3], cov[9], score, mean[3];
      for (int r = -20; r < R; ++r)
      {
        pose[0] = pose[1] = pose[2] = 0.0;
        const double t0 = now_us();
        if (ndt2d_matcher_match_scan(pair[q], zero, scan_pts, N_BEAMS, pose, cov, &score) != NDT2D_OK) return 6;
        if (r >= 0) ts[r] = now_us() - t0;
      }
      qsort(ts, R, sizeof(double), cmp);
      call_us[q][0] = ts[R / 2];
      if (q == 1) snprintf(ov_variant[0], sizeof(ov_variant[0]), "%s", ndt2d_matcher_last_variant(m));
      for (int r = -20; r < R; ++r)
      {
        double c9[9] = {0};
        const double t0 = now_us();
        if (ndt2d_matcher_pf_measure(pair[q], sp, NSMALL, scan_pts, N_BEAMS, sw, mean, c9) != NDT2D_OK) return 6;
        if (r >= 0) ts[r] = now_us() - t0;
      }
      qsort(ts, R, sizeof(double), cmp);
      call_us[q][1] = ts[R / 2];
      if (q == 1) snprintf(ov_variant[1], sizeof(ov_variant[1]), "%s", ndt2d_matcher_last_variant(m));
    }
    len += (size_t)snprintf(out + len, sizeof(out) - len,
                            ", \"dealing_overhead\": {\"search\": {\"workload\": \"cfg-1 lattice, 17640 candidates x 720 beams\", "
                            "\"one_device_call_us\": %.1f, \"dealt_call_us\": %.1f, \"difference_us\": %.1f, \"variant\": \"%s\"",
                            call_us[0][0], call_us[1][0], call_us[1][0] - call_us[0][0], ov_variant[0]);
    /* (the fan-out of the last dealt call is the particle call's: queried below) */
    len += (size_t)snprintf(out + len, sizeof(out) - len,
                            "}, \"pf_measure\": {\"workload\": \"16384 particles x 720 beams, cfg-1 map, pageable host memory\", "
                            "\"one_device_call_us\": %.1f, \"dealt_call_us\": %.1f, \"difference_us\": %.1f, \"variant\": \"%s\"",
                            call_us[0][1], call_us[1][1], call_us[1][1] - call_us[0][1], ov_variant[1]);
    if (strncmp(ov_variant[1], "multi[", 6) == 0) len += fanout_json(out + len, sizeof(out) - len);
    len += (size_t)snprintf(out + len, sizeof(out) - len, "}}");
    ndt2d_matcher_destroy(one);
    if (len >= sizeof(out)) return 7;
  }
  if (strcmp(workload, "all") == 0)
  {
    ndt2d_matcher_destroy(m);
    m = NULL;
    char * p5 = NULL;
    const int rc5 = cfg5_mode(ids, n_dev, exchange, &p5);
    if (rc5 != 0) return rc5;
    len += (size_t)snprintf(out + len, sizeof(out) - len, ", \"cfg5\": %s", p5);
    free(p5);
    if (len >= sizeof(out)) return 7;
  }
  printf("\n%s}\n", out);
  if (m != NULL) ndt2d_matcher_destroy(m);
  return 0;
}

int main(int argc, char ** argv)
{
  {
    const char * devices = NULL;
    const char * exchange = "auto";
    const char * workload = "search";
    for (int i = 1; i + 1 < argc; ++i)
    {
      if (strcmp(argv[i], "--devices") == 0) devices = argv[i + 1];
      if (strcmp(argv[i], "--exchange") == 0) exchange = argv[i + 1];
      if (strcmp(argv[i], "--workload") == 0) workload = argv[i + 1];   /* search (default) | cfg5 | all */
    }
    if (devices != NULL) return multi_mode(devices, exchange, workload);
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
  /* reset + addScans as the node issues it: scoreScan + matchScan of the scan follow every one
   * (untimed here).  Back to back -- nothing fetched in between -- each install first
   * has to ask the stream whether the one before has read the staging buffer. */
  measure_between(op_add, op_score_and_match, REPS / 4, &med[2], &p99[2]);
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
  measure_between(op_add, op_score_and_match, REPS / 4, &rmed[2], &rp99[2]);
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
