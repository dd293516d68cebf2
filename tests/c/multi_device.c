/* Plain-C consumer of the multi-device matcher (ndt2d_matcher_create_multi,
 * include/ndt2d_hip.h): the 8-GPU split of matchScan / ParticleFilter::measure as
 * an unchanged C or C++ host reaches it -- one handle, several device contexts.
 *
 *   multi_device <ids> [exchange] [cfg5]     ids = comma-separated device ids, e.g. 0,0,0
 *
 * Runs cfg-1, cfg-2 and cfg-4's searches (the same world, map and
 * query scan as ndt_2d_amd/synth.py) on a single-device matcher and on the
 * multi-device one, and a 20,000-particle measure, and prints one JSON object:
 * the winners, whether the multi-device results equal the single-device ones bit
 * for bit (score, index, pose) and how far the covariances are apart.
 * With "cfg5": BASELINE.json configs[4] instead -- 1,000,000 particles x 720 beams on the
 * 801 x 801 NDT -- through a multi-device matcher whose thresholds are the library's DEFAULTS
 * (the call must be dealt out by itself): raw scores bit for bit the single-device ones, the
 * normalised weights and the statistics to rounding (the total weight is summed per device).
 * Exit code 0 = every comparison holds, 2 = no GPU.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ndt2d_hip.h"

#define N_SCANS 9
#define N_BEAMS 720
#define PI 3.14159265358979323846

static double map_poses[3 * N_SCANS], map_pts[2 * N_BEAMS * N_SCANS], scan_pts[2 * N_BEAMS];
static size_t map_off[N_SCANS + 1];

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

typedef struct
{
  double pose[3], cov[9], score;
  uint64_t best;
  size_t n_cand;
  double ms;
} result;

static int run_match(ndt2d_matcher * m, double lin_size, double lin_res, double ang_size, double ang_res, result * r)
{
  const double guess[3] = {0.0, 0.0, 0.0};
  if (ndt2d_matcher_initialize(m, 0.25, ang_res, ang_size, lin_res, lin_size, N_BEAMS, 4.75) != NDT2D_OK) return 1;
  if (ndt2d_matcher_reset(m) != NDT2D_OK) return 1;
  if (ndt2d_matcher_add_scans(m, map_poses, map_pts, map_off, N_SCANS) != NDT2D_OK) return 1;
  r->ms = 1e300;
  for (int rep = 0; rep < 3; ++rep)
  {
    r->pose[0] = r->pose[1] = r->pose[2] = 0.0;
    const double t0 = now_ms();
    if (ndt2d_matcher_match_scan_ex(m, guess, scan_pts, N_BEAMS, r->pose, r->cov, &r->score, NULL, 0, &r->n_cand,
                                    &r->best) != NDT2D_OK)
    {
      fprintf(stderr, "match_scan: %s\n", ndt2d_matcher_last_error(m));
      return 1;
    }
    const double dt = now_ms() - t0;
    if (dt < r->ms) r->ms = dt;
  }
  return 0;
}

static double cov_rel_diff(const result * a, const result * b)
{
  double worst = 0.0;
  for (int i = 0; i < 9; ++i)
  {
    const double d = fabs(a->cov[i] - b->cov[i]) / (fabs(b->cov[i]) > 1e-300 ? fabs(b->cov[i]) : 1.0);
    if (d > worst) worst = d;
  }
  return worst;
}

static int same_winner(const result * a, const result * b)
{
  return a->best == b->best && a->n_cand == b->n_cand && memcmp(&a->score, &b->score, sizeof(double)) == 0 &&
         memcmp(a->pose, b->pose, sizeof(a->pose)) == 0;
}


/* BASELINE.json configs[4] (the map scans, query scan and particles of
 * ndt_2d_amd/synth.py): the particle set of the 8-GPU global localisation, sharded by the
 * matcher's own default thresholds. */
static int run_cfg5(const int * ids, int n_dev, const char * exchange)
{
  enum { K = 40, NP = 1000000 };
  const ndt2d_world world = {95.0, 5.0, 0.25};
  double * poses = (double *)malloc(sizeof(double) * 3 * K * K);
  double * pts = (double *)malloc(sizeof(double) * 2 * N_BEAMS * K * K);
  size_t * off = (size_t *)malloc(sizeof(size_t) * (K * K + 1));
  double * parts = (double *)malloc(sizeof(double) * 3 * NP);
  double * u = (double *)malloc(sizeof(double) * 3 * NP);
  double * s1 = (double *)malloc(sizeof(double) * NP);
  double * sn = (double *)malloc(sizeof(double) * NP);
  if (!poses || !pts || !off || !parts || !u || !s1 || !sn) return 1;
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
    parts[3 * i + 2] = (2.0 * u[3 * i + 2] - 1.0) * PI;
  }
  ndt2d_matcher *single = NULL, *multi = NULL;
  int rc = ndt2d_matcher_create(&single, ids[0]);
  if (rc == NDT2D_ERR_NO_DEVICE || rc == NDT2D_ERR_HIP)
  {
    printf("no GPU: ndt2d_matcher_create -> %d (no CPU fallback)\n", rc);
    return 2;
  }
  if (rc != NDT2D_OK || ndt2d_matcher_create_multi(&multi, ids, n_dev) != NDT2D_OK) return 3;
  if (ndt2d_matcher_set_exchange(multi, exchange) != NDT2D_OK) return 6;
  /* (the thresholds are NOT touched: 7.2e8 units must be above the default for pose batches) */
  double thr_search = 0.0, thr_poses = 0.0;
  if (ndt2d_matcher_get_multi_thresholds(multi, &thr_search, &thr_poses) != NDT2D_OK) return 7;
  ndt2d_matcher * both[2] = {single, multi};
  for (int q = 0; q < 2; ++q)
  {
    if (ndt2d_matcher_initialize(both[q], 0.25, 0.0025, 0.1, 0.005, 0.05, N_BEAMS, 12.25) != NDT2D_OK) return 8;
    if (ndt2d_matcher_add_scans(both[q], poses, pts, off, n_scans) != NDT2D_OK) return 9;
  }
  uint32_t gsx = 0, gsy = 0;
  ndt2d_matcher_grid_info(multi, &gsx, &gsy, NULL, NULL, NULL);
  int ok = gsx == 801 && gsy == 801;
  /* raw scores: a particle's score does not depend on which device scored it */
  if (ndt2d_matcher_score_poses(single, scan_pts, N_BEAMS, parts, NP, s1) != NDT2D_OK) return 10;
  if (ndt2d_matcher_score_poses(multi, scan_pts, N_BEAMS, parts, NP, sn) != NDT2D_OK)
  {
    fprintf(stderr, "score_poses: %s\n", ndt2d_matcher_last_error(multi));
    return 11;
  }
  char sp_variant[160];
  snprintf(sp_variant, sizeof(sp_variant), "%s", ndt2d_matcher_last_variant(multi));
  const int scores_identical = memcmp(s1, sn, sizeof(double) * NP) == 0;
  size_t nonzero = 0;
  for (int i = 0; i < NP; ++i) nonzero += s1[i] != 0.0;
  /* the whole of measure */
  double mean1[3], cov1[9] = {0}, meann[3], covn[9] = {0};
  cov1[8] = covn[8] = 0.125;
  if (ndt2d_matcher_pf_measure(single, parts, NP, scan_pts, N_BEAMS, s1, mean1, cov1) != NDT2D_OK) return 12;
  const double t0 = now_ms();
  if (ndt2d_matcher_pf_measure(multi, parts, NP, scan_pts, N_BEAMS, sn, meann, covn) != NDT2D_OK)
  {
    fprintf(stderr, "pf_measure: %s\n", ndt2d_matcher_last_error(multi));
    return 13;
  }
  const double multi_ms = now_ms() - t0;
  char pf_variant[160];
  snprintf(pf_variant, sizeof(pf_variant), "%s", ndt2d_matcher_last_variant(multi));
  double w_rel = 0.0, stat_diff = 0.0, w_sum = 0.0;
  for (int i = 0; i < NP; ++i)
  {
    const double d = fabs(s1[i] - sn[i]) / (fabs(s1[i]) > 0.0 ? fabs(s1[i]) : 1.0);
    if (d > w_rel) w_rel = d;
    w_sum += sn[i];
  }
  for (int i = 0; i < 3; ++i) stat_diff = fmax(stat_diff, fabs(mean1[i] - meann[i]));
  for (int i = 0; i < 9; ++i) stat_diff = fmax(stat_diff, fabs(cov1[i] - covn[i]));
  double fan[64];
  size_t n_fan = 0;
  ndt2d_matcher_last_fanout_us(multi, fan, 64, &n_fan);
  if (!scores_identical || !(w_rel < 1e-12) || !(stat_diff < 1e-9) || !(fabs(w_sum - 1.0) < 1e-9) || nonzero < 1000) ok = 0;
  if (n_dev > 1 && (strncmp(pf_variant, "multi[", 6) != 0 || strncmp(sp_variant, "multi[", 6) != 0 || n_fan != (size_t)n_dev)) ok = 0;
  printf("{\"devices\": %d, \"exchange\": \"%s\", \"ok\": %s, \"cfg5\": {\"particles\": %d, \"grid\": [%u, %u], "
         "\"default_min_pose_units\": %.3e, \"units\": %.3e, \"raw_scores_bit_identical\": %s, \"scoring_particles\": %zu, "
         "\"max_rel_weight_diff\": %.3e, \"max_stat_diff\": %.3e, \"multi_call_ms\": %.3f, \"score_poses_variant\": \"%s\", "
         "\"variant\": \"%s\"}}\n",
         n_dev, exchange, ok ? "true" : "false", (int)NP, gsx, gsy, thr_poses, (double)NP * N_BEAMS,
         scores_identical ? "true" : "false", nonzero, w_rel, stat_diff, multi_ms, sp_variant, pf_variant);
  ndt2d_matcher_destroy(multi);
  ndt2d_matcher_destroy(single);
  free(poses); free(pts); free(off); free(parts); free(u); free(s1); free(sn);
  return ok ? 0 : 20;
}

int main(int argc, char ** argv)
{
  int ids[64], n_dev = 0;
  if (argc < 2)
  {
    fprintf(stderr, "usage: %s <device ids, comma separated> [auto|host|rccl]\n", argv[0]);
    return 64;
  }
  {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s", argv[1]);
    for (char * tok = strtok(buf, ","); tok != NULL && n_dev < 64; tok = strtok(NULL, ",")) ids[n_dev++] = atoi(tok);
  }
  const char * exchange = argc > 2 ? argv[2] : "auto";
  if (argc > 3 && strcmp(argv[3], "cfg5") == 0) return run_cfg5(ids, n_dev, exchange);

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

  ndt2d_matcher *single = NULL, *multi = NULL;
  int rc = ndt2d_matcher_create(&single, ids[0]);
  if (rc == NDT2D_ERR_NO_DEVICE || rc == NDT2D_ERR_HIP)
  {
    printf("no GPU: ndt2d_matcher_create -> %d (no CPU fallback)\n", rc);
    return 2;
  }
  if (rc != NDT2D_OK) return 3;
  rc = ndt2d_matcher_create_multi(&multi, ids, n_dev);
  if (rc != NDT2D_OK)
  {
    fprintf(stderr, "ndt2d_matcher_create_multi -> %d\n", rc);
    return 4;
  }
  if (ndt2d_matcher_device_count(multi) != n_dev) return 5;
  if (ndt2d_matcher_set_exchange(multi, exchange) != NDT2D_OK) return 6;
  if (ndt2d_matcher_set_multi_min_units(multi, 0.0) != NDT2D_OK) return 7;   /* cfg-1 is dealt out too */
  ndt2d_matcher_set_timing(single, 0);
  ndt2d_matcher_set_timing(multi, 0);

  /* lin size/res, ang size/res: cfg-1, cfg-2, cfg-4 (BASELINE.json configs) */
  const double cfg[3][4] = {{0.5, 0.05, 0.2, 0.01}, {1.0, 0.02, 0.5, 0.005}, {5.0, 0.02, PI, 0.005}};
  const char * names[3] = {"cfg1", "cfg2", "cfg4"};
  const uint64_t winners[3] = {0, 1065647ull, 80443810ull};   /* tests/golden/big_winners.json (cfg-1: see the JSON) */
  int ok = 1;
  char variant[3][160];
  result rs[3], rm[3];
  for (int c = 0; c < 3; ++c)
  {
    if (run_match(single, cfg[c][0], cfg[c][1], cfg[c][2], cfg[c][3], &rs[c]) != 0) return 8;
    if (run_match(multi, cfg[c][0], cfg[c][1], cfg[c][2], cfg[c][3], &rm[c]) != 0) return 9;
    snprintf(variant[c], sizeof(variant[c]), "%s", ndt2d_matcher_last_variant(multi));
    if (!same_winner(&rs[c], &rm[c])) ok = 0;
    if (cov_rel_diff(&rm[c], &rs[c]) > 1e-9) ok = 0;
    if (c > 0 && rm[c].best != winners[c]) ok = 0;
    if (n_dev > 1 && strncmp(variant[c], "multi[", 6) != 0) ok = 0;
  }

  /* ParticleFilter::measure, 20,000 particles in the room (uniform; theta uniform) */
  enum { NP = 20000 };
  static double parts[3 * NP], u[3 * NP], w1[NP], wn[NP];
  ndt2d_synth_uniform(303u, 3 * NP, u);
  for (int i = 0; i < NP; ++i)
  {
    parts[3 * i] = (2.0 * u[3 * i] - 1.0) * 3.9;
    parts[3 * i + 1] = (2.0 * u[3 * i + 1] - 1.0) * 3.9;
    parts[3 * i + 2] = (2.0 * u[3 * i + 2] - 1.0) * PI;
  }
  double mean1[3], cov1[9] = {0}, meann[3], covn[9] = {0};
  cov1[8] = covn[8] = 0.125;   /* cov_(2,2) accumulates (src/particle_filter.cpp:216) */
  if (ndt2d_matcher_pf_measure(single, parts, NP, scan_pts, N_BEAMS, w1, mean1, cov1) != NDT2D_OK) return 10;
  if (ndt2d_matcher_pf_measure(multi, parts, NP, scan_pts, N_BEAMS, wn, meann, covn) != NDT2D_OK)
  {
    fprintf(stderr, "pf_measure: %s\n", ndt2d_matcher_last_error(multi));
    return 11;
  }
  char pf_variant[160];
  snprintf(pf_variant, sizeof(pf_variant), "%s", ndt2d_matcher_last_variant(multi));
  double w_diff = 0.0, stat_diff = 0.0, w_sum = 0.0;
  for (int i = 0; i < NP; ++i)
  {
    const double d = fabs(w1[i] - wn[i]);
    if (d > w_diff) w_diff = d;
    w_sum += wn[i];
  }
  for (int i = 0; i < 3; ++i) stat_diff = fmax(stat_diff, fabs(mean1[i] - meann[i]));
  for (int i = 0; i < 9; ++i) stat_diff = fmax(stat_diff, fabs(cov1[i] - covn[i]));
  if (!(w_diff < 1e-15) || !(stat_diff < 1e-10) || !(fabs(w_sum - 1.0) < 1e-9)) ok = 0;
  if (n_dev > 1 && strncmp(pf_variant, "multi[", 6) != 0) ok = 0;

  printf("{\"devices\": %d, \"exchange\": \"%s\", \"ok\": %s", n_dev, exchange, ok ? "true" : "false");
  for (int c = 0; c < 3; ++c)
  {
    printf(", \"%s\": {\"best_index\": %llu, \"n_candidates\": %zu, \"same_winner_score_pose\": %s, "
           "\"cov_rel_diff\": %.3e, \"single_ms\": %.3f, \"multi_ms\": %.3f, \"score\": %.17g, \"variant\": \"%s\"}",
           names[c], (unsigned long long)rm[c].best, rm[c].n_cand, same_winner(&rs[c], &rm[c]) ? "true" : "false",
           cov_rel_diff(&rm[c], &rs[c]), rs[c].ms, rm[c].ms, rm[c].score, variant[c]);
  }
  printf(", \"pf_measure\": {\"particles\": %d, \"max_weight_diff\": %.3e, \"max_stat_diff\": %.3e, \"variant\": \"%s\"}}\n",
         (int)NP, w_diff, stat_diff, pf_variant);
  ndt2d_matcher_destroy(multi);
  ndt2d_matcher_destroy(single);
  return ok ? 0 : 20;
}
