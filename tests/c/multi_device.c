/* Plain-C consumer of the multi-device matcher (ndt2d_matcher_create_multi,
 * include/ndt2d_hip.h): the 8-GPU split of matchScan / ParticleFilter::measure as
 * an unchanged C or C++ host reaches it -- one handle, several device contexts.
 *
 *   multi_device <ids> [exchange]     ids = comma-separated device ids, e.g. 0,0,0
 *
 * Runs cfg-1, cfg-2 and cfg-4's searches (the same world, map and
 * query scan as ndt_2d_amd/synth.py) on a single-device matcher and on the
 * multi-device one, and a 20,000-particle measure, and prints one JSON object:
 * the winners, whether the multi-device results equal the single-device ones bit
 * for bit (score, index, pose) and how far the covariances are apart.
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
