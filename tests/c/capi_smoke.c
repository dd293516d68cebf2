/* Plain-C consumer of include/ndt2d_hip.h: the boundary has no C++ in it.
 * Built with gcc, linked against libndt2d_hip.so.  Without a GPU it checks that
 * the compute entry points fail loudly (no CPU fallback); with one it runs the
 * reference's known-answer case test_ndt (test/ndt_model_tests.cpp:191-230)
 * through ndt2d_matcher_*: likelihood((3.5, 3.5)) = 0.7659 +- 1e-3. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "ndt2d_hip.h"

int main(void)
{
  /* host-only entry points work anywhere */
  double off[64];
  size_t n = 0;
  if (ndt2d_search_offsets(0.05, 0.005, off, 64, &n) != NDT2D_OK || n != 21) return 10;
  if (ndt2d_abi_version() != NDT2D_ABI_VERSION || NDT2D_ABI_VERSION != 4) return 11;

  const double poses[3] = {0.0, 0.0, 0.0};
  const double pts[10] = {3.5, 3.5, 3.45, 3.4, 3.55, 3.6, 3.45, 3.6, 3.45, 3.6};
  const size_t offsets[2] = {0, 5};
  double cells[11 * 11 * 6];
  uint32_t sx = 0, sy = 0;
  double ox = 0, oy = 0;
  if (ndt2d_host_build_grid(1.0, 5.0, poses, pts, offsets, 1, cells, 121, &sx, &sy, &ox, &oy) !=
      NDT2D_OK)
    return 12;
  if (sx != 11 || sy != 11 || ox != -5.0 || oy != -5.0 || cells[96 * 6 + 5] != 5.0) return 13;

  ndt2d_matcher * m = NULL;
  int rc = ndt2d_matcher_create(&m, 0);
  if (rc == NDT2D_ERR_NO_DEVICE || rc == NDT2D_ERR_HIP)
  {
    printf("no GPU: ndt2d_matcher_create -> %d (no CPU fallback)\n", rc);
    return 0;
  }
  if (rc != NDT2D_OK) return 14;
  if (ndt2d_matcher_initialize(m, 1.0, 0.0025, 0.1, 0.005, 0.05, 100, 5.0) != NDT2D_OK) return 15;
  double score = 1.0;
  const double q[2] = {3.5, 3.5};
  const double id[3] = {0.0, 0.0, 0.0};
  /* no NDT yet: 0.0 (reference src/scan_matcher_ndt.cpp:159) */
  if (ndt2d_matcher_score_points(m, q, 1, id, &score) != NDT2D_OK || score != 0.0) return 16;
  if (ndt2d_matcher_add_scans(m, poses, pts, offsets, 1) != NDT2D_OK) return 17;
  if (ndt2d_matcher_score_points(m, q, 1, id, &score) != NDT2D_OK) return 18;
  printf("likelihood((3.5,3.5)) = %.15f\n", -score);
  if (fabs(-score - 0.7659) > 1e-3) return 19;
  if (fabs(-score - 0.76592833836492369) > 1e-12) return 20;
  if (ndt2d_matcher_reset(m) != NDT2D_OK || ndt2d_matcher_has_ndt(m)) return 21;
  ndt2d_matcher_destroy(m);
  printf("ok\n");
  return 0;
}
