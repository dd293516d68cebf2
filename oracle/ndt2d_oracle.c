/*
 * ndt2d_oracle.c -- CPU restatement of ndt_2d's NDT scan-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline); see ndt2d_oracle.h
 * for the pinning status.  Every function cites the reference file:line whose
 * arithmetic (and operation order) it restates.  All arithmetic is IEEE double
 * with no FMA contraction (build with -ffp-contract=off; the reference's
 * default x86-64 Release build has no FMA instructions to contract into).
 */
#define _GNU_SOURCE /* sincos() */
#include "ndt2d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* cos(t) and sin(t) of the same argument, as the reference evaluates them.
 * Wherever the reference writes the pair -- `cos(pose.theta)` / `sin(pose.theta)`
 * and the like -- GCC (the ROS 2 toolchain; -O1 and up, no fast-math needed)
 * merges the two calls into ONE call of glibc's sincos().  That matters at the
 * last bit: glibc's sincos() and sin() are different code paths and can disagree
 * by one ulp (t = 0.4710119964311561: sin() ends ...722, sincos() ...721), which
 * then shows in every point transformed with it.  The pair is therefore spelled
 * out as sincos() here instead of being left to whatever compiler builds this file. */
static inline void cos_sin(double t, double * c, double * s)
{
  sincos(t, s, c);
}

/* ------------------------------------------------------------------------- */
/* Cell                                                                      */
/* ------------------------------------------------------------------------- */

/* Cell::Cell(), src/ndt_model.cpp:40-48 */
void orc_cell_init(orc_cell * c)
{
  memset(c, 0, sizeof(*c));
}

/* Cell::addPoint, src/ndt_model.cpp:50-63.
 * mean = (mean * n + point) / (n + 1); only the upper triangle (j >= i) of
 * `correlation` is updated, so correlation(1,0) stays 0. */
void orc_cell_add_point(orc_cell * c, double x, double y)
{
  const double p[2] = {x, y};
  const double n = c->n;
  c->mean[0] = (c->mean[0] * n + p[0]) / (n + 1);
  c->mean[1] = (c->mean[1] * n + p[1]) / (n + 1);
  for (int i = 0; i < 2; ++i)
  {
    for (int j = i; j < 2; ++j)
    {
      c->correlation[i * 2 + j] = (c->correlation[i * 2 + j] * n + p[i] * p[j]) / (n + 1);
    }
  }
  c->n += 1;
  c->valid = 0;
}

/* Eigenvalues of the 2x2 covariance as
 * Eigen::EigenSolver<Matrix2d>(covariance).eigenvalues().real() arrives at them
 * (src/ndt_model.cpp:84-85).  Eigen is a third-party dependency absent from this image
 * (find_package(Eigen3), CMakeLists.txt:17; ROS 2 Humble: 3.4.0); this is a transcription of
 * Eigen 3.4.0 for a real 2x2 input, operation by operation:
 *   RealSchur::compute (Eigenvalues/RealSchur.h): scale = matrix.cwiseAbs().maxCoeff(); the
 *     Schur form T is computed of matrix / scale and multiplied by scale at the end; a 2x2
 *     matrix is its own Hessenberg form (the Householder step has tau = 0).
 *   computeFromHessenberg: norm = sum over columns of |entries on and above the sub-diagonal|,
 *     considerAsZero = max(norm * eps^2, DBL_MIN); findSmallSubdiagEntry:
 *     |t10| <= max(eps * (|t00| + |t11|), considerAsZero)  =>  the diagonal is the result.
 *   splitOffTwoRows: p = (t00 - t11) / 2, q = p*p + t10*t01; q >= 0: z = sqrt|q|, a Givens
 *     rotation of (p + z, t10) (p >= 0) or (p - z, t10) applied as rot.adjoint() from the left
 *     and rot from the right, then t10 = 0.
 *   JacobiRotation::makeGivens (Jacobi/Jacobi.h, real case) and apply_rotation_in_the_plane:
 *     x' = c x + s y, y' = -s x + c y with separate multiplies and add (no FMA on x86-64).
 *   EigenSolver::compute (Eigenvalues/EigenSolver.h): t10 == 0: the diagonal of T; else a
 *     complex pair whose real part t11 + (t00 - t11) / 2 is returned twice.
 * PARITY UNPINNED at this level: the reference holds no vector for the eigenvalues themselves
 * and Eigen cannot be run here; test/ndt_model_tests.cpp pins only what follows from them
 * (tests/test_oracle_reference_vectors.py).  orc_set_eigen_form(1) selects the closed form
 * d + p +- z that rounds 1-4 used (ulps apart; tests/test_eigen_form.py counts the cells it
 * changes). */
static int g_eigen_form = 0;

void orc_set_eigen_form(int form) { g_eigen_form = form; }
int orc_get_eigen_form(void) { return g_eigen_form; }

static void make_givens(double p, double q, double * c, double * s)
{
  if (q == 0.0)
  {
    *c = p < 0.0 ? -1.0 : 1.0;
    *s = 0.0;
  }
  else if (p == 0.0)
  {
    *c = 0.0;
    *s = q < 0.0 ? 1.0 : -1.0;
  }
  else if (fabs(p) > fabs(q))
  {
    const double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    *c = 1.0 / u;
    *s = -t * *c;
  }
  else
  {
    const double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    *s = -1.0 / u;
    *c = -t * *s;
  }
}

static void eigenvalues_closed_form(const double * m, double * e0, double * e1)
{
  const double a = m[0], b = m[1], c = m[2], d = m[3];
  const double norm = fabs(a) + fabs(b) + fabs(c) + fabs(d);
  if (norm == 0.0)
  {
    *e0 = a;
    *e1 = d;
    return;
  }
  double consider_as_zero = norm * (DBL_EPSILON * DBL_EPSILON);
  if (consider_as_zero < DBL_MIN) consider_as_zero = DBL_MIN;
  double s = (fabs(a) + fabs(d)) * DBL_EPSILON;
  if (s < consider_as_zero) s = consider_as_zero;
  if (fabs(c) <= s)
  {
    *e0 = a;
    *e1 = d;
    return;
  }
  const double p = 0.5 * (a - d);
  const double q = p * p + c * b;
  const double z = sqrt(fabs(q));
  if (q >= 0.0)
  {
    *e0 = (d + p) + z;
    *e1 = (d + p) - z;
  }
  else
  {
    /* complex pair: .real() of both is d + p */
    *e0 = d + p;
    *e1 = d + p;
  }
}

static void eigenvalues_2x2(const double * m, double * e0, double * e1)
{
  if (g_eigen_form == 1)
  {
    eigenvalues_closed_form(m, e0, e1);
    return;
  }
  /* RealSchur::compute */
  double scale = fabs(m[0]);
  if (fabs(m[2]) > scale) scale = fabs(m[2]);
  if (fabs(m[1]) > scale) scale = fabs(m[1]);
  if (fabs(m[3]) > scale) scale = fabs(m[3]);
  if (scale < DBL_MIN)
  {
    *e0 = 0.0; /* m_matT.setZero() */
    *e1 = 0.0;
    return;
  }
  double t00 = m[0] / scale, t01 = m[1] / scale, t10 = m[2] / scale, t11 = m[3] / scale;
  /* computeNormOfT */
  double norm = 0.0;
  norm += fabs(t00) + fabs(t10);
  norm += fabs(t01) + fabs(t11);
  if (norm != 0.0)
  {
    double consider_as_zero = norm * (DBL_EPSILON * DBL_EPSILON);
    if (consider_as_zero < DBL_MIN) consider_as_zero = DBL_MIN;
    /* findSmallSubdiagEntry */
    double s = fabs(t00) + fabs(t11);
    s = s * DBL_EPSILON;
    if (s < consider_as_zero) s = consider_as_zero;
    if (fabs(t10) <= s)
    {
      t10 = 0.0;
    }
    else
    {
      /* splitOffTwoRows */
      const double p = 0.5 * (t00 - t11);
      const double q = p * p + t10 * t01;
      if (q >= 0.0)
      {
        const double z = sqrt(fabs(q));
        double c, sn;
        if (p >= 0.0) make_givens(p + z, t10, &c, &sn);
        else make_givens(p - z, t10, &c, &sn);
        const double jc = c, js = -sn; /* rot.adjoint() and rot.transpose(), real case */
        if (!(jc == 1.0 && js == 0.0))
        {
          double x, y;
          /* applyOnTheLeft(0, 1, rot.adjoint()) */
          x = t00; y = t10;
          t00 = jc * x + js * y;
          t10 = -js * x + jc * y;
          x = t01; y = t11;
          t01 = jc * x + js * y;
          t11 = -js * x + jc * y;
          /* applyOnTheRight(0, 1, rot) */
          x = t00; y = t01;
          t00 = jc * x + js * y;
          t01 = -js * x + jc * y;
          x = t10; y = t11;
          t10 = jc * x + js * y;
          t11 = -js * x + jc * y;
        }
        t10 = 0.0;
      }
    }
  }
  /* m_matT *= scale */
  t00 *= scale;
  t10 *= scale;
  t11 *= scale;
  /* EigenSolver::compute */
  if (t10 == 0.0)
  {
    *e0 = t00;
    *e1 = t11;
  }
  else
  {
    const double p = 0.5 * (t00 - t11);
    *e0 = t11 + p;
    *e1 = t11 + p;
  }
}

/* Cell::compute, src/ndt_model.cpp:65-103 */
void orc_cell_compute(orc_cell * c)
{
  if (c->valid || c->n < 3)
  {
    return;
  }

  const double scale = c->n / (c->n - 1);
  for (int i = 0; i < 2; ++i)
  {
    for (int j = i; j < 2; ++j)
    {
      c->covariance[i * 2 + j] =
        (c->correlation[i * 2 + j] - (c->mean[i] * c->mean[j])) * scale;
      c->covariance[j * 2 + i] = c->covariance[i * 2 + j];
    }
  }

  double small, large;
  eigenvalues_2x2(c->covariance, &small, &large);
  if (small > large)
  {
    double t = small;
    small = large;
    large = t;
  }
  if (small < 0.001 * large)
  {
    /* :88-96 */
    const double determinant = (0.001 * large) * large;
    c->information[0] = c->covariance[3] / determinant;
    c->information[1] = -c->covariance[2] / determinant;
    c->information[2] = -c->covariance[1] / determinant;
    c->information[3] = c->covariance[0] / determinant;
  }
  else
  {
    /* :99, Eigen fixed-size 2x2 inverse: invdet = 1/det,
     * result = [[d, -b], [-c, a]] * invdet */
    const double * m = c->covariance;
    const double det = m[0] * m[3] - m[2] * m[1];
    const double invdet = 1.0 / det;
    c->information[0] = m[3] * invdet;
    c->information[2] = -m[2] * invdet;
    c->information[1] = -m[1] * invdet;
    c->information[3] = m[0] * invdet;
  }
  c->valid = 1;
}

/* Cell::score, src/ndt_model.cpp:105-116.
 * exponent = ((-0.5 * q^T) * information) * q; `valid` is not consulted. */
double orc_cell_score(const orc_cell * c, double x, double y)
{
  if (c->n < 5)
  {
    return 0.0;
  }
  const double q0 = x - c->mean[0];
  const double q1 = y - c->mean[1];
  const double a0 = -0.5 * q0;
  const double a1 = -0.5 * q1;
  const double r0 = a0 * c->information[0] + a1 * c->information[2];
  const double r1 = a0 * c->information[1] + a1 * c->information[3];
  const double exponent = r0 * q0 + r1 * q1;
  return exp(exponent);
}

/* ------------------------------------------------------------------------- */
/* NDT                                                                       */
/* ------------------------------------------------------------------------- */

struct orc_ndt
{
  double cell_size;
  size_t size_x, size_y;
  double origin_x, origin_y;
  orc_cell * cells;
};

/* NDT::NDT, src/ndt_model.cpp:118-126: size_x_ = (size_t)(size_x/cell + 1) */
orc_ndt * orc_ndt_create(double cell_size, double size_x, double size_y,
                         double origin_x, double origin_y)
{
  orc_ndt * ndt = (orc_ndt *)calloc(1, sizeof(orc_ndt));
  ndt->cell_size = cell_size;
  ndt->size_x = (size_t)((size_x / cell_size) + 1);
  ndt->size_y = (size_t)((size_y / cell_size) + 1);
  ndt->origin_x = origin_x;
  ndt->origin_y = origin_y;
  size_t ncell = ndt->size_x * ndt->size_y;
  ndt->cells = (orc_cell *)calloc(ncell ? ncell : 1, sizeof(orc_cell));
  return ndt;
}

void orc_ndt_destroy(orc_ndt * ndt)
{
  if (!ndt) return;
  free(ndt->cells);
  free(ndt);
}

/* NDT::getIndex, src/ndt_model.cpp:203-218.  `unsigned int g = (double)`
 * truncates toward zero (x86-64: cvttsd2si to 64 bit, low 32 bits kept). */
int orc_ndt_get_index(const orc_ndt * ndt, double x, double y)
{
  if (x < ndt->origin_x || y < ndt->origin_y)
  {
    return -1;
  }
  unsigned int grid_x = (unsigned int)(long long)((x - ndt->origin_x) / ndt->cell_size);
  unsigned int grid_y = (unsigned int)(long long)((y - ndt->origin_y) / ndt->cell_size);
  if (grid_x >= ndt->size_x || grid_y >= ndt->size_y)
  {
    return -1;
  }
  return (int)((grid_y * ndt->size_x) + grid_x);
}

/* NDT::addScan, src/ndt_model.cpp:132-152 */
void orc_ndt_add_scan(orc_ndt * ndt, double pose_x, double pose_y, double pose_theta,
                      const double * points_xy, size_t n_points)
{
  double cos_th, sin_th;
  cos_sin(pose_theta, &cos_th, &sin_th);
  for (size_t k = 0; k < n_points; ++k)
  {
    const double px = points_xy[2 * k], py = points_xy[2 * k + 1];
    double p0 = pose_x;
    double p1 = pose_y;
    p0 += px * cos_th - py * sin_th;
    p1 += px * sin_th + py * cos_th;
    int index = orc_ndt_get_index(ndt, p0, p1);
    if (index >= 0)
    {
      orc_cell_add_point(&ndt->cells[index], p0, p1);
    }
  }
}

/* NDT::compute, src/ndt_model.cpp:154-160 */
void orc_ndt_compute(orc_ndt * ndt)
{
  size_t ncell = ndt->size_x * ndt->size_y;
  for (size_t i = 0; i < ncell; ++i)
  {
    orc_cell_compute(&ndt->cells[i]);
  }
}

/* NDT::likelihood(Vector2d), src/ndt_model.cpp:162-170 */
double orc_ndt_likelihood_point(const orc_ndt * ndt, double x, double y)
{
  int index = orc_ndt_get_index(ndt, x, y);
  if (index >= 0)
  {
    return orc_cell_score(&ndt->cells[index], x, y);
  }
  return 0.0;
}

/* NDT::likelihood(std::vector<Point>), src/ndt_model.cpp:178-187 */
double orc_ndt_likelihood_points(const orc_ndt * ndt, const double * points_xy,
                                 size_t n_points)
{
  double score = 0.0;
  for (size_t k = 0; k < n_points; ++k)
  {
    score += orc_ndt_likelihood_point(ndt, points_xy[2 * k], points_xy[2 * k + 1]);
  }
  return score;
}

/* toEigen(Pose2d) * (x, y, 1), include/ndt_2d/conversions.hpp:64-68 with Eigen's
 * Isometry3d * Vector3d: res = translation; res += linear * v where
 * linear = AngleAxisd(theta, Z).toRotationMatrix() = [[c,-s,0],[s,c,0],[0,0,(1-c)+c]]
 * and the third product term is 0 * 1 = 0 (adding it is exact). */
static void transform_point(double tx, double ty, double c, double s, double x, double y,
                            double * ox, double * oy)
{
  *ox = tx + (c * x + (-s) * y);
  *oy = ty + (s * x + c * y);
}

/* NDT::likelihood(ScanPtr), src/ndt_model.cpp:189-201 */
double orc_ndt_likelihood_scan(const orc_ndt * ndt, double pose_x, double pose_y,
                               double pose_theta, const double * points_xy,
                               size_t n_points)
{
  double c, s;
  cos_sin(pose_theta, &c, &s);
  double score = 0.0;
  for (size_t k = 0; k < n_points; ++k)
  {
    double px, py;
    transform_point(pose_x, pose_y, c, s, points_xy[2 * k], points_xy[2 * k + 1], &px, &py);
    score += orc_ndt_likelihood_point(ndt, px, py);
  }
  return score;
}

size_t orc_ndt_size_x(const orc_ndt * ndt) { return ndt->size_x; }
size_t orc_ndt_size_y(const orc_ndt * ndt) { return ndt->size_y; }
double orc_ndt_cell_size(const orc_ndt * ndt) { return ndt->cell_size; }
double orc_ndt_origin_x(const orc_ndt * ndt) { return ndt->origin_x; }
double orc_ndt_origin_y(const orc_ndt * ndt) { return ndt->origin_y; }
const orc_cell * orc_ndt_cells(const orc_ndt * ndt) { return ndt->cells; }

void orc_ndt_export_cells6(const orc_ndt * ndt, double * out)
{
  size_t ncell = ndt->size_x * ndt->size_y;
  for (size_t i = 0; i < ncell; ++i)
  {
    const orc_cell * c = &ndt->cells[i];
    out[6 * i + 0] = c->mean[0];
    out[6 * i + 1] = c->mean[1];
    out[6 * i + 2] = c->information[0];
    out[6 * i + 3] = c->information[1];
    out[6 * i + 4] = c->information[3];
    out[6 * i + 5] = c->n;
  }
}

/* ------------------------------------------------------------------------- */
/* ScanMatcherNDT                                                            */
/* ------------------------------------------------------------------------- */

struct orc_matcher
{
  double resolution;
  double angular_res, angular_size;
  double linear_res, linear_size;
  size_t laser_max_beams;
  double range_max;
  orc_ndt * ndt;
};

orc_matcher * orc_matcher_create(void)
{
  orc_matcher * m = (orc_matcher *)calloc(1, sizeof(orc_matcher));
  /* defaults of the declared parameters, src/scan_matcher_ndt.cpp:37-44 */
  m->resolution = 0.25;
  m->angular_res = 0.0025;
  m->angular_size = 0.1;
  m->linear_res = 0.005;
  m->linear_size = 0.05;
  m->laser_max_beams = 100;
  m->range_max = 0.0;
  return m;
}

void orc_matcher_destroy(orc_matcher * m)
{
  if (!m) return;
  orc_ndt_destroy(m->ndt);
  free(m);
}

/* ScanMatcherNDT::initialize, src/scan_matcher_ndt.cpp:35-47 */
void orc_matcher_initialize(orc_matcher * m, double ndt_resolution,
                            double search_angular_resolution, double search_angular_size,
                            double search_linear_resolution, double search_linear_size,
                            size_t laser_max_beams, double range_max)
{
  m->resolution = ndt_resolution;
  m->angular_res = search_angular_resolution;
  m->angular_size = search_angular_size;
  m->linear_res = search_linear_resolution;
  m->linear_size = search_linear_size;
  m->laser_max_beams = laser_max_beams;
  m->range_max = range_max;
}

/* ScanMatcherNDT::addScans, src/scan_matcher_ndt.cpp:49-74.  max_x_/max_y_
 * start at numeric_limits<double>::min() (smallest positive normal), as the
 * reference has it (:54,:56). */
void orc_matcher_add_scans(orc_matcher * m, const double * poses_xyt,
                           const double * points_xy, const size_t * offsets,
                           size_t n_scans)
{
  double min_x = DBL_MAX;
  double max_x = DBL_MIN;
  double min_y = DBL_MAX;
  double max_y = DBL_MIN;
  for (size_t k = 0; k < n_scans; ++k)
  {
    const double px = poses_xyt[3 * k], py = poses_xyt[3 * k + 1];
    min_x = fmin(px - m->range_max, min_x);
    max_x = fmax(px + m->range_max, max_x);
    min_y = fmin(py - m->range_max, min_y);
    max_y = fmax(py + m->range_max, max_y);
  }
  orc_ndt_destroy(m->ndt);
  m->ndt = orc_ndt_create(m->resolution, (max_x - min_x), (max_y - min_y), min_x, min_y);
  for (size_t k = 0; k < n_scans; ++k)
  {
    orc_ndt_add_scan(m->ndt, poses_xyt[3 * k], poses_xyt[3 * k + 1], poses_xyt[3 * k + 2],
                     points_xy + 2 * offsets[k], offsets[k + 1] - offsets[k]);
  }
  orc_ndt_compute(m->ndt);
}

void orc_matcher_reset(orc_matcher * m)
{
  orc_ndt_destroy(m->ndt);
  m->ndt = NULL;
}

int orc_matcher_has_ndt(const orc_matcher * m) { return m->ndt != NULL; }
const orc_ndt * orc_matcher_ndt(const orc_matcher * m) { return m->ndt; }

size_t orc_search_offsets(double size, double res, double * out, size_t cap)
{
  size_t n = 0;
  for (double v = -size; v < size; v += res)
  {
    if (out && n < cap) out[n] = v;
    ++n;
    if (!(res > 0.0)) break;  /* guard: the reference would spin forever */
  }
  return n;
}

/* One theta slab of matchScan's search, src/scan_matcher_ndt.cpp:104-142. */
typedef struct match_acc
{
  double best_score;
  double best_pose[3];
  int have_best;
  uint64_t best_index;
  double k[9];
  double u[3];
  double s;
} match_acc;

static void match_theta_slab(const orc_matcher * m, const double * scan_pose,
                             const double * points_xy, size_t n_points,
                             size_t scan_points_to_use, double scan_step, double dth,
                             uint64_t flat_base, double * outer, double * inner,
                             match_acc * acc, double * all_scores, size_t all_scores_cap)
{
  (void)n_points;
  /* :106-115 */
  double costh, sinth;
  cos_sin(scan_pose[2] + dth, &costh, &sinth);
  for (size_t i = 0; i < scan_points_to_use; ++i)
  {
    size_t scan_idx = (size_t)(i * scan_step);
    outer[2 * i] = points_xy[2 * scan_idx] * costh - points_xy[2 * scan_idx + 1] * sinth +
                   scan_pose[0];
    outer[2 * i + 1] = points_xy[2 * scan_idx] * sinth + points_xy[2 * scan_idx + 1] * costh +
                       scan_pose[1];
  }

  uint64_t flat = flat_base;
  for (double dx = -m->linear_size; dx < m->linear_size; dx += m->linear_res)
  {
    for (double dy = -m->linear_size; dy < m->linear_size; dy += m->linear_res)
    {
      /* :121-125 */
      for (size_t i = 0; i < scan_points_to_use; ++i)
      {
        inner[2 * i] = outer[2 * i] + dx;
        inner[2 * i + 1] = outer[2 * i + 1] + dy;
      }
      /* :127 */
      double score = -orc_ndt_likelihood_points(m->ndt, inner, scan_points_to_use);
      if (all_scores && flat < all_scores_cap) all_scores[flat] = score;
      /* :128-134 */
      if (score < acc->best_score)
      {
        acc->best_score = score;
        acc->best_pose[0] = dx;
        acc->best_pose[1] = dy;
        acc->best_pose[2] = dth;
        acc->have_best = 1;
        acc->best_index = flat;
      }
      /* :137-140: k += x * x^T * score; u += x * score; s += score */
      const double x[3] = {dx, dy, dth};
      for (int r = 0; r < 3; ++r)
      {
        for (int c = 0; c < 3; ++c)
        {
          acc->k[r * 3 + c] += (x[r] * x[c]) * score;
        }
        acc->u[r] += x[r] * score;
      }
      acc->s += score;
      ++flat;
    }
  }
}

static void match_finish(const match_acc * acc, double * pose_inout, double * covariance_out)
{
  if (acc->have_best && pose_inout)
  {
    pose_inout[0] = acc->best_pose[0];
    pose_inout[1] = acc->best_pose[1];
    pose_inout[2] = acc->best_pose[2];
  }
  /* :146: covariance = (1 / s) * k + (1 / (s * s) * u * u^T) */
  if (covariance_out)
  {
    const double inv_s = 1 / acc->s;
    const double inv_s2 = 1 / (acc->s * acc->s);
    for (int r = 0; r < 3; ++r)
    {
      for (int c = 0; c < 3; ++c)
      {
        covariance_out[r * 3 + c] = inv_s * acc->k[r * 3 + c] + (inv_s2 * acc->u[r]) * acc->u[c];
      }
    }
  }
}

/* ScanMatcherNDT::matchScan, src/scan_matcher_ndt.cpp:76-149 */
double orc_matcher_match_scan(const orc_matcher * m, const double * scan_pose_xyt,
                              const double * points_xy, size_t n_points,
                              double * pose_inout, double * covariance_out,
                              double * all_scores, size_t all_scores_cap,
                              size_t * n_candidates_out, uint64_t * best_index_out)
{
  if (n_candidates_out) *n_candidates_out = 0;
  if (best_index_out) *best_index_out = UINT64_MAX;
  /* :80 */
  if (!m->ndt) return 0.0;

  match_acc acc;
  memset(&acc, 0, sizeof(acc));
  acc.best_index = UINT64_MAX;

  /* :95-96 */
  size_t scan_points_to_use = m->laser_max_beams < n_points ? m->laser_max_beams : n_points;
  double scan_step = (double)n_points / (double)scan_points_to_use;

  double * outer = (double *)malloc(sizeof(double) * 2 * (scan_points_to_use + 1));
  double * inner = (double *)malloc(sizeof(double) * 2 * (scan_points_to_use + 1));

  const size_t n_lin = orc_search_offsets(m->linear_size, m->linear_res, NULL, 0);
  uint64_t flat = 0;
  /* :103 */
  for (double dth = -m->angular_size; dth < m->angular_size; dth += m->angular_res)
  {
    match_theta_slab(m, scan_pose_xyt, points_xy, n_points, scan_points_to_use, scan_step,
                     dth, flat, outer, inner, &acc, all_scores, all_scores_cap);
    flat += (uint64_t)n_lin * n_lin;
    if (!(m->angular_res > 0.0)) break;
  }
  free(outer);
  free(inner);

  match_finish(&acc, pose_inout, covariance_out);
  if (n_candidates_out) *n_candidates_out = (size_t)flat;
  if (best_index_out) *best_index_out = acc.best_index;
  /* :148 */
  return acc.best_score / scan_points_to_use;
}

/* One (theta, dx) strip of the search: the dy loop of src/scan_matcher_ndt.cpp:119-142 for a
 * given rotated scan `outer`.  Work unit of the OpenMP variant below. */
static void match_strip(const orc_matcher * m, size_t scan_points_to_use, double dth, double dx,
                        uint64_t flat_base, const double * outer, double * inner, match_acc * acc,
                        double * all_scores, size_t all_scores_cap)
{
  uint64_t flat = flat_base;
  for (double dy = -m->linear_size; dy < m->linear_size; dy += m->linear_res)
  {
    for (size_t i = 0; i < scan_points_to_use; ++i)
    {
      inner[2 * i] = outer[2 * i] + dx;
      inner[2 * i + 1] = outer[2 * i + 1] + dy;
    }
    double score = -orc_ndt_likelihood_points(m->ndt, inner, scan_points_to_use);
    if (all_scores && flat < all_scores_cap) all_scores[flat] = score;   /* each strip its own range */
    if (score < acc->best_score)
    {
      acc->best_score = score;
      acc->best_pose[0] = dx;
      acc->best_pose[1] = dy;
      acc->best_pose[2] = dth;
      acc->have_best = 1;
      acc->best_index = flat;
    }
    const double x[3] = {dx, dy, dth};
    for (int r = 0; r < 3; ++r)
    {
      for (int c = 0; c < 3; ++c) acc->k[r * 3 + c] += (x[r] * x[c]) * score;
      acc->u[r] += x[r] * score;
    }
    acc->s += score;
    ++flat;
    if (!(m->linear_res > 0.0)) break;
  }
}

/* matchScan with the (theta, dx) strips of the lattice dealt to OpenMP threads: the CPU
 * baseline on all host cores and the generator of full-lattice winners.  Every strip keeps
 * the reference's candidate order and strict `<`; strips are combined in lattice order, so
 * the winner (score, pose, flat index) is the sequential loop's.  The covariance
 * accumulators are per-strip partial sums added in lattice order (the sequential loop
 * keeps one running sum: equal to rounding). */
double orc_matcher_match_scan_omp_scores(const orc_matcher * m, const double * scan_pose_xyt,
                                         const double * points_xy, size_t n_points,
                                         double * pose_inout, double * covariance_out,
                                         int n_threads, uint64_t * best_index_out,
                                         int * threads_used_out, double * all_scores,
                                         size_t all_scores_cap)
{
  if (best_index_out) *best_index_out = UINT64_MAX;
  if (threads_used_out) *threads_used_out = 0;
  if (!m->ndt) return 0.0;
  size_t scan_points_to_use = m->laser_max_beams < n_points ? m->laser_max_beams : n_points;
  double scan_step = (double)n_points / (double)scan_points_to_use;

  const size_t n_th = orc_search_offsets(m->angular_size, m->angular_res, NULL, 0);
  const size_t n_lin = orc_search_offsets(m->linear_size, m->linear_res, NULL, 0);
  double * dths = (double *)malloc(sizeof(double) * (n_th + 1));
  double * dlin = (double *)malloc(sizeof(double) * (n_lin + 1));
  orc_search_offsets(m->angular_size, m->angular_res, dths, n_th);
  orc_search_offsets(m->linear_size, m->linear_res, dlin, n_lin);
  const long long n_strips = (long long)n_th * (long long)n_lin;
  match_acc * accs = (match_acc *)calloc((size_t)n_strips + 1, sizeof(match_acc));
  int threads_used = 1;

#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
#pragma omp parallel
  {
    double * outer = (double *)malloc(sizeof(double) * 2 * (scan_points_to_use + 1));
    double * inner = (double *)malloc(sizeof(double) * 2 * (scan_points_to_use + 1));
    long long outer_t = -1;
#ifdef _OPENMP
#pragma omp single
    threads_used = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 4)
    for (long long w = 0; w < n_strips; ++w)
    {
      const long long t = w / (long long)n_lin;
      const long long ix = w - t * (long long)n_lin;
      if (t != outer_t)
      {
        /* :106-115 */
        double costh, sinth;
        cos_sin(scan_pose_xyt[2] + dths[t], &costh, &sinth);
        for (size_t i = 0; i < scan_points_to_use; ++i)
        {
          size_t scan_idx = (size_t)(i * scan_step);
          outer[2 * i] = points_xy[2 * scan_idx] * costh - points_xy[2 * scan_idx + 1] * sinth +
                         scan_pose_xyt[0];
          outer[2 * i + 1] = points_xy[2 * scan_idx] * sinth + points_xy[2 * scan_idx + 1] * costh +
                             scan_pose_xyt[1];
        }
        outer_t = t;
      }
      accs[w].best_index = UINT64_MAX;
      match_strip(m, scan_points_to_use, dths[t], dlin[ix],
                  ((uint64_t)t * n_lin + (uint64_t)ix) * n_lin, outer, inner, &accs[w],
                  all_scores, all_scores_cap);
    }
    free(outer);
    free(inner);
  }

  match_acc acc;
  memset(&acc, 0, sizeof(acc));
  acc.best_index = UINT64_MAX;
  for (long long w = 0; w < n_strips; ++w)
  {
    if (accs[w].have_best && accs[w].best_score < acc.best_score)
    {
      acc.best_score = accs[w].best_score;
      memcpy(acc.best_pose, accs[w].best_pose, sizeof(acc.best_pose));
      acc.have_best = 1;
      acc.best_index = accs[w].best_index;
    }
    for (int i = 0; i < 9; ++i) acc.k[i] += accs[w].k[i];
    for (int i = 0; i < 3; ++i) acc.u[i] += accs[w].u[i];
    acc.s += accs[w].s;
  }
  free(accs);
  free(dths);
  free(dlin);
  match_finish(&acc, pose_inout, covariance_out);
  if (best_index_out) *best_index_out = acc.best_index;
  if (threads_used_out) *threads_used_out = threads_used;
  return acc.best_score / scan_points_to_use;
}

double orc_matcher_match_scan_omp_ex(const orc_matcher * m, const double * scan_pose_xyt,
                                     const double * points_xy, size_t n_points,
                                     double * pose_inout, double * covariance_out,
                                     int n_threads, uint64_t * best_index_out,
                                     int * threads_used_out)
{
  return orc_matcher_match_scan_omp_scores(m, scan_pose_xyt, points_xy, n_points, pose_inout,
                                           covariance_out, n_threads, best_index_out,
                                           threads_used_out, NULL, 0);
}

double orc_matcher_match_scan_omp(const orc_matcher * m, const double * scan_pose_xyt,
                                  const double * points_xy, size_t n_points,
                                  double * pose_inout, double * covariance_out,
                                  int n_threads)
{
  return orc_matcher_match_scan_omp_ex(m, scan_pose_xyt, points_xy, n_points, pose_inout,
                                       covariance_out, n_threads, NULL, NULL);
}

/* ScanMatcherNDT::scorePoints, src/scan_matcher_ndt.cpp:156-178 */
double orc_matcher_score_points(const orc_matcher * m, const double * points_xy,
                                size_t n_points, const double * pose_xyt)
{
  /* :159 */
  if (!m->ndt) return 0.0;
  /* :162, conversions.hpp:64-68 */
  double c, s;
  cos_sin(pose_xyt[2], &c, &s);
  /* :165-166 */
  size_t scan_points_to_use = m->laser_max_beams < n_points ? m->laser_max_beams : n_points;
  double scan_step = (double)n_points / (double)scan_points_to_use;

  double score = 0.0;
  for (size_t i = 0; i < scan_points_to_use; ++i)
  {
    size_t scan_idx = (size_t)(i * scan_step);
    double px, py;
    transform_point(pose_xyt[0], pose_xyt[1], c, s, points_xy[2 * scan_idx],
                    points_xy[2 * scan_idx + 1], &px, &py);
    score += -orc_ndt_likelihood_point(m->ndt, px, py);
  }
  /* :177 */
  return score / scan_points_to_use;
}

/* ScanMatcherNDT::scoreScan, src/scan_matcher_ndt.cpp:151-154 */
double orc_matcher_score_scan(const orc_matcher * m, const double * scan_pose_xyt,
                              const double * points_xy, size_t n_points)
{
  return orc_matcher_score_points(m, points_xy, n_points, scan_pose_xyt);
}

/* ------------------------------------------------------------------------- */
/* ParticleFilter                                                            */
/* ------------------------------------------------------------------------- */

/* ParticleFilter::measure loop, src/particle_filter.cpp:81-87 */
void orc_pf_measure(const orc_matcher * m, const double * particles_xyt, size_t n_particles,
                    const double * points_xy, size_t n_points, double * weights_out,
                    int copy_points)
{
  for (size_t i = 0; i < n_particles; ++i)
  {
    if (copy_points)
    {
      /* scan->getPoints() returns the vector by value, src/scan.cpp:67-70 */
      double * tmp = (double *)malloc(sizeof(double) * 2 * (n_points + 1));
      memcpy(tmp, points_xy, sizeof(double) * 2 * n_points);
      weights_out[i] = orc_matcher_score_points(m, tmp, n_points, particles_xyt + 3 * i);
      free(tmp);
    }
    else
    {
      weights_out[i] = orc_matcher_score_points(m, points_xy, n_points, particles_xyt + 3 * i);
    }
  }
}

void orc_pf_measure_omp(const orc_matcher * m, const double * particles_xyt,
                        size_t n_particles, const double * points_xy, size_t n_points,
                        double * weights_out, int n_threads)
{
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n_particles; ++i)
  {
    weights_out[i] = orc_matcher_score_points(m, points_xy, n_points, particles_xyt + 3 * i);
  }
}

/* ROS angles::normalize_angle (angles/angles.h, unpinned dependency):
 *   result = fmod(angle + pi, 2pi); result <= 0 ? result + pi : result - pi */
double orc_normalize_angle(double a)
{
  const double result = fmod(a + M_PI, 2.0 * M_PI);
  if (result <= 0.0) return result + M_PI;
  return result - M_PI;
}

/* angles::shortest_angular_distance(from, to) = normalize_angle(to - from) */
double orc_shortest_angular_distance(double from, double to)
{
  return orc_normalize_angle(to - from);
}

/* ParticleFilter::updateStatistics, src/particle_filter.cpp:163-218 */
void orc_pf_update_statistics(const double * particles_xyt, double * weights,
                              size_t n_particles, double * mean_out, double * cov_inout)
{
  /* :166-174 */
  double sum_weight = 0.0;
  for (size_t i = 0; i < n_particles; ++i) sum_weight += weights[i];
  for (size_t i = 0; i < n_particles; ++i) weights[i] /= sum_weight;

  /* :177-200 */
  double mean[3] = {0, 0, 0};
  double corr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double sum_cos_th = 0.0, sum_sin_th = 0.0;
  for (size_t i = 0; i < n_particles; ++i)
  {
    const double * p = particles_xyt + 3 * i;
    const double w = weights[i];
    mean[0] += w * p[0];
    mean[1] += w * p[1];
    mean[2] += w * p[2];
    double cos_p, sin_p;
    cos_sin(p[2], &cos_p, &sin_p);
    sum_cos_th += w * cos_p;
    sum_sin_th += w * sin_p;
    for (int j = 0; j < 2; ++j)
    {
      for (int k = j; k < 2; ++k)
      {
        corr[j * 3 + k] += w * p[j] * p[k];
      }
    }
  }

  /* :203-205 */
  mean_out[0] = mean[0];
  mean_out[1] = mean[1];
  mean_out[2] = atan2(sum_sin_th, sum_cos_th);

  /* :208-215 */
  for (int j = 0; j < 2; ++j)
  {
    for (int k = j; k < 2; ++k)
    {
      cov_inout[j * 3 + k] = corr[j * 3 + k] - mean[j] * mean[k];
      cov_inout[k * 3 + j] = cov_inout[j * 3 + k];
    }
  }

  /* :218-222: cov_(2,2) accumulates; it is never zeroed */
  for (size_t i = 0; i < n_particles; ++i)
  {
    double d = orc_shortest_angular_distance(particles_xyt[3 * i + 2], mean_out[2]);
    cov_inout[8] += weights[i] * d * d;
  }
}

/* ------------------------------------------------------------------------- */
/* MotionModel                                                               */
/* ------------------------------------------------------------------------- */

/* MotionModel::sample, src/motion_model.cpp:45-83 */
void orc_motion_sample(double dx, double dy, double dth, const double * alphas5,
                       double * poses_xyt, size_t n, const float * z, double * params_out)
{
  const double a1 = alphas5[0], a2 = alphas5[1], a3 = alphas5[2], a4 = alphas5[3];
  /* :49-52 */
  const double trans = hypot(dx, dy);
  const double rot1 = (trans > 0.01) ? atan2(dy, dx) : 0.0;
  const double rot2 = orc_shortest_angular_distance(rot1, dth);
  /* :55-58 */
  const double rot1_ = fmin(fabs(orc_shortest_angular_distance(rot1, 0.0)),
                            fabs(orc_shortest_angular_distance(rot1, M_PI)));
  const double rot2_ = fmin(fabs(orc_shortest_angular_distance(rot2, 0.0)),
                            fabs(orc_shortest_angular_distance(rot2, M_PI)));
  /* :61-67 */
  const double sigma_rot1 = sqrt(a1 * rot1_ * rot1_ + a2 * trans * trans);
  const double sigma_trans = sqrt(a3 * trans * trans + a4 * rot1_ * rot1_ + a4 * rot2_ * rot2_);
  const double sigma_rot2 = sqrt(a1 * rot2_ * rot2_ + a2 * trans * trans);
  if (params_out)
  {
    params_out[0] = rot1;
    params_out[1] = trans;
    params_out[2] = rot2;
    params_out[3] = sigma_rot1;
    params_out[4] = sigma_trans;
    params_out[5] = sigma_rot2;
  }
  /* :70-72 normal_distribution<float>(mean, sigma): parameters held as float */
  const float m1 = (float)rot1, s1 = (float)sigma_rot1;
  const float mt = (float)trans, st = (float)sigma_trans;
  const float m2 = (float)rot2, s2 = (float)sigma_rot2;
  for (size_t i = 0; i < n; ++i)
  {
    /* :76-78 */
    const float r1 = z[3 * i] * s1 + m1;
    const float t = z[3 * i + 1] * st + mt;
    const float r2 = z[3 * i + 2] * s2 + m2;
    double * pose = poses_xyt + 3 * i;
    /* :80-82 */
    double cos_h, sin_h;
    cos_sin(pose[2] + r1, &cos_h, &sin_h);
    pose[0] += t * cos_h;
    pose[1] += t * sin_h;
    pose[2] = orc_normalize_angle(pose[2] + r1 + r2);
  }
}

/* ParticleFilter::init, src/particle_filter.cpp:53-69 (the sampling loop) */
void orc_pf_init(double x, double y, double theta, double sigma_x, double sigma_y,
                 double sigma_theta, double * poses_xyt, size_t n, const float * z)
{
  /* :56-58 normal_distribution<float>(mean, sigma) */
  const float mx = (float)x, sx = (float)sigma_x;
  const float my = (float)y, sy = (float)sigma_y;
  const float mt = (float)theta, st = (float)sigma_theta;
  for (size_t i = 0; i < n; ++i)
  {
    /* :62-64 float draws widened to double */
    poses_xyt[3 * i] = z[3 * i] * sx + mx;
    poses_xyt[3 * i + 1] = z[3 * i + 1] * sy + my;
    poses_xyt[3 * i + 2] = orc_normalize_angle(z[3 * i + 2] * st + mt);
  }
}

/* ------------------------------------------------------------------------- */
/* LaserScan -> Scan conversion                                              */
/* ------------------------------------------------------------------------- */

/* NdtMapper::laserCallback, src/ndt_mapper.cpp:385-453 */
size_t orc_convert_scan(const float * ranges, size_t n_ranges, const orc_laser_scan * scan,
                        double * points_xy_out)
{
  /* :391-394 */
  const double per_x = scan->motion_x / n_ranges;
  const double per_y = scan->motion_y / n_ranges;
  const double per_th = scan->motion_theta / n_ranges;
  /* :403-404 */
  double cos_lt, sin_lt;
  cos_sin(scan->laser_theta, &cos_lt, &sin_lt);
  const double range_max = scan->range_max;
  size_t n_out = 0;
  if (scan->inverted)
  {
    if (n_ranges == 0) return 0;  /* size() - 1 would wrap in the reference */
    /* :410 `for (i = size - 1; i > 0; --i)`: index 0 is never visited */
    for (size_t i = n_ranges - 1; i > 0; --i)
    {
      /* :413 */
      if (isnan(ranges[i]) || ranges[i] > range_max) continue;
      /* :415 float arithmetic (size_t * float -> float), negated, widened */
      const float a = -(scan->angle_min + (float)i * scan->angle_increment);
      const double angle = a;
      double cos_a, sin_a;
      cos_sin(angle, &cos_a, &sin_a);
      const double lx = cos_a * ranges[i];
      const double ly = sin_a * ranges[i];
      /* :419-420 */
      const double px = cos_lt * lx - sin_lt * ly + scan->laser_x;
      const double py = sin_lt * lx + cos_lt * ly + scan->laser_y;
      /* :422-426 */
      double cos_tt, sin_tt;
      cos_sin(scan->motion_theta - (per_th * i), &cos_tt, &sin_tt);
      points_xy_out[2 * n_out] = cos_tt * px - sin_tt * py + (scan->motion_x - (per_x * i));
      points_xy_out[2 * n_out + 1] = sin_tt * px + cos_tt * py + (scan->motion_y - (per_y * i));
      ++n_out;
    }
  }
  else
  {
    /* :433 */
    for (size_t i = 0; i < n_ranges; ++i)
    {
      /* :436 */
      if (isnan(ranges[i]) || ranges[i] > range_max) continue;
      /* :438 */
      const float a = (scan->angle_min + (float)i * scan->angle_increment);
      const double angle = a;
      double cos_a, sin_a;
      cos_sin(angle, &cos_a, &sin_a);
      const double lx = cos_a * ranges[i];
      const double ly = sin_a * ranges[i];
      /* :442-443 */
      const double px = cos_lt * lx - sin_lt * ly + scan->laser_x;
      const double py = sin_lt * lx + cos_lt * ly + scan->laser_y;
      /* :445-448 */
      double cos_tt, sin_tt;
      cos_sin(per_th * i, &cos_tt, &sin_tt);
      points_xy_out[2 * n_out] = cos_tt * px - sin_tt * py + (per_x * i);
      points_xy_out[2 * n_out + 1] = sin_tt * px + cos_tt * py + (per_y * i);
      ++n_out;
    }
  }
  return n_out;
}

/* ------------------------------------------------------------------------- */
/* OccupancyGrid                                                             */
/* ------------------------------------------------------------------------- */

/* OccupancyGrid::updateBounds, src/occupancy_grid.cpp:154-185 */
void orc_occupancy_update_bounds(double * bounds, double resolution, const double * poses_xyt,
                                 const double * points_xy, const size_t * offsets,
                                 size_t first_scan, size_t n_scans)
{
  double min_x = bounds[0], max_x = bounds[1], min_y = bounds[2], max_y = bounds[3];
  for (size_t i = first_scan; i < n_scans; ++i)
  {
    const double x = poses_xyt[3 * i], y = poses_xyt[3 * i + 1];
    double cos_th, sin_th;
    cos_sin(poses_xyt[3 * i + 2], &cos_th, &sin_th);
    for (size_t k = offsets[i]; k < offsets[i + 1]; ++k)
    {
      const double qx = points_xy[2 * k], qy = points_xy[2 * k + 1];
      /* :171-173 */
      double px = x, py = y;
      px += qx * cos_th - qy * sin_th;
      py += qx * sin_th + qy * cos_th;
      /* :174-177 std::min(p, m) = (m < p) ? m : p;  std::max(p, m) = (p < m) ? m : p */
      min_x = (min_x < px) ? min_x : px;
      max_x = (px < max_x) ? max_x : px;
      min_y = (min_y < py) ? min_y : py;
      max_y = (py < max_y) ? max_y : py;
    }
  }
  /* :181-184 */
  bounds[0] = floor(min_x / resolution) * resolution;
  bounds[1] = ceil(max_x / resolution) * resolution;
  bounds[2] = floor(min_y / resolution) * resolution;
  bounds[3] = ceil(max_y / resolution) * resolution;
}

/* OccupancyGrid::getMsg after the bounds update, src/occupancy_grid.cpp:56-151 */
void orc_occupancy_render(const double * bounds, double resolution, double occ_thresh,
                          const double * poses_xyt, const double * points_xy,
                          const size_t * offsets, size_t n_scans, uint32_t * info_wh,
                          double * origin_xy, signed char * data)
{
  /* :57-65 (info.width / height are uint32: the quotient is truncated) */
  const double pad = 5 * resolution;
  const uint32_t width = (uint32_t)((bounds[1] - bounds[0] + 2 * pad) / resolution);
  const uint32_t height = (uint32_t)((bounds[3] - bounds[2] + 2 * pad) / resolution);
  const double origin_x = bounds[0] - pad;
  const double origin_y = bounds[2] - pad;
  info_wh[0] = width;
  info_wh[1] = height;
  origin_xy[0] = origin_x;
  origin_xy[1] = origin_y;
  if (data == NULL) return;
  const size_t n_cells = (size_t)width * height;
  int * hit = (int *)calloc(n_cells ? n_cells : 1, sizeof(int));
  int * empty = (int *)calloc(n_cells ? n_cells : 1, sizeof(int));
  for (size_t i = 0; i < n_scans; ++i)
  {
    const double pose_x = poses_xyt[3 * i], pose_y = poses_xyt[3 * i + 1];
    double cos_th, sin_th;
    cos_sin(poses_xyt[3 * i + 2], &cos_th, &sin_th);
    /* :82-83 */
    const int start_x = (int)((pose_x - origin_x) / resolution);
    const int start_y = (int)((pose_y - origin_y) / resolution);
    for (size_t k = offsets[i]; k < offsets[i + 1]; ++k)
    {
      const double qx = points_xy[2 * k], qy = points_xy[2 * k + 1];
      /* :87-91 */
      const double point_x = qx * cos_th - qy * sin_th + pose_x;
      const double point_y = qx * sin_th + qy * cos_th + pose_y;
      const int end_x = (int)((point_x - origin_x) / resolution);
      const int end_y = (int)((point_y - origin_y) / resolution);
      /* :93-98 simplified Bresenham */
      int dx = abs(end_x - start_x);
      int sx = (start_x < end_x) ? 1 : -1;
      int dy = -abs(end_y - start_y);
      int sy = (start_y < end_y) ? 1 : -1;
      int error = dx + dy;
      int x = start_x, y = start_y;
      while (1)
      {
        const int inside = x >= 0 && y >= 0 && (uint32_t)x < width && (uint32_t)y < height;
        const size_t index = inside ? (size_t)x + (size_t)y * width : 0;
        if (x == end_x && y == end_y)
        {
          if (inside) ++hit[index];
          break;
        }
        if (inside) ++empty[index];
        if (2 * error >= dy)
        {
          if (x == end_x)
          {
            if (inside) ++hit[index];
            break;
          }
          error = error + dy;
          x += sx;
        }
        if (2 * error <= dx)
        {
          if (y == end_y)
          {
            if (inside) ++hit[index];
            break;
          }
          error = error + dx;
          y += sy;
        }
      }
    }
  }
  /* :134-150 */
  for (size_t i = 0; i < n_cells; ++i)
  {
    const double touches = hit[i] + empty[i];
    data[i] = -1;
    if (touches > 0.5)
    {
      data[i] = ((double)hit[i] / touches > occ_thresh) ? 100 : 0;
    }
  }
  free(hit);
  free(empty);
}
