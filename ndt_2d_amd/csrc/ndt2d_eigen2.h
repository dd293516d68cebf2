// Eigenvalues of Cell::compute's 2x2 covariance as Eigen::EigenSolver<Eigen::Matrix2d> arrives at
// them (reference src/ndt_model.cpp:84-85: `solver.eigenvalues().real()`), for the host NDT build
// (ndt2d_host.cpp, g++) and the device build (ndt2d_build.hip, hipcc) alike.
//
// Eigen is not in this image; this is a transcription of Eigen 3.4.0 (the version ROS 2 Humble
// resolves `find_package(Eigen3)` to) for a real 2 x 2 input, operation by operation:
//   RealSchur::compute            Eigenvalues/RealSchur.h   scale = max |m_ij|; the Schur form is
//                                 computed of m / scale and multiplied by scale afterwards
//   HessenbergDecomposition       a 2 x 2 matrix is its own Hessenberg form (tau = 0: x * 1.0)
//   computeNormOfT, findSmallSubdiagEntry, splitOffTwoRows    the deflation test
//                                 |t10| <= max(eps (|t00| + |t11|), max(norm eps^2, DBL_MIN)) and
//                                 the split of the 2 x 2 block: p = (t00 - t11) / 2,
//                                 q = p p + t10 t01, z = sqrt |q|, Givens rotation of
//                                 (p +- z, t10) applied from the left (adjoint) and the right
//   JacobiRotation::makeGivens    Jacobi/Jacobi.h, real case
//   apply_rotation_in_the_plane   x' = c x + s y, y' = -s x + c y (separate multiplies and add:
//                                 an x86-64 build has no fused multiply-add)
//   EigenSolver::compute          Eigenvalues/EigenSolver.h   reads the diagonal (a complex pair
//                                 -- impossible for a symmetric input -- gives t11 + p twice)
// The eigenvalues differ from the closed form d + p +- z in the last ulps; they reach the result
// through the branch choice `small < 0.001 * large` (:88) and the clamp branch's determinant (:91).
// form == 1 selects that closed form (rounds 1-4; kept for A/B: tests/test_eigen_form.py counts
// the cells of the synthetic maps whose information matrix differs between the two).
#ifndef NDT2D_EIGEN2_H_
#define NDT2D_EIGEN2_H_

#include <math.h>

#if defined(__HIPCC__)
#define NDT2D_HD __host__ __device__ __forceinline__
#else
#define NDT2D_HD inline
#endif

namespace ndt2d
{

constexpr int kEigenFormSchur = 0;    // Eigen 3.4.0's RealSchur / EigenSolver, transcribed
constexpr int kEigenFormClosed = 1;   // d + p +- z

// JacobiRotation<double>::makeGivens(p, q)
NDT2D_HD void make_givens(double p, double q, double * c, double * s)
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

// m = [[m00, m01], [m10, m11]]; e0, e1 = EigenSolver<Matrix2d>(m).eigenvalues().real()
NDT2D_HD void eigen_solver_2x2(double m00, double m01, double m10, double m11, double * e0, double * e1)
{
  const double eps = 2.220446049250313e-16;         // NumTraits<double>::epsilon()
  const double tiny_min = 2.2250738585072014e-308;  // std::numeric_limits<double>::min()
  // RealSchur::compute: scale = matrix.cwiseAbs().maxCoeff()
  double scale = fabs(m00);
  if (fabs(m10) > scale) scale = fabs(m10);   // (column-major visiting order; a maximum either way)
  if (fabs(m01) > scale) scale = fabs(m01);
  if (fabs(m11) > scale) scale = fabs(m11);
  if (scale < tiny_min)
  {
    *e0 = 0.0;   // m_matT.setZero()
    *e1 = 0.0;
    return;
  }
  double t00 = m00 / scale, t01 = m01 / scale, t10 = m10 / scale, t11 = m11 / scale;
  // computeNormOfT: column by column, the entries on and above the sub-diagonal
  double norm = 0.0;
  norm += fabs(t00) + fabs(t10);
  norm += fabs(t01) + fabs(t11);
  if (norm != 0.0)
  {
    double consider_as_zero = norm * (eps * eps);
    if (consider_as_zero < tiny_min) consider_as_zero = tiny_min;   // numext::maxi
    // findSmallSubdiagEntry(iu = 1)
    double s = fabs(t00) + fabs(t11);
    s = s * eps;
    if (s < consider_as_zero) s = consider_as_zero;
    if (fabs(t10) <= s)
    {
      t10 = 0.0;   // two single roots: the diagonal
    }
    else
    {
      // splitOffTwoRows(iu = 1, exshift = 0)
      const double p = 0.5 * (t00 - t11);
      const double q = p * p + t10 * t01;
      if (q >= 0.0)
      {
        const double z = sqrt(fabs(q));
        double c, sn;
        if (p >= 0.0) make_givens(p + z, t10, &c, &sn);
        else make_givens(p - z, t10, &c, &sn);
        // m_matT.applyOnTheLeft(0, 1, rot.adjoint()): j = (c, -s); rows 0 and 1, every column
        const double jc = c, js = -sn;
        if (!(jc == 1.0 && js == 0.0))
        {
          double x = t00, y = t10;
          t00 = jc * x + js * y;
          t10 = -js * x + jc * y;
          x = t01;
          y = t11;
          t01 = jc * x + js * y;
          t11 = -js * x + jc * y;
          // m_matT.applyOnTheRight(0, 1, rot): apply_rotation_in_the_plane(col 0, col 1, rot.transpose() = (c, -s))
          x = t00;
          y = t01;
          t00 = jc * x + js * y;
          t01 = -js * x + jc * y;
          x = t10;
          y = t11;
          t10 = jc * x + js * y;
          t11 = -js * x + jc * y;
        }
        t10 = 0.0;
      }
    }
  }
  // m_matT *= scale
  t00 *= scale;
  t10 *= scale;
  t11 *= scale;
  // EigenSolver::compute
  if (t10 == 0.0)
  {
    *e0 = t00;
    *e1 = t11;
  }
  else
  {
    const double p = 0.5 * (t00 - t11);   // a complex pair: the real part of both
    *e0 = t11 + p;
    *e1 = t11 + p;
  }
}

// The closed form of rounds 1-4: deflation test as above (unscaled), else d + p +- z.
NDT2D_HD void closed_form_eigenvalues(double a, double b, double d, double * e0, double * e1)
{
  const double eps = 2.220446049250313e-16;
  const double tiny_min = 2.2250738585072014e-308;
  const double norm = fabs(a) + 2.0 * fabs(b) + fabs(d);
  double tiny = norm * (eps * eps);
  if (tiny < tiny_min) tiny = tiny_min;
  double thresh = (fabs(a) + fabs(d)) * eps;
  if (thresh < tiny) thresh = tiny;
  if (norm == 0.0 || fabs(b) <= thresh)
  {
    *e0 = a;
    *e1 = d;
    return;
  }
  const double p = 0.5 * (a - d);
  const double q = p * p + b * b;
  const double z = sqrt(fabs(q));
  *e0 = (d + p) + z;
  *e1 = (d + p) - z;
}

// Cell::compute needs the eigenvalues for ONE decision -- `small < 0.001 * large` (src/ndt_model.cpp:88)
// -- and, only when that holds, for the clamp branch's determinant (:91).  For a covariance that is
// clear of the threshold the decision is known without them, whichever form would compute them:
// with t = a + d and det = a d - b b the exact eigenvalues l1 <= l2 have r = l1 / l2 with
// r / (1 + r)^2 = det / t^2 =: q, increasing in r.  The computed det is within eps (a d + b b) <= eps t^2 / 2
// of the exact one -- a relative 5e2 eps at q = 0.001 -- t^2 within three ulps, and either form
// returns both eigenvalues within a few tens of ulps OF THE LARGER ONE, i.e. `small` within 1e-12 of
// itself at this conditioning.  q >= 0.0011 means r >= 0.0011024: a tenth above the threshold, eleven
// orders of magnitude more than the forms' ulps.  True = "the test at :88 is false, the branch at :99
// is taken"; false = compute them.
// (Round 6: the transcribed EigenSolver is 20-34 ns per cell on the host -- 9 us of every addScans of
// the mapper's cycle.  A lidar map's cells are walls, many seen at a grazing angle and thin: with the
// first version's limit, 0.004, a third of the 41 x 41 map's cells and half of the 245 x 245 map's
// still went through the solver -- 93 % / 53 % of those with q in [0.0011, 0.004).  What still does:
// the cells inside the clamp, whose determinant takes the solver's `large` bit for bit, and a band
// of a tenth above it.  tests/cpp/eigen_screen_check.cpp samples that band densely.)
NDT2D_HD bool clamp_test_surely_false(double a, double b, double d)
{
  const double t = a + d;
  const double tt = t * t;
  const double det = a * d - b * b;
  // (t^2 inside the normal range: no product above has overflowed or lost bits to underflow)
  return a > 0.0 && d > 0.0 && tt < 1.0e300 && tt > 1.0e-280 && det >= 0.0011 * tt;
}

// Eigenvalues of the symmetric covariance [[a, b], [b, d]] in the chosen form.
NDT2D_HD void covariance_eigenvalues(int form, double a, double b, double d, double * e0, double * e1)
{
  if (form == kEigenFormClosed) closed_form_eigenvalues(a, b, d, e0, e1);
  else eigen_solver_2x2(a, b, b, d, e0, e1);
}

}  // namespace ndt2d

#endif  // NDT2D_EIGEN2_H_
