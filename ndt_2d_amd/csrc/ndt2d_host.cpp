// Host side of libndt2d_hip.so (include/ndt2d_hip.h, section 2 + generator):
//   * HostNdt: the NDT build of ScanMatcherNDT::addScans on the host, with the
//     reference's incremental formulas and point order (it produces the kernels'
//     input and must be bit-faithful),
//   * ndt2d_matcher_*: the reference's ScanMatcherNDT object restated over the
//     device layer -- subsampling, search tables, final covariance formula,
//   * the batched particle path (ParticleFilter::measure),
//   * the synthetic workload generator.
// Scoring arithmetic on the HOST exists in exactly two places, both by design
// and neither a fallback -- a matcher cannot be created without a GPU: (1) one pose of a scan
// of at most 256 subsampled beams (scorePoints / scoreScan as the unchanged
// ParticleFilter::measure calls them, once per particle: a launch + PCIe round trip per call
// costs 25x the arithmetic) is scored by the calling thread from the host copy of the NDT,
// host_score_points(); (2) the few candidates of a marked near-tie are rescored in the
// reference's own arithmetic, settle_near_tie().  Every search, every batch of poses and every
// longer scan is evaluated on the GPU.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include <atomic>
#include <chrono>

#include "ndt2d_eigen2.h"
#include "ndt2d_exchange.h"
#include "ndt2d_guard.h"
#include "ndt2d_hip.h"
#include "ndt2d_workers.h"

namespace
{

// cos(t) and sin(t) of one argument, as a GCC-built reference gets them: wherever
// the reference writes the pair, GCC merges the two calls into one glibc sincos(),
// whose sine can differ from sin()'s in the last ulp (t = 0.4710119964311561).
// (Same helper as in ndt2d_kernels.h; this file stays free of HIP headers.)
inline void ndt2d_cos_sin(double t, double * c, double * s)
{
  sincos(t, s, c);
}

// ---------------------------------------------------------------------------
// Host NDT build
// ---------------------------------------------------------------------------

// One NDT cell; fields as ndt_2d::Cell (reference include/ndt_2d/ndt_model.hpp:43-65).
// Symmetric 2x2 matrices keep {xx, xy, yy}; correlation(1,0) is never written
// by the reference and never read.
struct HostCell
{
  bool valid = false;
  double n = 0.0;
  double mean_x = 0.0, mean_y = 0.0;
  double corr_xx = 0.0, corr_xy = 0.0, corr_yy = 0.0;
  double cov_xx = 0.0, cov_xy = 0.0, cov_yy = 0.0;
  double info_xx = 0.0, info_xy = 0.0, info_yy = 0.0;

  // Cell::addPoint, reference src/ndt_model.cpp:50-63.  The five running values are
  // consecutive doubles updated by one expression shape, (v * n + t) / n1: the AVX2
  // clone of NDT::addScan below does them as packed operations (each lane the same
  // IEEE operation as the scalar code, so the results are bit-identical).
  __attribute__((always_inline)) void add(double x, double y)
  {
    typedef double v4d __attribute__((vector_size(32)));
    const double n1 = n + 1;
    v4d v;
    std::memcpy(&v, &mean_x, sizeof(v));             // mean_x, mean_y, corr_xx, corr_xy
    const v4d t = {x, y, x * x, x * y};
    v = (v * n + t) / n1;
    std::memcpy(&mean_x, &v, sizeof(v));
    corr_yy = (corr_yy * n + y * y) / n1;
    n += 1;
    valid = false;
  }

  // Cell::compute, reference src/ndt_model.cpp:65-103.  eigen_form: how the eigenvalues of
  // :84-85 are formed (ndt2d_eigen2.h: Eigen 3.4.0's EigenSolver transcribed, or the closed form).
  void compute(int eigen_form)
  {
    if (valid || n < 3) return;
    const double scale = n / (n - 1);
    cov_xx = (corr_xx - (mean_x * mean_x)) * scale;
    cov_xy = (corr_xy - (mean_x * mean_y)) * scale;
    cov_yy = (corr_yy - (mean_y * mean_y)) * scale;

    // (the eigenvalues decide the branch and feed the clamp's determinant: a covariance far from
    // the threshold does not need them, ndt2d_eigen2.h clamp_test_surely_false)
    double small = 1.0, large = 1.0;
    if (!ndt2d::clamp_test_surely_false(cov_xx, cov_xy, cov_yy))
    {
      ndt2d::covariance_eigenvalues(eigen_form, cov_xx, cov_xy, cov_yy, &small, &large);
      if (small > large) std::swap(small, large);
    }
    if (small < 0.001 * large)
    {
      // eigenvalue clamp (:88-96)
      const double determinant = (0.001 * large) * large;
      info_xx = cov_yy / determinant;
      info_xy = -cov_xy / determinant;
      info_yy = cov_xx / determinant;
    }
    else
    {
      // Matrix2d::inverse() (:99): adjugate times 1/det
      const double det = cov_xx * cov_yy - cov_xy * cov_xy;
      const double invdet = 1.0 / det;
      info_xx = cov_yy * invdet;
      info_xy = -cov_xy * invdet;
      info_yy = cov_xx * invdet;
    }
    valid = true;
  }
};

// class NDT (reference include/ndt_2d/ndt_model.hpp:67-134), build side only.
class HostNdt
{
public:
  // NDT::NDT, reference src/ndt_model.cpp:118-126
  HostNdt(double cell_size, double size_x, double size_y, double origin_x, double origin_y)
  {
    reset(cell_size, size_x, size_y, origin_x, origin_y);
  }

  // A new, empty NDT in this object's storage: the mapper rebuilds its local NDT for
  // every scan (src/ndt_mapper.cpp:508-509) with the same geometry more often than not,
  // and of its cells only the few hundred that received points need clearing.
  void reset(double cell_size, double size_x, double size_y, double origin_x, double origin_y)
  {
    reset_cells(cell_size, static_cast<size_t>((size_x / cell_size) + 1), static_cast<size_t>((size_y / cell_size) + 1),
                origin_x, origin_y);
  }

  // ... given its size in cells
  void reset_cells(double cell_size, size_t sx, size_t sy, double origin_x, double origin_y)
  {
    // The storage is a pool of cells of which only the touched ones are not in their initial
    // state: clearing those makes it an empty grid of ANY geometry that fits (the extent
    // follows the scan poses, so its size changes by a cell now and then; a real lidar's
    // grid is tens of thousands of cells, 6 MB, of which a scan touches a thousand).
    for (const uint32_t i : touched_) cells_[i] = HostCell();
    if (cells_.size() < sx * sy + 1) cells_.resize(sx * sy + 1);   // (+ 1: add_scan's scratch cell)
    n_cells_ = sx * sy;
    touched_.clear();
    {
      int e = 0;
      pow2_ = cell_size > 0.0 && std::isfinite(cell_size) && std::frexp(cell_size, &e) == 0.5 &&
              std::fpclassify(cell_size) == FP_NORMAL && std::fpclassify(1.0 / cell_size) == FP_NORMAL;
      inv_cell_size_ = 1.0 / cell_size;
    }
    cell_size_ = cell_size;
    size_x_ = sx;
    size_y_ = sy;
    origin_x_ = origin_x;
    origin_y_ = origin_y;
  }

  // NDT::getIndex, reference src/ndt_model.cpp:203-218
  __attribute__((always_inline)) long index(double x, double y) const
  {
    if (x < origin_x_ || y < origin_y_) return -1;
    // (a power-of-two cell size: multiplying by its exact reciprocal is the correctly
    // rounded quotient, bit-identical to the reference's divide)
    const double fx = pow2_ ? (x - origin_x_) * inv_cell_size_ : (x - origin_x_) / cell_size_;
    const double fy = pow2_ ? (y - origin_y_) * inv_cell_size_ : (y - origin_y_) / cell_size_;
    const unsigned int gx = static_cast<unsigned int>(static_cast<long long>(fx));
    const unsigned int gy = static_cast<unsigned int>(static_cast<long long>(fy));
    if (gx >= size_x_ || gy >= size_y_) return -1;
    return static_cast<long>(gy * size_x_ + gx);
  }

  // NDT::addScan, reference src/ndt_model.cpp:132-152.
  //
  // Cell::addPoint is a recurrence -- v = (v * n + t) / (n + 1) on five running values -- and
  // consecutive beams of a scan fall into the same cell more often than not: one dependent chain
  // of multiply, add, DIVIDE and a store-to-load round trip per point (~20 cycles; the whole of
  // addScans' host time, 4.3 ns per point on the GPU box's EPYC 9575F).  Round 6: a scan is cut
  // into four quarters of consecutive beams and the quarters advance side by side -- four chains
  // in flight, and the four corr_yy updates of a step share ONE packed divide (five 256-bit
  // divides per four points instead of eight divide operations).  A cell's values depend on the
  // ORDER of its points (the reference's: scan after scan, beam after beam): pass 1 transforms the
  // points and looks their cells up (branch-free, vectorised by the compiler), then stamps every
  // cell with the first quarter of the scan that reaches it; the points of a LATER quarter in
  // such a cell (the cell a quarter boundary falls into; a robot boxed in closer than a cell) are
  // taken out of their quarter and added behind the quarters, in beam order.  So every cell
  // still receives its points in the reference's order, and every lane of a packed operation is
  // the IEEE operation of the scalar code: bit-identical cells (tests/test_host_logic.py, and
  // every host-build == oracle test).  Points outside the grid go to a scratch cell behind it.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
  __attribute__((target_clones("avx2", "default")))
#endif
  void add_scan(double pose_x, double pose_y, double pose_theta, const double * pts, size_t n)
  {
    typedef double v4d __attribute__((vector_size(32)));
    double cos_th, sin_th;
    ndt2d_cos_sin(pose_theta, &cos_th, &sin_th);  // :135-136
    const size_t quarter = n / 4;
    // (shorter scans: nothing to gain.  Nor on a grid whose cells do not stay in the host's L2:
    // there the sequential loop's cache misses overlap by themselves and the stamps only add
    // to them -- GPU box's EPYC 9575F, nine 720-beam scans: 41 x 41 cells 24.3 -> 20.8 us side by
    // side, 245 x 245 cells 17.3 -> 20.5 us)
    const bool side_by_side = quarter >= 8 && interleave_ && n_cells_ * sizeof(HostCell) <= side_by_side_max_bytes_;
    if (!side_by_side)
    {
      // the reference's loop as it stands
      for (size_t k = 0; k < n; ++k)
      {
        const double px = pts[2 * k], py = pts[2 * k + 1];
        double wx = pose_x;
        double wy = pose_y;
        wx += px * cos_th - py * sin_th;
        wy += px * sin_th + py * cos_th;
        const long i = index(wx, wy);
        if (i >= 0)
        {
          HostCell & c = cells_[static_cast<size_t>(i)];
          if (c.n == 0.0) touched_.push_back(static_cast<uint32_t>(i));
          c.add(wx, wy);
        }
      }
      return;
    }
    if (scan_idx_.size() < n)
    {
      scan_idx_.resize(n);
      scan_xy_.resize(2 * n);
    }
    if (cells_.size() < n_cells_ + 1) cells_.resize(n_cells_ + 1);
    if (stamp_.size() < cells_.size()) stamp_.resize(cells_.size(), 0u);
    if (epoch_ > 0xfffffff0u)
    {
      std::fill(stamp_.begin(), stamp_.end(), 0u);
      epoch_ = 0;
    }
    const uint32_t scan_first = epoch_ + 1;
    epoch_ += 4;
    int32_t * const idx = scan_idx_.data();
    double * const xy = scan_xy_.data();
    const int32_t outside = static_cast<int32_t>(n_cells_);   // the scratch cell
    // pass 1a: points_world (:138-143) and NDT::getIndex (:203-218).  For a point that passed
    // `x >= origin`, trunc(f) < size  <=>  f < size, so the four comparisons are getIndex's.
    // (Everything the loop reads of *this is copied out first: its stores could alias members.)
    {
      const double ox = origin_x_, oy = origin_y_, inv = inv_cell_size_, cs = cell_size_;
      const double fsx = static_cast<double>(size_x_), fsy = static_cast<double>(size_y_);
      const int32_t sx = static_cast<int32_t>(size_x_);
      if (pow2_)
      {
        for (size_t k = 0; k < n; ++k)
        {
          const double px = pts[2 * k], py = pts[2 * k + 1];
          double wx = pose_x;
          double wy = pose_y;
          wx += px * cos_th - py * sin_th;
          wy += px * sin_th + py * cos_th;
          xy[2 * k] = wx;
          xy[2 * k + 1] = wy;
          const double fx = (wx - ox) * inv, fy = (wy - oy) * inv;   // (exact reciprocal: == the divide)
          const bool in = (wx >= ox) & (wy >= oy) & (fx < fsx) & (fy < fsy);
          const int32_t gx = static_cast<int32_t>(in ? fx : 0.0), gy = static_cast<int32_t>(in ? fy : 0.0);
          idx[k] = in ? gy * sx + gx : outside;
        }
      }
      else
      {
        for (size_t k = 0; k < n; ++k)
        {
          const double px = pts[2 * k], py = pts[2 * k + 1];
          double wx = pose_x;
          double wy = pose_y;
          wx += px * cos_th - py * sin_th;
          wy += px * sin_th + py * cos_th;
          xy[2 * k] = wx;
          xy[2 * k + 1] = wy;
          const double fx = (wx - ox) / cs, fy = (wy - oy) / cs;
          const bool in = (wx >= ox) & (wy >= oy) & (fx < fsx) & (fy < fsy);
          const int32_t gx = static_cast<int32_t>(in ? fx : 0.0), gy = static_cast<int32_t>(in ? fy : 0.0);
          idx[k] = in ? gy * sx + gx : outside;
        }
      }
    }
    // pass 1b: first touches (in beam order: `touched_` keeps the order the sequential loop gave
    // it) and the quarter stamps.  A point whose cell an EARLIER quarter of this scan has reached
    // (the cell a quarter boundary falls into, mostly) leaves its quarter: it is added after the
    // quarters, in beam order -- every point of that cell from the later quarter does, so the
    // cell still sees its points in the reference's order.  The n % 4 beams behind the fourth
    // quarter go the same way.
    // (Written with selects instead of branches -- stamp, first touch and late list stored for every
    // point -- the loop is SLOWER: EPYC 9575F, toy map 19.4 -> 26.8 us per addScans; consecutive beams
    // share cells, and a stamp stored for every point is a store-to-load chain through that cell.)
    size_t n_late = 0;
    {
      const size_t n_touched_before = touched_.size();
      touched_.resize(n_touched_before + n);
      if (late_.size() < n) late_.resize(n);
      uint32_t * touched_out = touched_.data() + n_touched_before;
      uint32_t * const stamp = stamp_.data();
      uint32_t * const late = late_.data();
      const HostCell * const cells = cells_.data();
      for (size_t part = 0; part < 4; ++part)
      {
        const size_t k_end = part == 3 ? n : (part + 1) * quarter;
        const uint32_t mine = scan_first + static_cast<uint32_t>(part);
        for (size_t k = part * quarter; k < k_end; ++k)
        {
          const int32_t i = idx[k];
          if (i == outside) continue;
          const uint32_t seen = stamp[i];
          if (seen < scan_first)
          {
            stamp[i] = mine;
            *touched_out = static_cast<uint32_t>(i);
            touched_out += cells[i].n == 0.0 ? 1 : 0;
          }
          if ((seen >= scan_first && seen != mine) || k >= 4 * quarter)
          {
            late[n_late++] = static_cast<uint32_t>(k);
          }
        }
      }
      touched_.resize(static_cast<size_t>(touched_out - touched_.data()));
    }
    // pass 2: Cell::addPoint (:50-63)
    HostCell * const cells = cells_.data();
    {
      const size_t q = quarter;
      // (the late points step aside: their quarter adds to the scratch cell in their place)
      if (late_cell_.size() < n_late) late_cell_.resize(n_late);
      for (size_t l = 0; l < n_late; ++l)
      {
        const uint32_t k = late_[l];
        late_cell_[l] = idx[k];
        idx[k] = outside;
      }
      for (size_t j = 0; j < q; ++j)
      {
        HostCell & c0 = cells[idx[j]];
        HostCell & c1 = cells[idx[j + q]];
        HostCell & c2 = cells[idx[j + 2 * q]];
        HostCell & c3 = cells[idx[j + 3 * q]];
        const double x0 = xy[2 * j], y0 = xy[2 * j + 1];
        const double x1 = xy[2 * (j + q)], y1 = xy[2 * (j + q) + 1];
        const double x2 = xy[2 * (j + 2 * q)], y2 = xy[2 * (j + 2 * q) + 1];
        const double x3 = xy[2 * (j + 3 * q)], y3 = xy[2 * (j + 3 * q) + 1];
        const v4d nn = {c0.n, c1.n, c2.n, c3.n};
        const v4d n1 = nn + 1.0;
        v4d yy = {c0.corr_yy, c1.corr_yy, c2.corr_yy, c3.corr_yy};
        const v4d ty = {y0 * y0, y1 * y1, y2 * y2, y3 * y3};
        yy = (yy * nn + ty) / n1;
        v4d v0, v1, v2, v3;
        std::memcpy(&v0, &c0.mean_x, sizeof(v4d));     // mean_x, mean_y, corr_xx, corr_xy
        std::memcpy(&v1, &c1.mean_x, sizeof(v4d));
        std::memcpy(&v2, &c2.mean_x, sizeof(v4d));
        std::memcpy(&v3, &c3.mean_x, sizeof(v4d));
        const v4d t0 = {x0, y0, x0 * x0, x0 * y0}, t1 = {x1, y1, x1 * x1, x1 * y1};
        const v4d t2 = {x2, y2, x2 * x2, x2 * y2}, t3 = {x3, y3, x3 * x3, x3 * y3};
        v0 = (v0 * nn[0] + t0) / n1[0];
        v1 = (v1 * nn[1] + t1) / n1[1];
        v2 = (v2 * nn[2] + t2) / n1[2];
        v3 = (v3 * nn[3] + t3) / n1[3];
        std::memcpy(&c0.mean_x, &v0, sizeof(v4d));
        std::memcpy(&c1.mean_x, &v1, sizeof(v4d));
        std::memcpy(&c2.mean_x, &v2, sizeof(v4d));
        std::memcpy(&c3.mean_x, &v3, sizeof(v4d));
        c0.corr_yy = yy[0];
        c1.corr_yy = yy[1];
        c2.corr_yy = yy[2];
        c3.corr_yy = yy[3];
        c0.n = n1[0];
        c1.n = n1[1];
        c2.n = n1[2];
        c3.n = n1[3];
        c0.valid = c1.valid = c2.valid = c3.valid = false;
      }
      for (size_t l = 0; l < n_late; ++l)
      {
        const uint32_t k = late_[l];
        cells[late_cell_[l]].add(xy[2 * k], xy[2 * k + 1]);
      }
    }
    cells[outside] = HostCell();
  }

  // (tests: the sequential order for every scan -- the two must agree bit for bit)
  void set_interleave(bool on) { interleave_ = on; }
  void set_side_by_side_max_bytes(size_t bytes) { side_by_side_max_bytes_ = bytes; }   // (experiments)

  // NDT::likelihood(Vector2d) (reference src/ndt_model.cpp:162-170) with Cell::score (:105-116)
  // inlined: exp(((-0.5 * q^T) * information) * q) in that order, libm's exp.  Used by the
  // single-pose calls (scorePoints once per particle, src/particle_filter.cpp:81-87) and by the
  // adjudication of near-ties -- never by a search or a batch.
  __attribute__((always_inline)) double likelihood(double x, double y) const
  {
    const long i = index(x, y);
    if (i < 0) return 0.0;
    const HostCell & c = cells_[static_cast<size_t>(i)];
    if (c.n < 5) return 0.0;
    const double q0 = x - c.mean_x, q1 = y - c.mean_y;
    const double a0 = -0.5 * q0, a1 = -0.5 * q1;
    const double r0 = a0 * c.info_xx + a1 * c.info_xy;
    const double r1 = a0 * c.info_xy + a1 * c.info_yy;
    return std::exp(r0 * q0 + r1 * q1);
  }

  // A grid from its packed records (a grid that was built on the device, fetched back once).
  void load6(const double * cells6)
  {
    for (size_t i = 0; i < n_cells_; ++i)
    {
      const double * r = cells6 + 6 * i;
      HostCell & c = cells_[i];
      if (r[5] == 0.0) continue;
      c.mean_x = r[0];
      c.mean_y = r[1];
      c.info_xx = r[2];
      c.info_xy = r[3];
      c.info_yy = r[4];
      c.n = r[5];
      touched_.push_back(static_cast<uint32_t>(i));
    }
  }

  // NDT::compute, reference src/ndt_model.cpp:154-160 (a cell without points returns
  // at once there: only the cells that received points are visited here)
  void compute(int eigen_form)
  {
    for (const uint32_t i : touched_) cells_[i].compute(eigen_form);
  }

  // The cells that hold points, as ndt2d_set_grid_sparse takes them.
  size_t n_touched() const { return touched_.size(); }
  void sparse6(uint32_t * index, double * cells6) const
  {
    for (size_t k = 0; k < touched_.size(); ++k)
    {
      index[k] = touched_[k];
      const HostCell & c = cells_[touched_[k]];
      double * out = cells6 + 6 * k;
      out[0] = c.mean_x;
      out[1] = c.mean_y;
      out[2] = c.info_xx;
      out[3] = c.info_xy;
      out[4] = c.info_yy;
      out[5] = c.n;
    }
  }

  void pack6(double * out) const
  {
    for (size_t i = 0; i < n_cells_; ++i)
    {
      const HostCell & c = cells_[i];
      out[6 * i + 0] = c.mean_x;
      out[6 * i + 1] = c.mean_y;
      out[6 * i + 2] = c.info_xx;
      out[6 * i + 3] = c.info_xy;
      out[6 * i + 4] = c.info_yy;
      out[6 * i + 5] = c.n;
    }
  }

  double cell_size() const { return cell_size_; }
  size_t size_x() const { return size_x_; }
  size_t size_y() const { return size_y_; }
  double origin_x() const { return origin_x_; }
  double origin_y() const { return origin_y_; }
  size_t ncell() const { return n_cells_; }

private:
  double cell_size_ = 0.0, inv_cell_size_ = 0.0;
  bool pow2_ = false;
  size_t size_x_ = 0, size_y_ = 0;
  double origin_x_ = 0.0, origin_y_ = 0.0;
  std::vector<HostCell> cells_;     // a pool: the first n_cells_ are the grid
  size_t n_cells_ = 0;
  std::vector<uint32_t> touched_;   // cells that hold at least one point
  // add_scan: world points and cell indices of the scan being added; per-cell stamp = the
  // (scan, quarter) that last reached the cell (ids from a running counter: never cleared)
  std::vector<double> scan_xy_;
  std::vector<int32_t> scan_idx_;
  std::vector<uint32_t> stamp_;
  std::vector<uint32_t> late_;      // beams of the scan that are added after the quarters ...
  std::vector<int32_t> late_cell_;  // ... and their cells
  uint32_t epoch_ = 0;
  bool interleave_ = true;
  size_t side_by_side_max_bytes_ = 1u << 20;
};

// ScanMatcherNDT::addScans' extent + NDT build, reference src/scan_matcher_ndt.cpp:49-74.
// max_x_/max_y_ start at numeric_limits<double>::min(), as the reference has it.
std::unique_ptr<HostNdt> build_ndt(double resolution, double range_max, const double * poses,
                                   const double * pts, const size_t * offsets, size_t n_scans,
                                   std::unique_ptr<HostNdt> reuse = nullptr, int eigen_form = ndt2d::kEigenFormSchur,
                                   bool side_by_side = true)
{
  double min_x = std::numeric_limits<double>::max();
  double max_x = std::numeric_limits<double>::min();
  double min_y = std::numeric_limits<double>::max();
  double max_y = std::numeric_limits<double>::min();
  for (size_t k = 0; k < n_scans; ++k)
  {
    min_x = std::min(poses[3 * k] - range_max, min_x);
    max_x = std::max(poses[3 * k] + range_max, max_x);
    min_y = std::min(poses[3 * k + 1] - range_max, min_y);
    max_y = std::max(poses[3 * k + 1] + range_max, max_y);
  }
  // NDT::NDT (src/ndt_model.cpp:118-126) sizes the grid (size_t)(extent / cell_size + 1) per axis:
  // a pose of 1e15 or a NaN makes that a count no allocation can serve (or, cast from NaN, undefined
  // behaviour).  Refused here, before any storage is asked for: nullptr.
  {
    const double fsx = ((max_x - min_x) / resolution) + 1, fsy = ((max_y - min_y) / resolution) + 1;
    if (!(fsx >= 1.0) || !(fsy >= 1.0) || !(fsx * fsy < 2147483648.0)) return nullptr;
  }
  std::unique_ptr<HostNdt> ndt = std::move(reuse);
  if (ndt)
  {
    ndt->reset(resolution, (max_x - min_x), (max_y - min_y), min_x, min_y);
  }
  else
  {
    ndt.reset(new HostNdt(resolution, (max_x - min_x), (max_y - min_y), min_x, min_y));
  }
  ndt->set_interleave(side_by_side);
  for (size_t k = 0; k < n_scans; ++k)
  {
    ndt->add_scan(poses[3 * k], poses[3 * k + 1], poses[3 * k + 2], pts + 2 * offsets[k],
                  offsets[k + 1] - offsets[k]);
  }
  ndt->compute(eigen_form);
  return ndt;
}

// The reference's `for (v = -size; v < size; v += res)` (src/scan_matcher_ndt.cpp:103,117,119):
// the visited values come from repeated floating-point addition.
std::vector<double> search_offsets(double size, double res)
{
  std::vector<double> out;
  if (!(res > 0.0))
  {
    if (-size < size) out.push_back(-size);  // the reference would never terminate
    return out;
  }
  for (double v = -size; v < size; v += res) out.push_back(v);
  return out;
}

// Whether search_offsets(size, res) ends and stays within `limit` values.
bool offsets_fit(double size, double res, size_t limit)
{
  if (!std::isfinite(size) || !std::isfinite(res)) return false;
  if (!(res > 0.0)) return true;                  // (search_offsets: at most one value)
  if (!(size > 0.0)) return true;                 // (-size < size fails at once: no value)
  return 2.0 * size / res <= static_cast<double>(limit) - 2.0;
}

// Subsampling of matchScan / scorePoints, reference src/scan_matcher_ndt.cpp:95-96,110.
void subsample_into(std::vector<double> & out, const double * pts, size_t n_points,
                    size_t laser_max_beams)
{
  const size_t use = std::min(laser_max_beams, n_points);
  out.resize(2 * use);
  if (use == 0) return;
  const double scan_step = static_cast<double>(n_points) / use;
  for (size_t i = 0; i < use; ++i)
  {
    const size_t idx = static_cast<size_t>(i * scan_step);
    out[2 * i] = pts[2 * idx];
    out[2 * i + 1] = pts[2 * idx + 1];
  }
}

// ROS angles::normalize_angle / shortest_angular_distance (unpinned dependency
// of the reference, used by updateStatistics src/particle_filter.cpp:215).
double normalize_angle(double a)
{
  const double r = std::fmod(a + M_PI, 2.0 * M_PI);
  return r <= 0.0 ? r + M_PI : r - M_PI;
}

}  // namespace

// What one device of a multi-device matcher keeps for the sharded calls (device memory of
// ITS GPU, sized on demand).
struct MatcherShard
{
  double * d_table = nullptr;     // [n_dev][12] record / [n_dev][8] moment table of the exchange
  double * d_sum = nullptr;       // 8 moment sums (all devices) | 8 statistics of this device
  double * d_scores = nullptr;    // per-candidate scores of this device's theta steps (optional)
  size_t scores_cap = 0;
  double * d_poses = nullptr;     // this device's particle range
  double * d_weights = nullptr;
  size_t poses_cap = 0;
  uint64_t beams_epoch = ~0ull;   // ndt2d_matcher::beams_epoch of the beams ndt2d_set_beams put there
};

struct ndt2d_matcher
{
  ndt2d_handle dev = nullptr;            // == devs[0]: every single-pose and small call runs here
  std::vector<ndt2d_handle> devs;        // one context per entry of device_ids
  std::vector<int> device_ids;
  std::vector<MatcherShard> shards;
  int exchange_mode = 0;                 // 0 auto, 1 host, 2 rccl (ndt2d_matcher_set_exchange)
  ndt2d::Exchange * exchange = nullptr;  // RCCL communicators, made when first needed
  bool exchange_tried = false;
  std::string exchange_note;             // why "auto" did not take RCCL
  // Work below these stays on the first device: candidates x beams of a search (~0.3 ms of one
  // GPU) and particles x beams of a batch (~0.2 ms; BASELINE configs[4], 7.2e8, is above it).
  // Dealing costs the call ~25 us over its slowest share (profiles/r05_multi_device_summary.json).
  double multi_min_units = 1.0e9;
  double multi_min_pose_units = 2.0e8;
  std::unique_ptr<ndt2d::DeviceWorkers> workers;   // one thread per device beyond the first
  std::vector<double> fanout_us;         // last dealt call: when each device's launch was queued
  uint64_t beams_epoch = 0;              // counts the changes of `beams` (what the other devices hold: MatcherShard)
  double * pinned = nullptr;             // host block of the exchanges (layout: multi_pinned_*)
  std::string variant;                   // ndt2d_matcher_last_variant
  bool last_multi = false;
  std::string err;
  // the six declared parameters, reference src/scan_matcher_ndt.cpp:37-44
  double resolution = 0.25;
  double angular_res = 0.0025, angular_size = 0.1;
  double linear_res = 0.005, linear_size = 0.05;
  size_t laser_max_beams = 100;
  double range_max = 0.0;
  std::unique_ptr<HostNdt> ndt;   // host copy; empty when the NDT was built on the device
  std::unique_ptr<HostNdt> spare; // the storage of the NDT that reset() dropped, for the next build
  bool have_ndt = false;          // `ndt_` is set (reference scan_matcher_ndt.hpp:102)
  int build_mode = 0;             // 0 auto, 1 host, 2 device
  int eigen_form = ndt2d::kEigenFormSchur;   // ndt2d_matcher_set_eigenvalue_form
  // state of the last prepare_search (subsampled beams + visited offsets)
  std::vector<double> beams, dth, dlin;
  size_t n_use = 0;               // beams in use (the N of `best / N`, :148)
  bool search_ready = false;
  // `beams` is what the device context currently holds as its beams: a scoring call
  // that arrives with the same points again (the unchanged ParticleFilter::measure
  // calls scorePoints once per particle with one scan, src/particle_filter.cpp:81-87;
  // the mapper calls scoreScan and matchScan on one scan, src/ndt_mapper.cpp:514-515)
  // skips the upload.  The matcher must be the only writer of its context's beams.
  bool beams_on_device = false;
  std::vector<double> scratch_beams, cos_th, sin_th;
  // The mapper calls scoreScan(scan) and then matchScan(scan, ...) (reference
  // src/ndt_mapper.cpp:514-515, 552-553).  Once that pair has been seen, scoreScan queues the
  // scan's search behind its own kernel before it waits for the score (`ahead`): the search
  // then starts when the scoring kernel ends, not a host round trip later, and the matchScan
  // that follows only collects it.  A call that is not that matchScan waits the search out,
  // discards it, and scoreScan stops doing it until the pair is seen again.
  bool pair_seen = false;         // the last matchScan was of the scan and pose of the scoreScan before it
  bool score_scan_last = false;   // the previous device call was a scoreScan ...
  double score_scan_pose[3] = {0.0, 0.0, 0.0};   // ... from this pose (its beams are `beams`)
  bool ahead = false;             // a search launched by scoreScan has not been collected
  double ahead_pose[3] = {0.0, 0.0, 0.0};
  size_t ahead_n_th = 0;
  uint64_t ahead_launch_id = 0, ahead_fetch_id = 0;   // ndt2d_match_status right after that launch
  // One pose at a time (scorePoints, scoreScan): scored on the host from the host NDT when the
  // scan is short (ndt2d_matcher_set_single_pose_path).  `scored` = the subsampled beams of the
  // last scoreScan (the pair detection above compares the next matchScan's scan with it).
  // near-tie adjudication (ndt2d_matcher_set_adjudication)
  bool adjudicate = true;
  uint64_t adj_marked = 0, adj_changed = 0, adj_truncated = 0;
  bool single_pose_host = true;
  size_t single_pose_max_beams = 256;
  std::unique_ptr<HostNdt> fetched;  // host copy of a grid that was built on the device
  std::vector<double> scored;
  int ahead_enabled = 1;          // ndt2d_matcher_set_search_ahead
  uint64_t ahead_launched = 0, ahead_collected = 0;
};

namespace
{

// ndt2d_guard.h: where the text of an exception caught at the C boundary goes
void guard_note(ndt2d_matcher * m, const char * what) noexcept
{
  if (m == nullptr) return;
  try
  {
    m->err = what;
  }
  catch (...)
  {
  }
}
void guard_note(std::nullptr_t, const char *) noexcept {}

int mfail(ndt2d_matcher * m, int code, const std::string & msg)
{
  if (m != nullptr) m->err = msg;
  return code;
}

int dev_fail(ndt2d_matcher * m, int code, const char * what)
{
  return mfail(m, code, std::string(what) + ": " + ndt2d_last_error(m->dev));
}

// The visited offsets of the search and the per-theta cos/sin (reference
// src/scan_matcher_ndt.cpp:103-107,117,119).  host_beams != nullptr: they are uploaded
// together with the beams in one copy; nullptr: the beams are on the device already
// (LaserScan conversion) and only the tables travel.
int prepare_tables(ndt2d_matcher * m, const double * scan_pose_xyt, size_t use,
                   const double * host_beams, bool beams_cached, size_t * n_th_out,
                   size_t * n_lin_out)
{
  // (the visited offsets depend on the parameters only: computed by initialize())
  const size_t n_th = m->dth.size(), n_lin = m->dlin.size();
  if (n_th_out != nullptr) *n_th_out = n_th;
  if (n_lin_out != nullptr) *n_lin_out = n_lin;
  m->search_ready = false;
  if (use == 0 || n_th == 0 || n_lin == 0 || !m->have_ndt) return NDT2D_OK;  // nothing to upload

  m->cos_th.resize(n_th);
  m->sin_th.resize(n_th);
  for (size_t i = 0; i < n_th; ++i)
  {
    // reference src/scan_matcher_ndt.cpp:106-107
    ndt2d_cos_sin(scan_pose_xyt[2] + m->dth[i], &m->cos_th[i], &m->sin_th[i]);
  }
  int rc;
  if (host_beams != nullptr || beams_cached)
  {
    rc = ndt2d_set_search_beams(m->dev, host_beams, use, scan_pose_xyt[0], scan_pose_xyt[1],
                                m->dth.data(), m->cos_th.data(), m->sin_th.data(), n_th,
                                m->dlin.data(), n_lin);
    if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_set_search_beams");
    m->beams_on_device = true;
  }
  else
  {
    rc = ndt2d_set_search(m->dev, scan_pose_xyt[0], scan_pose_xyt[1], m->dth.data(),
                          m->cos_th.data(), m->sin_th.data(), n_th, m->dlin.data(), n_lin);
    if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_set_search");
  }
  m->search_ready = true;
  return NDT2D_OK;
}

// Subsample `points` (src/scan_matcher_ndt.cpp:165-166,171) and make them the device
// context's beams -- unless they are exactly what it holds already.  *use_out = beams.
// defer_upload: a changed scan is only noted (*pending_out = true; m->beams holds it):
// the caller hands it to ndt2d_score_poses_beams, which uploads it or passes it along as
// kernel arguments.
int stage_beams(ndt2d_matcher * m, const double * points_xy, size_t n_points, size_t * use_out,
                bool * pending_out = nullptr)
{
  if (pending_out != nullptr) *pending_out = false;
  subsample_into(m->scratch_beams, points_xy, n_points, m->laser_max_beams);
  const size_t use = m->scratch_beams.size() / 2;
  *use_out = use;
  if (use == 0) return NDT2D_OK;
  if (m->beams_on_device && m->beams.size() == m->scratch_beams.size() &&
      std::memcmp(m->beams.data(), m->scratch_beams.data(), m->beams.size() * sizeof(double)) == 0)
  {
    return NDT2D_OK;   // same scan as the last call: the beams are there
  }
  m->search_ready = false;  // the device beams are replaced: a prepared search is void
  m->beams_on_device = false;
  if (pending_out != nullptr)
  {
    m->beams.swap(m->scratch_beams);
    ++m->beams_epoch;
    *pending_out = true;
    return NDT2D_OK;
  }
  int rc = ndt2d_set_beams(m->dev, m->scratch_beams.data(), use);
  if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_set_beams");
  m->beams.swap(m->scratch_beams);
  ++m->beams_epoch;
  m->beams_on_device = true;
  return NDT2D_OK;
}

// A search launched ahead that the call now arriving cannot use: wait it out (its record is
// dropped), and do not launch ahead again until the scoreScan / matchScan pair reappears.
void discard_ahead(ndt2d_matcher * m)
{
  m->score_scan_last = false;
  if (!m->ahead) return;
  uint64_t launched = 0, fetched = 0;
  if (ndt2d_match_status(m->dev, &launched, &fetched) == NDT2D_OK && launched == m->ahead_launch_id &&
      fetched == m->ahead_fetch_id)
  {
    ndt2d_match_result res;
    (void)ndt2d_match_fetch(m->dev, &res);
  }
  m->ahead = false;
  m->pair_seen = false;
}

// The host NDT the single-pose path and the near-tie adjudication score against: the one
// addScans built on the host, or -- for a grid built on the DEVICE (maps of 73,728 points and
// more) -- its records fetched back once per addScans.  That fetch is the whole dense grid
// (48 bytes per cell, synchronous) plus a host pool of the same size: worth it for the grids
// the single-pose path exists for, not for a loop-closure map of a million cells, where it
// would cost tens of megabytes over PCIe to save a 30 us launch.  for_single_pose: the caller
// can take the device path instead, so a grid above kHostFetchMaxCells is left on the device;
// the adjudication of a marked near-tie (rare, and it has no device path) fetches any size.
// nullptr also when memory runs out -- nothing may throw through the C-ABI.
constexpr size_t kHostFetchMaxCells = 65536;   // 3 MB of records

const HostNdt * host_ndt(ndt2d_matcher * m, bool for_single_pose)
{
  if (m->ndt) return m->ndt.get();
  if (m->fetched) return m->fetched.get();
  uint32_t sx = 0, sy = 0;
  double cs = 0.0, ox = 0.0, oy = 0.0;
  if (ndt2d_get_grid(m->dev, nullptr, 0, &sx, &sy, &cs, &ox, &oy) != NDT2D_OK) return nullptr;
  const size_t ncell = static_cast<size_t>(sx) * sy;
  if (for_single_pose && ncell > kHostFetchMaxCells) return nullptr;
  try
  {
    std::vector<double> cells(ncell * 6);
    if (ndt2d_get_grid(m->dev, cells.data(), ncell, nullptr, nullptr, nullptr, nullptr, nullptr) != NDT2D_OK)
    {
      return nullptr;
    }
    std::unique_ptr<HostNdt> g(new HostNdt(cs, 0.0, 0.0, ox, oy));
    g->reset_cells(cs, sx, sy, ox, oy);
    g->load6(cells.data());
    m->fetched = std::move(g);
  }
  catch (const std::bad_alloc &)
  {
    m->fetched.reset();
    return nullptr;
  }
  return m->fetched.get();
}

// ScanMatcherNDT::scorePoints on the host (reference src/scan_matcher_ndt.cpp:156-178): the
// pose as toEigen makes it (conversions.hpp:64-68: [[c, -s], [s, c]] and the translation),
// the subsampling of :165-171, `score += -likelihood(p)` in beam order, score / N.
double host_score_points(const HostNdt & ndt, const double * points_xy, size_t n_points, size_t laser_max_beams,
                         const double * pose_xyt)
{
  double c, s;
  ndt2d_cos_sin(pose_xyt[2], &c, &s);
  const size_t use = std::min(laser_max_beams, n_points);
  const double scan_step = static_cast<double>(n_points) / use;
  double score = 0.0;
  for (size_t i = 0; i < use; ++i)
  {
    const size_t idx = static_cast<size_t>(i * scan_step);
    const double x = points_xy[2 * idx], y = points_xy[2 * idx + 1];
    const double px = pose_xyt[0] + (c * x + (-s) * y);
    const double py = pose_xyt[1] + (s * x + c * y);
    score += -ndt.likelihood(px, py);
  }
  return score / use;
}

// One candidate of matchScan's lattice as the reference scores it (src/scan_matcher_ndt.cpp:
// 106-127): points_outer from the subsampled beams `beams` (m->beams) and cos/sin of
// scan_pose.theta + dth, points_inner = outer + (dx, dy), score = -(likelihoods summed in order).
double host_score_candidate(const HostNdt & ndt, const double * beams_xy, size_t use, const double * scan_pose_xyt,
                            double costh, double sinth, double dx, double dy)
{
  double sum = 0.0;
  for (size_t i = 0; i < use; ++i)
  {
    const double bx = beams_xy[2 * i], by = beams_xy[2 * i + 1];
    const double ox = bx * costh - by * sinth + scan_pose_xyt[0];
    const double oy = bx * sinth + by * costh + scan_pose_xyt[1];
    sum += ndt.likelihood(ox + dx, oy + dy);
  }
  return -sum;
}

// A search's record came back with its winner marked (index + 0.5: another candidate within
// the near-tie tolerance, ndt2d_device_fn.h merge_best): list the candidates that close to the best,
// rescore them as the reference would and apply its rule -- strict `<` in visiting order
// (src/scan_matcher_ndt.cpp:128-134).  The first device holds the prepared search.  record[1]
// leaves here as a plain index.
int settle_near_tie(ndt2d_matcher * m, const double * scan_pose_xyt, size_t n_th, size_t n_lin, size_t use, double * record)
{
  if (!(record[1] >= 0.0) || record[1] == std::floor(record[1])) return NDT2D_OK;   // not marked
  record[1] = std::floor(record[1]);
  ++m->adj_marked;
  if (!m->adjudicate || m->beams.size() != 2 * use || m->cos_th.size() != n_th) return NDT2D_OK;
  const HostNdt * ndt = host_ndt(m, false);
  if (ndt == nullptr) return NDT2D_OK;
  constexpr size_t kCap = 256;
  uint64_t idx[kCap + 1];
  size_t n = 0;
  const int rc = ndt2d_match_near_best(m->dev, 0, n_th, NDT2D_NEAR_TIE_REL, idx, kCap, &n, nullptr);
  if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_match_near_best");
  if (n > kCap) ++m->adj_truncated;
  size_t listed = std::min(n, kCap);
  // The device's own winner is always among the rescored: a plateau of more than kCap
  // candidates is cut to the first kCap in visiting order, and the winner may lie beyond the
  // cut -- it must then not be replaced by a candidate that neither the device nor the
  // reference would pick (only by one whose reference-order score is strictly lower).
  {
    const uint64_t winner = static_cast<uint64_t>(record[1]);
    uint64_t * const end = idx + listed;
    uint64_t * const at = std::lower_bound(idx, end, winner);
    if (at == end || *at != winner)
    {
      const size_t pos = static_cast<size_t>(at - idx);
      for (size_t k = listed; k > pos; --k) idx[k] = idx[k - 1];
      idx[pos] = winner;
      ++listed;
    }
  }
  const uint64_t per_th = static_cast<uint64_t>(n_lin) * n_lin;
  double best_s = 0.0;   // `double best_score = 0;` (:83)
  uint64_t best_i = NDT2D_NO_INDEX;
  for (size_t k = 0; k < listed; ++k)   // ascending flat index = the reference's visiting order
  {
    const uint64_t ith = idx[k] / per_th, rem = idx[k] % per_th;
    if (ith >= n_th) continue;
    const double score = host_score_candidate(*ndt, m->beams.data(), use, scan_pose_xyt, m->cos_th[ith], m->sin_th[ith],
                                              m->dlin[rem / n_lin], m->dlin[rem % n_lin]);
    if (score < best_s)
    {
      best_s = score;
      best_i = idx[k];
    }
  }
  if (best_i != NDT2D_NO_INDEX)
  {
    if (static_cast<double>(best_i) != record[1]) ++m->adj_changed;
    record[0] = best_s;
    record[1] = static_cast<double>(best_i);
  }
  return NDT2D_OK;
}

// ---------------------------------------------------------------------------
// Multi-device matcher (ndt2d_matcher_create_multi): the sharded calls
// ---------------------------------------------------------------------------

constexpr size_t kRec = NDT2D_MATCH_RECORD_DOUBLES;
constexpr size_t kStats = NDT2D_POSE_STATS_DOUBLES;

// Layout of m->pinned (doubles; n = number of devices):
//   [r n 12 .. (r + 1) n 12)  initial image of device r's record table: zero, its own row {0, -1, 0 ...}
//   zeros [n 12]              initial image of a moment table
//   rows  [n 12]              the table read back from the first device / the rows the host combines
//   sum   [8]                 the summed moments on their way to the devices (host exchange)
size_t pinned_init_off(size_t n, size_t r) { return r * n * kRec; }
size_t pinned_zero_off(size_t n) { return n * n * kRec; }
size_t pinned_rows_off(size_t n) { return n * n * kRec + n * kRec; }
size_t pinned_sum_off(size_t n) { return n * n * kRec + 2 * n * kRec; }
size_t pinned_doubles(size_t n) { return pinned_sum_off(n) + kStats; }

int dev_fail_at(ndt2d_matcher * m, size_t r, int code, const char * what)
{
  return mfail(m, code, std::string(what) + " (device " + std::to_string(m->device_ids[r]) + ", rank " +
                          std::to_string(r) + "): " + ndt2d_last_error(m->devs[r]));
}

// The exchange buffers every sharded call needs: made on the first one.
int ensure_multi(ndt2d_matcher * m)
{
  const size_t n = m->devs.size();
  if (m->pinned == nullptr)
  {
    void * p = nullptr;
    const int rc = ndt2d_host_alloc(m->dev, pinned_doubles(n) * sizeof(double), &p);
    if (rc != NDT2D_OK) return dev_fail_at(m, 0, rc, "ndt2d_host_alloc");
    m->pinned = static_cast<double *>(p);
    std::memset(m->pinned, 0, pinned_doubles(n) * sizeof(double));
    for (size_t r = 0; r < n; ++r) m->pinned[pinned_init_off(n, r) + r * kRec + 1] = -1.0;   // "no candidate"
  }
  for (size_t r = 0; r < n; ++r)
  {
    MatcherShard & sh = m->shards[r];
    if (sh.d_table == nullptr)
    {
      void * d = nullptr;
      int rc = ndt2d_device_alloc(m->devs[r], n * kRec * sizeof(double), &d);
      if (rc != NDT2D_OK) return dev_fail_at(m, r, rc, "ndt2d_device_alloc");
      sh.d_table = static_cast<double *>(d);
      rc = ndt2d_device_alloc(m->devs[r], 2 * kStats * sizeof(double), &d);
      if (rc != NDT2D_OK) return dev_fail_at(m, r, rc, "ndt2d_device_alloc");
      sh.d_sum = static_cast<double *>(d);
    }
  }
  return NDT2D_OK;
}

// Which exchange this call takes (ndt2d_matcher_set_exchange).
int pick_exchange(ndt2d_matcher * m, bool * rccl)
{
  *rccl = false;
  if (m->exchange_mode == 1) return NDT2D_OK;
  if (!m->exchange_tried)
  {
    m->exchange_tried = true;
    std::string why;
    const int rc = ndt2d::exchange_create(&m->exchange, m->device_ids.data(), static_cast<int>(m->device_ids.size()), &why);
    if (rc != NDT2D_OK)
    {
      m->exchange = nullptr;
      m->exchange_note = why;
    }
  }
  if (m->exchange == nullptr)
  {
    if (m->exchange_mode == 2) return mfail(m, NDT2D_ERR_HIP, "exchange \"rccl\" is not available: " + m->exchange_note);
    return NDT2D_OK;   // "auto": the host exchange
  }
  *rccl = true;
  return NDT2D_OK;
}

bool multi_search_wanted(const ndt2d_matcher * m, size_t n_th, size_t n_lin, size_t use)
{
  // (a one-device matcher told to use "rccl" takes the dealt path with one rank: the collective
  // code can then be exercised on a single-GPU box)
  return (m->devs.size() > 1 || m->exchange_mode == 2) && n_th >= 2 &&
         static_cast<double>(n_th) * static_cast<double>(n_lin) * static_cast<double>(n_lin) * static_cast<double>(use) >=
           m->multi_min_units;
}

bool multi_poses_wanted(const ndt2d_matcher * m, size_t n_poses, size_t use)
{
  return (m->devs.size() > 1 || m->exchange_mode == 2) && n_poses >= m->devs.size() &&
         static_cast<double>(n_poses) * static_cast<double>(use) >= m->multi_min_pose_units;
}

void note_variant(ndt2d_matcher * m, bool multi, bool rccl)
{
  m->last_multi = multi;
  m->variant.clear();
  if (multi)
  {
    m->variant = "multi[" + std::to_string(m->devs.size()) + "]/" + (rccl ? "rccl" : "host") + "/";
  }
  m->variant += ndt2d_last_variant(m->dev);
}

// The reference's first-wins rule over the devices' records (src/scan_matcher_ndt.cpp:128, strict
// `<` in visiting order): the lower score, and between equal scores the lower flat index -- the
// candidate the reference's loops visit first; accumulators summed in device order.
// rows[n][12], used[r] = device r searched; out[12].
void combine_records(const double * rows, const std::vector<size_t> & count, double * out)
{
  out[0] = 0.0;
  out[1] = -1.0;
  for (size_t k = 2; k < kRec; ++k) out[k] = 0.0;
  bool first = true;
  for (size_t r = 0; r < count.size(); ++r)
  {
    if (count[r] == 0) continue;
    const double * rec = rows + r * kRec;
    if (rec[1] >= 0.0 && rec[0] < 0.0)
    {
      // (an index ending in .5 is a winner marked "another candidate within the near-tie tolerance": the
      // mark stays with the winner, and two devices' winners that close mark it as well)
      const bool near = out[1] >= 0.0 && std::fabs(rec[0] - out[0]) <=
                                           std::max(std::fabs(rec[0]), std::fabs(out[0])) * NDT2D_NEAR_TIE_REL;
      if (out[1] < 0.0 || rec[0] < out[0] || (rec[0] == out[0] && std::floor(rec[1]) < std::floor(out[1])))
      {
        out[0] = rec[0];
        out[1] = rec[1];
      }
      if (near) out[1] = std::floor(out[1]) + 0.5;
    }
    // (the first device's sums are taken as they are: one device gives the single-device bits)
    for (size_t k = 2; k < kRec; ++k) out[k] = first ? rec[k] : out[k] + rec[k];
    first = false;
  }
}

// What one device's thread reports of its part of a dealt call (the message is made by the
// calling thread afterwards: ndt2d_matcher::err is not the threads' to write).
struct RankStatus
{
  int rc = NDT2D_OK;
  const char * what = "";
};

double elapsed_us(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

// A dealt call is given up: nothing of it may stay in flight once the caller has the error -- copies
// out of / into the caller's buffers, searches launched but not fetched.  Every device is waited out.
void drain_devices(ndt2d_matcher * m)
{
  for (ndt2d_handle h : m->devs)
  {
    uint64_t launched = 0, fetched = 0;
    if (ndt2d_match_status(h, &launched, &fetched) == NDT2D_OK && launched > fetched)
    {
      ndt2d_match_result res;
      (void)ndt2d_match_fetch(h, &res);
    }
    (void)ndt2d_synchronize(h);
  }
}

int first_failure(ndt2d_matcher * m, const std::vector<RankStatus> & st)
{
  for (size_t r = 0; r < st.size(); ++r)
  {
    if (st[r].rc != NDT2D_OK)
    {
      const int rc = dev_fail_at(m, r, st[r].rc, st[r].what);   // (the message, before the devices are touched again)
      drain_devices(m);
      return rc;
    }
  }
  return NDT2D_OK;
}

// matchScan's search dealt to all devices.  The first device has been prepared by the caller
// (beams + tables); m->beams holds the subsampled beams unless `beams_everywhere` (every device
// converted the LaserScan itself).  all_scores (host, optional): the whole lattice's scores.
// Every device's tables, beams and launch go out on its own thread (ndt2d_workers.h).
int multi_match(ndt2d_matcher * m, const double * scan_pose_xyt, size_t n_th, size_t n_lin, size_t use,
                bool beams_everywhere, double * all_scores, double * record_out)
{
  const size_t n = m->devs.size();
  const auto t_start = std::chrono::steady_clock::now();
  int rc = ensure_multi(m);
  if (rc != NDT2D_OK) return rc;
  bool rccl = false;
  if ((rc = pick_exchange(m, &rccl)) != NDT2D_OK) return rc;
  const size_t per_th = n_lin * n_lin;
  std::vector<size_t> count(n, 0);
  for (size_t r = 0; r < n; ++r) count[r] = r < n_th ? (n_th - r + n - 1) / n : 0;
  if (all_scores != nullptr)
  {
    for (size_t r = 0; r < n; ++r)
    {
      MatcherShard & sh = m->shards[r];
      const size_t want = count[r] * per_th;
      if (want > sh.scores_cap)
      {
        if (sh.d_scores != nullptr) ndt2d_device_free(m->devs[r], sh.d_scores);
        sh.d_scores = nullptr;
        sh.scores_cap = 0;
        void * d = nullptr;
        rc = ndt2d_device_alloc(m->devs[r], want * sizeof(double), &d);
        if (rc != NDT2D_OK) return dev_fail_at(m, r, rc, "ndt2d_device_alloc");
        sh.d_scores = static_cast<double *>(d);
        sh.scores_cap = want;
      }
    }
  }
  // every device's search goes out before any result is waited for, all of them side by side
  std::vector<RankStatus> st(n);
  m->fanout_us.assign(n, 0.0);
  auto deal = [&](size_t r) {
    RankStatus & s = st[r];
    MatcherShard & sh = m->shards[r];
    if (r > 0)
    {
      s.what = "ndt2d_set_search_beams";
      if (beams_everywhere)
      {
        s.rc = ndt2d_set_search(m->devs[r], scan_pose_xyt[0], scan_pose_xyt[1], m->dth.data(), m->cos_th.data(),
                                m->sin_th.data(), n_th, m->dlin.data(), n_lin);
      }
      else
      {
        s.rc = ndt2d_set_search_beams(m->devs[r], m->beams.data(), use, scan_pose_xyt[0], scan_pose_xyt[1], m->dth.data(),
                                      m->cos_th.data(), m->sin_th.data(), n_th, m->dlin.data(), n_lin);
      }
      sh.beams_epoch = ~0ull;   // (the device's beams now live in the search's upload)
      if (s.rc != NDT2D_OK) return;
    }
    if (rccl)
    {
      s.what = "ndt2d_copy_to_device_async";
      s.rc = ndt2d_copy_to_device_async(m->devs[r], sh.d_table, m->pinned + pinned_init_off(n, r), n * kRec * sizeof(double));
      if (s.rc != NDT2D_OK) return;
    }
    if (count[r] > 0)
    {
      s.what = "ndt2d_match_launch_strided";
      s.rc = ndt2d_match_launch_strided(m->devs[r], r, n, count[r], all_scores != nullptr ? sh.d_scores : nullptr,
                                        rccl ? sh.d_table + r * kRec : nullptr);
    }
    m->fanout_us[r] = elapsed_us(t_start);
  };
  m->workers->run(deal);
  if ((rc = first_failure(m, st)) != NDT2D_OK) return rc;
  double * rows = m->pinned + pinned_rows_off(n);
  if (rccl)
  {
    // the ONE collective of the search: all-reduce(sum) of the [n, 12] table, every device its own row
    std::vector<double *> tables(n);
    std::vector<void *> streams(n);
    for (size_t r = 0; r < n; ++r)
    {
      tables[r] = m->shards[r].d_table;
      streams[r] = ndt2d_get_stream(m->devs[r]);
    }
    std::string why;
    rc = ndt2d::exchange_all_reduce(m->exchange, tables.data(), n * kRec, streams.data(), &why);
    if (rc != NDT2D_OK)
    {
      drain_devices(m);
      return mfail(m, rc, why);
    }
    rc = ndt2d_copy_to_host_async(m->dev, rows, m->shards[0].d_table, n * kRec * sizeof(double));
    if (rc == NDT2D_OK) rc = ndt2d_synchronize(m->dev);
    if (rc != NDT2D_OK)
    {
      const int frc = dev_fail_at(m, 0, rc, "ndt2d_copy_to_host_async");
      drain_devices(m);
      return frc;
    }
  }
  for (size_t r = 0; r < n; ++r)
  {
    if (count[r] == 0) continue;
    // host exchange: the record through the context's host-coherent result block.  (After an
    // all-reduce the flags are up already -- every stream's collective follows its search --
    // and the fetch only settles the context's state.)
    ndt2d_match_result res;
    rc = ndt2d_match_fetch(m->devs[r], &res);
    if (rc != NDT2D_OK)
    {
      const int frc = dev_fail_at(m, r, rc, "ndt2d_match_fetch");
      drain_devices(m);
      return frc;
    }
    if (!rccl)
    {
      double * rec = rows + r * kRec;
      rec[0] = res.best_score;
      rec[1] = res.best_index == NDT2D_NO_INDEX ? -1.0 : static_cast<double>(res.best_index) + (res.near_tie ? 0.5 : 0.0);
      for (int i = 0; i < 10; ++i) rec[2 + i] = res.acc[i];
    }
  }
  if (all_scores != nullptr)
  {
    // device r holds the scores of the steps r, r + n, ... in that order
    std::vector<double> tmp;
    for (size_t r = 0; r < n; ++r)
    {
      if (count[r] == 0) continue;
      tmp.resize(count[r] * per_th);
      rc = ndt2d_copy_to_host(m->devs[r], tmp.data(), m->shards[r].d_scores, tmp.size() * sizeof(double));
      if (rc != NDT2D_OK)
      {
        const int frc = dev_fail_at(m, r, rc, "ndt2d_copy_to_host");
        drain_devices(m);
        return frc;
      }
      for (size_t k = 0; k < count[r]; ++k)
      {
        std::memcpy(all_scores + (r + k * n) * per_th, tmp.data() + k * per_th, per_th * sizeof(double));
      }
    }
  }
  combine_records(rows, count, record_out);
  note_variant(m, true, rccl);
  return NDT2D_OK;
}

// Room for `n_poses` particles and their weights on device r.
int ensure_shard_poses(ndt2d_matcher * m, size_t r, size_t n_poses)
{
  MatcherShard & sh = m->shards[r];
  if (n_poses <= sh.poses_cap) return NDT2D_OK;
  if (sh.d_poses != nullptr) ndt2d_device_free(m->devs[r], sh.d_poses);
  if (sh.d_weights != nullptr) ndt2d_device_free(m->devs[r], sh.d_weights);
  sh.d_poses = sh.d_weights = nullptr;
  sh.poses_cap = 0;
  const size_t cap = n_poses + n_poses / 8;
  void * d = nullptr;
  int rc = ndt2d_device_alloc(m->devs[r], 3 * cap * sizeof(double), &d);
  if (rc != NDT2D_OK) return dev_fail_at(m, r, rc, "ndt2d_device_alloc");
  sh.d_poses = static_cast<double *>(d);
  rc = ndt2d_device_alloc(m->devs[r], cap * sizeof(double), &d);
  if (rc != NDT2D_OK) return dev_fail_at(m, r, rc, "ndt2d_device_alloc");
  sh.d_weights = static_cast<double *>(d);
  sh.poses_cap = cap;
  return NDT2D_OK;
}

// Contiguous share [begin, end) of n items for rank r of `world` (sizes differ by at most one).
void shard_range(size_t n, size_t r, size_t world, size_t * begin, size_t * end)
{
  const size_t base = n / world, rem = n % world;
  *begin = r * base + std::min(r, rem);
  *end = *begin + base + (r < rem ? 1 : 0);
}

// scorePoses / ParticleFilter::measure over all devices: contiguous particle ranges.  The first
// device holds the beams (the caller staged them), m->beams is their host copy.  stats_out ==
// nullptr: scores only.  Otherwise the whole of measure: scores_out receives the normalised
// weights and stats_out NDT2D_PF_RESULT_DOUBLES values as ndt2d_pf_finalize_launch defines them
// ([7] summed over the devices in device order).
//
// Every device's share runs on its own thread from the upload to the weights' way back.  Host
// exchange: the device's eight moment sums arrive in its host-coherent block behind a flag
// (ndt2d_pose_sums_fetch), the threads meet (DeviceWorkers::barrier), each adds the rows in device
// order -- the "total particle weight" of src/particle_filter.cpp:166-174 -- and launches
// updateStatistics with the totals as kernel arguments: no copy and no stream synchronisation
// between the two halves.  RCCL exchange: the shares are dealt the same way, the two
// all-reduces are issued by the calling thread.
int multi_score_poses(ndt2d_matcher * m, const double * poses_xyt, size_t n_poses, size_t use, double * scores_out,
                      double * stats_out)
{
  const size_t n = m->devs.size();
  const auto t_start = std::chrono::steady_clock::now();
  int rc = ensure_multi(m);
  if (rc != NDT2D_OK) return rc;
  bool rccl = false;
  if (stats_out != nullptr && (rc = pick_exchange(m, &rccl)) != NDT2D_OK) return rc;
  std::vector<size_t> begin(n), end(n);
  for (size_t r = 0; r < n; ++r)
  {
    shard_range(n_poses, r, n, &begin[r], &end[r]);
    if ((rc = ensure_shard_poses(m, r, end[r] - begin[r])) != NDT2D_OK) return rc;
  }
  const double * zeros = m->pinned + pinned_zero_off(n);
  double * rows = m->pinned + pinned_rows_off(n);
  std::vector<RankStatus> st(n);
  m->fanout_us.assign(n, 0.0);
  std::atomic<bool> give_up{false};
  // (host exchange: the devices' rows of sums and of results, written by their threads)
  std::vector<double> sums(n * kStats, 0.0), results(n * kStats, 0.0);
  const bool host_measure = stats_out != nullptr && !rccl;

  auto share = [&](size_t r) {
    RankStatus & s = st[r];
    MatcherShard & sh = m->shards[r];
    const size_t nr = end[r] - begin[r];
    auto fail = [&](const char * what, bool sync) {
      s.what = what;
      give_up.store(true, std::memory_order_release);
      if (sync) (void)ndt2d_synchronize(m->devs[r]);   // nothing of this share stays in flight
    };
    if (r > 0 && sh.beams_epoch != m->beams_epoch)
    {
      if ((s.rc = ndt2d_set_beams(m->devs[r], m->beams.data(), use)) != NDT2D_OK) return fail("ndt2d_set_beams", true);
      sh.beams_epoch = m->beams_epoch;
    }
    s.rc = ndt2d_copy_to_device_async(m->devs[r], sh.d_poses, poses_xyt + 3 * begin[r], 3 * nr * sizeof(double));
    if (s.rc == NDT2D_OK && rccl)
    {
      s.rc = ndt2d_copy_to_device_async(m->devs[r], sh.d_table, zeros, n * kStats * sizeof(double));
    }
    if (s.rc != NDT2D_OK) return fail("ndt2d_copy_to_device_async", true);
    if (!host_measure)
    {
      // scores only, or the first half of the RCCL form (the moment sums into the device's row)
      double * d_stats = stats_out == nullptr ? nullptr : sh.d_table + r * kStats;
      s.rc = ndt2d_score_poses_launch(m->devs[r], sh.d_poses, nr, sh.d_weights, d_stats);
      m->fanout_us[r] = elapsed_us(t_start);
      if (s.rc != NDT2D_OK) return fail("ndt2d_score_poses_launch", true);
      if (stats_out == nullptr)
      {
        s.rc = ndt2d_copy_to_host_async(m->devs[r], scores_out + begin[r], sh.d_weights, nr * sizeof(double));
        if (s.rc == NDT2D_OK) s.rc = ndt2d_synchronize(m->devs[r]);
        if (s.rc != NDT2D_OK) return fail("ndt2d_copy_to_host_async", true);
      }
      return;
    }
    s.rc = ndt2d_pose_sums_launch(m->devs[r], sh.d_poses, nr, sh.d_weights);
    m->fanout_us[r] = elapsed_us(t_start);
    if (s.rc != NDT2D_OK) return fail("ndt2d_pose_sums_launch", true);
    if ((s.rc = ndt2d_pose_sums_fetch(m->devs[r], sums.data() + r * kStats)) != NDT2D_OK) return fail("ndt2d_pose_sums_fetch", true);
    if (!m->workers->barrier(give_up))
    {
      // another device's share failed (its status says how) or never came: this one is abandoned
      if (!give_up.load()) { s.rc = NDT2D_ERR_HIP; fail("the devices' moment sums did not meet", false); }
      (void)ndt2d_synchronize(m->devs[r]);
      return;
    }
    // the rows summed in device order: the same bits on every thread
    double totals[kStats];
    for (size_t k = 0; k < kStats; ++k)
    {
      double acc = sums[k];
      for (size_t q = 1; q < n; ++q) acc += sums[q * kStats + k];
      totals[k] = acc;
    }
    // updateStatistics with the total sums: normalised weights, the mean and covariance (the same
    // on all devices), and the device's part of the theta variance (:213-217)
    s.rc = ndt2d_pf_finalize_totals_launch(m->devs[r], sh.d_poses, nr, sh.d_weights, totals);
    if (s.rc != NDT2D_OK) return fail("ndt2d_pf_finalize_totals_launch", true);
    s.rc = ndt2d_copy_to_host_async(m->devs[r], scores_out + begin[r], sh.d_weights, nr * sizeof(double));
    if (s.rc == NDT2D_OK) s.rc = ndt2d_synchronize(m->devs[r]);
    if (s.rc != NDT2D_OK) return fail("ndt2d_copy_to_host_async", true);
    if ((s.rc = ndt2d_pf_result_read(m->devs[r], results.data() + r * kStats)) != NDT2D_OK) return fail("ndt2d_pf_result_read", false);
  };
  m->workers->run(share);
  if ((rc = first_failure(m, st)) != NDT2D_OK) return rc;
  if (host_measure)
  {
    for (size_t k = 0; k < NDT2D_PF_RESULT_DOUBLES; ++k) stats_out[k] = results[k];
    for (size_t r = 1; r < n; ++r) stats_out[7] += results[r * kStats + 7];
    note_variant(m, true, false);
    return NDT2D_OK;
  }
  if (stats_out == nullptr)
  {
    note_variant(m, true, false);
    return NDT2D_OK;
  }

  // RCCL exchange.  On any failure from here on every device is waited out before the error returns.
  auto give_in = [&](int code, const std::string & msg) {
    const int frc = mfail(m, code, msg);
    drain_devices(m);
    return frc;
  };
  std::vector<double *> tables(n);
  std::vector<void *> streams(n);
  for (size_t r = 0; r < n; ++r)
  {
    tables[r] = m->shards[r].d_table;
    streams[r] = ndt2d_get_stream(m->devs[r]);
  }
  std::string why;
  // the "total particle weight" all-reduce (src/particle_filter.cpp:166-174) with the other
  // seven moment sums: [n, 8], every device its own row -- the ONE collective of the call
  // (until round 6 the devices' theta-variance parts went through a second one)
  rc = ndt2d::exchange_all_reduce(m->exchange, tables.data(), n * kStats, streams.data(), &why);
  if (rc != NDT2D_OK) return give_in(rc, why);
  // Behind it every device goes on by itself, on its own thread: the rows summed in device order
  // (the same bits everywhere), updateStatistics with them -- normalised weights, mean and
  // covariance, and the device's OWN part of the theta variance (:213-217) -- then the weights and
  // the eight results travel home together; the parts are added below, in device order.
  for (RankStatus & s : st) s = RankStatus();
  std::vector<std::string> whys(n);
  auto finish = [&](size_t r) {
    RankStatus & s = st[r];
    MatcherShard & sh = m->shards[r];
    const size_t nr = end[r] - begin[r];
    s.what = "sum_rows_launch";
    s.rc = ndt2d::sum_rows_launch(m->device_ids[r], tables[r], static_cast<int>(n), static_cast<int>(kStats), sh.d_sum,
                                  streams[r], &whys[r]);
    if (s.rc == NDT2D_OK)
    {
      s.what = "ndt2d_pf_finalize_launch";
      s.rc = ndt2d_pf_finalize_launch(m->devs[r], sh.d_poses, nr, sh.d_weights, sh.d_sum, sh.d_sum + kStats);
    }
    if (s.rc == NDT2D_OK)
    {
      s.what = "ndt2d_copy_to_host_async";
      s.rc = ndt2d_copy_to_host_async(m->devs[r], scores_out + begin[r], sh.d_weights, nr * sizeof(double));
    }
    if (s.rc == NDT2D_OK)
    {
      s.rc = ndt2d_copy_to_host_async(m->devs[r], rows + r * kStats, sh.d_sum + kStats, kStats * sizeof(double));
    }
    const int src = ndt2d_synchronize(m->devs[r]);   // (whatever happened: nothing of this device stays in flight)
    if (s.rc == NDT2D_OK) s.rc = src;
  };
  m->workers->run(finish);
  if ((rc = first_failure(m, st)) != NDT2D_OK) return rc;
  for (size_t k = 0; k < NDT2D_PF_RESULT_DOUBLES; ++k) stats_out[k] = rows[k];
  for (size_t r = 1; r < n; ++r) stats_out[7] += rows[r * kStats + 7];
  note_variant(m, true, true);
  return NDT2D_OK;
}

void destroy_matcher(ndt2d_matcher * m)
{
  m->workers.reset();   // (the threads end before the contexts they drive)
  // (a matcher whose creation failed half-way has contexts but no shard records yet)
  for (size_t r = 0; r < m->devs.size() && r < m->shards.size(); ++r)
  {
    MatcherShard & sh = m->shards[r];
    (void)ndt2d_synchronize(m->devs[r]);
    if (sh.d_table != nullptr) ndt2d_device_free(m->devs[r], sh.d_table);
    if (sh.d_sum != nullptr) ndt2d_device_free(m->devs[r], sh.d_sum);
    if (sh.d_scores != nullptr) ndt2d_device_free(m->devs[r], sh.d_scores);
    if (sh.d_poses != nullptr) ndt2d_device_free(m->devs[r], sh.d_poses);
    if (sh.d_weights != nullptr) ndt2d_device_free(m->devs[r], sh.d_weights);
  }
  if (m->exchange != nullptr) ndt2d::exchange_destroy(m->exchange);
  if (m->pinned != nullptr && !m->devs.empty()) ndt2d_host_free(m->dev, m->pinned);
  for (ndt2d_handle h : m->devs) ndt2d_destroy(h);
  delete m;
}

}  // namespace

extern "C" {

int ndt2d_matcher_create_multi(ndt2d_matcher ** out, const int * device_ids, int n_dev)
{
  if (out == nullptr) return NDT2D_ERR_INVALID;
  *out = nullptr;
  if (device_ids == nullptr || n_dev <= 0 || n_dev > 64) return NDT2D_ERR_INVALID;
  ndt2d_matcher * m = new (std::nothrow) ndt2d_matcher();
  if (m == nullptr) return NDT2D_ERR_ALLOC;
  try
  {
    m->devs.reserve(static_cast<size_t>(n_dev));
    m->device_ids.reserve(static_cast<size_t>(n_dev));
    for (int r = 0; r < n_dev; ++r)
    {
      ndt2d_handle dev = nullptr;
      const int rc = ndt2d_create(&dev, device_ids[r]);
      if (rc != NDT2D_OK)
      {
        destroy_matcher(m);
        return rc;
      }
      m->devs.push_back(dev);
      m->device_ids.push_back(device_ids[r]);
    }
    m->shards.resize(m->devs.size());
    m->dev = m->devs[0];
    m->workers.reset(new ndt2d::DeviceWorkers(m->devs.size()));
    m->dth = search_offsets(m->angular_size, m->angular_res);
    m->dlin = search_offsets(m->linear_size, m->linear_res);
  }
  catch (const std::bad_alloc &)
  {
    destroy_matcher(m);
    return NDT2D_ERR_ALLOC;
  }
  catch (...)
  {
    destroy_matcher(m);
    return NDT2D_ERR_INTERNAL;   // (no thread could be started)
  }
  *out = m;
  return NDT2D_OK;
}

int ndt2d_matcher_create(ndt2d_matcher ** out, int device_id)
{
  NDT2D_C_TRY
  return ndt2d_matcher_create_multi(out, &device_id, 1);
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_matcher_destroy(ndt2d_matcher * m)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  discard_ahead(m);
  destroy_matcher(m);
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_device_count(ndt2d_matcher * m) { return m != nullptr ? static_cast<int>(m->devs.size()) : 0; }

ndt2d_handle ndt2d_matcher_device_at(ndt2d_matcher * m, int rank)
{
  return (m != nullptr && rank >= 0 && static_cast<size_t>(rank) < m->devs.size()) ? m->devs[static_cast<size_t>(rank)] : nullptr;
}

int ndt2d_matcher_set_exchange(ndt2d_matcher * m, const char * mode)
{
  NDT2D_C_TRY
  if (m == nullptr || mode == nullptr) return NDT2D_ERR_INVALID;
  if (std::strcmp(mode, "auto") == 0) m->exchange_mode = 0;
  else if (std::strcmp(mode, "host") == 0) m->exchange_mode = 1;
  else if (std::strcmp(mode, "rccl") == 0) m->exchange_mode = 2;
  else return mfail(m, NDT2D_ERR_INVALID, "set_exchange: unknown mode (auto, host, rccl)");
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_multi_min_units(ndt2d_matcher * m, double units)
{
  NDT2D_C_TRY
  return ndt2d_matcher_set_multi_thresholds(m, units, units);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_multi_thresholds(ndt2d_matcher * m, double min_search_units, double min_pose_units)
{
  NDT2D_C_TRY
  if (m == nullptr || !(min_search_units >= 0.0) || !(min_pose_units >= 0.0)) return NDT2D_ERR_INVALID;
  m->multi_min_units = min_search_units;
  m->multi_min_pose_units = min_pose_units;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_get_multi_thresholds(ndt2d_matcher * m, double * min_search_units, double * min_pose_units)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (min_search_units != nullptr) *min_search_units = m->multi_min_units;
  if (min_pose_units != nullptr) *min_pose_units = m->multi_min_pose_units;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_last_fanout_us(ndt2d_matcher * m, double * out_us, size_t capacity, size_t * n_out)
{
  NDT2D_C_TRY
  if (m == nullptr || (capacity > 0 && out_us == nullptr)) return NDT2D_ERR_INVALID;
  if (n_out != nullptr) *n_out = m->fanout_us.size();
  for (size_t r = 0; r < m->fanout_us.size() && r < capacity; ++r) out_us[r] = m->fanout_us[r];
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

const char * ndt2d_matcher_last_variant(ndt2d_matcher * m)
{
  if (m == nullptr) return "";
  try
  {
    if (!m->last_multi) note_variant(m, false, false);   // whatever the first device ran last
  }
  catch (...)
  {
    return "";
  }
  return m->variant.c_str();
}

int ndt2d_matcher_set_timing(ndt2d_matcher * m, int enabled)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  for (ndt2d_handle h : m->devs)
  {
    const int rc = ndt2d_set_timing(h, enabled);
    if (rc != NDT2D_OK) return rc;
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

const char * ndt2d_matcher_last_error(ndt2d_matcher * m)
{
  return m != nullptr ? m->err.c_str() : "null matcher";
}

ndt2d_handle ndt2d_matcher_device(ndt2d_matcher * m) { return m != nullptr ? m->dev : nullptr; }

int ndt2d_matcher_initialize(ndt2d_matcher * m, double ndt_resolution,
                             double search_angular_resolution, double search_angular_size,
                             double search_linear_resolution, double search_linear_size,
                             size_t laser_max_beams, double range_max)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (!(ndt_resolution > 0.0) || !std::isfinite(ndt_resolution))
  {
    return mfail(m, NDT2D_ERR_INVALID, "ndt_resolution must be a finite number > 0");
  }
  if (!std::isfinite(range_max)) return mfail(m, NDT2D_ERR_INVALID, "range_max must be finite");
  // The lattice is visited by `for (v = -size; v < size; v += res)` (src/scan_matcher_ndt.cpp:103,
  // 117,119): a step the range never gets past would not end there, and one that needs more steps
  // than a search can take (ndt2d_set_search: 2^24 angular, 46,340 linear) would only fill memory
  // here -- refused before a single offset is stored.
  if (!offsets_fit(search_angular_size, search_angular_resolution, 1u << 24) ||
      !offsets_fit(search_linear_size, search_linear_resolution, 46340))
  {
    return mfail(m, NDT2D_ERR_INVALID, "search lattice: size / resolution must be finite, the resolution > 0, at most "
                                       "2^24 angular and 46,340 linear steps");
  }
  discard_ahead(m);
  m->pair_seen = false;
  m->resolution = ndt_resolution;
  m->angular_res = search_angular_resolution;
  m->angular_size = search_angular_size;
  m->linear_res = search_linear_resolution;
  m->linear_size = search_linear_size;
  m->laser_max_beams = laser_max_beams;
  m->range_max = range_max;
  m->dth = search_offsets(m->angular_size, m->angular_res);
  m->dlin = search_offsets(m->linear_size, m->linear_res);
  m->search_ready = false;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_add_scans(ndt2d_matcher * m, const double * poses_xyt,
                            const double * points_xy, const size_t * offsets, size_t n_scans)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (n_scans > 0 && (poses_xyt == nullptr || offsets == nullptr))
  {
    return mfail(m, NDT2D_ERR_INVALID, "add_scans: null input");
  }
  for (size_t k = 0; k < 3 * n_scans; ++k)
  {
    if (!std::isfinite(poses_xyt[k])) return mfail(m, NDT2D_ERR_INVALID, "add_scans: a scan pose is not finite");
  }
  for (size_t k = 0; k < n_scans; ++k)
  {
    if (offsets[k + 1] < offsets[k]) return mfail(m, NDT2D_ERR_INVALID, "add_scans: offsets must not decrease");
  }
  discard_ahead(m);
  m->fetched.reset();
  static const double no_points[2] = {0.0, 0.0};
  if (points_xy == nullptr) points_xy = no_points;
  static const size_t no_offsets[1] = {0};
  if (offsets == nullptr) offsets = no_offsets;
  m->have_ndt = false;
  // Device build (N1): the whole of addScans on the GPU, bit-identical to the host
  // build; "auto" uses it from kDeviceBuildFromPoints map points up, where it beats the host build
  // (round 5, whole addScans call, host / device: 0.068 / 0.134 ms at 6,480 points, 0.192 / 0.252
  // at 32,400, 0.324 / 0.340 at 64,800, 0.409 / 0.346 at 86,400, 0.64 / 0.44 at 144,000, 1.72 /
  // 0.79 at 378,000 -- experiments/build_crossover.py; the device build is ~130 us of launches
  // and a synchronisation plus 3.4 ns per point, the host build 40 us plus 4.4 ns per point).
  constexpr size_t kDeviceBuildFromPoints = 73728;   // ~100 scans of 720 beams (32,768 until round 5)
  const size_t n_map_points = n_scans > 0 ? offsets[n_scans] : 0;
  const bool on_device =
    n_scans > 0 && (m->build_mode == 2 || (m->build_mode == 0 && n_map_points >= kDeviceBuildFromPoints));
  if (on_device)
  {
    m->ndt.reset();
    // (every device of a multi-device matcher builds its own copy: the builds run side by side)
    for (size_t r = 0; r < m->devs.size(); ++r)
    {
      const int rc = ndt2d_build_grid(m->devs[r], m->resolution, m->range_max, poses_xyt, points_xy, offsets,
                                      n_scans);
      if (rc != NDT2D_OK)
      {
        const int frc = dev_fail_at(m, r, rc, "ndt2d_build_grid");
        for (ndt2d_handle h : m->devs) ndt2d_clear_grid(h);
        return frc;
      }
    }
    m->have_ndt = true;
    return NDT2D_OK;
  }
  if (m->ndt) m->spare = std::move(m->ndt);
  m->ndt = build_ndt(m->resolution, m->range_max, poses_xyt, points_xy, offsets, n_scans,
                     std::move(m->spare), m->eigen_form);
  if (!m->ndt || m->ndt->ncell() == 0)
  {
    m->ndt.reset();
    for (ndt2d_handle h : m->devs) ndt2d_clear_grid(h);
    return mfail(m, NDT2D_ERR_INVALID, "add_scans: degenerate grid extent (scan poses +- range_max must span a "
                                       "finite grid of fewer than 2^31 cells)");
  }
  // The cells that hold points travel, not the grid (ndt2d_set_grid_sparse: the install kernel
  // reads the staged list in place, two launches and no copy; 245 x 245 cells: 297 us dense ->
  // 29 us per addScans, and at 41 x 41 the list is ahead as well: 34 -> 32 us plus 7 us less
  // for the stream to be ready for the call that follows, experiments/cycle_breakdown.c).
  // (written straight into the library's pinned staging buffer: ndt2d_grid_stage_begin / _commit)
  for (size_t r = 0; r < m->devs.size(); ++r)
  {
    uint32_t * list_index = nullptr;
    double * list_cells6 = nullptr;
    int rc = ndt2d_grid_stage_begin(m->devs[r], static_cast<uint32_t>(m->ndt->size_x()),
                                    static_cast<uint32_t>(m->ndt->size_y()), m->ndt->n_touched(), &list_index,
                                    &list_cells6);
    if (rc == NDT2D_OK)
    {
      m->ndt->sparse6(list_index, list_cells6);
      rc = ndt2d_grid_stage_commit(m->devs[r], m->ndt->n_touched(), m->ndt->cell_size(), m->ndt->origin_x(),
                                   m->ndt->origin_y());
    }
    if (rc != NDT2D_OK)
    {
      const int frc = dev_fail_at(m, r, rc, "ndt2d_set_grid");
      m->ndt.reset();
      for (ndt2d_handle h : m->devs) ndt2d_clear_grid(h);
      return frc;
    }
  }
  m->have_ndt = true;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_eigenvalue_form(ndt2d_matcher * m, const char * form)
{
  NDT2D_C_TRY
  if (m == nullptr || form == nullptr) return NDT2D_ERR_INVALID;
  for (ndt2d_handle h : m->devs)
  {
    const int rc = ndt2d_set_eigenvalue_form(h, form);
    if (rc != NDT2D_OK) return mfail(m, rc, "set_eigenvalue_form: unknown form (eigen, closed)");
  }
  m->eigen_form = std::strcmp(form, "closed") == 0 ? ndt2d::kEigenFormClosed : ndt2d::kEigenFormSchur;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_build_mode(ndt2d_matcher * m, const char * mode)
{
  NDT2D_C_TRY
  if (m == nullptr || mode == nullptr) return NDT2D_ERR_INVALID;
  if (std::strcmp(mode, "auto") == 0) m->build_mode = 0;
  else if (std::strcmp(mode, "host") == 0) m->build_mode = 1;
  else if (std::strcmp(mode, "device") == 0) m->build_mode = 2;
  else return mfail(m, NDT2D_ERR_INVALID, "set_build_mode: unknown mode");
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_reset(ndt2d_matcher * m)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  discard_ahead(m);
  if (m->ndt) m->spare = std::move(m->ndt);   // `ndt_.reset()`; the storage serves the next addScans
  m->fetched.reset();
  m->have_ndt = false;
  int rc = NDT2D_OK;
  for (ndt2d_handle h : m->devs)
  {
    const int crc = ndt2d_clear_grid(h);
    if (rc == NDT2D_OK) rc = crc;
  }
  return rc;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_has_ndt(ndt2d_matcher * m) { return (m != nullptr && m->have_ndt) ? 1 : 0; }

static int prepare_search_impl(ndt2d_matcher * m, const double * scan_pose_xyt,
                               const double * points_xy, size_t n_points, size_t * n_th_out,
                               size_t * n_lin_out, size_t * n_beams_out, bool * same_out)
{
  if (same_out != nullptr) *same_out = false;
  if (m == nullptr || scan_pose_xyt == nullptr) return NDT2D_ERR_INVALID;
  if (n_points > 0 && points_xy == nullptr) return mfail(m, NDT2D_ERR_INVALID, "null points");
  subsample_into(m->scratch_beams, points_xy, n_points, m->laser_max_beams);
  const size_t use = m->scratch_beams.size() / 2;
  m->n_use = use;
  if (n_beams_out != nullptr) *n_beams_out = use;
  m->search_ready = false;
  // the scan scoreScan was just called with (src/ndt_mapper.cpp:514-515)?  Then the
  // device holds these beams already and only the tables are new.
  const bool same = m->beams_on_device && use > 0 && m->beams.size() == m->scratch_beams.size() &&
                    std::memcmp(m->beams.data(), m->scratch_beams.data(),
                                m->beams.size() * sizeof(double)) == 0;
  if (same_out != nullptr) *same_out = same;
  if (!same)
  {
    m->beams.swap(m->scratch_beams);
    ++m->beams_epoch;
    m->beams_on_device = false;
  }
  return prepare_tables(m, scan_pose_xyt, use, same ? nullptr : m->beams.data(), same, n_th_out,
                        n_lin_out);
}

int ndt2d_matcher_prepare_search(ndt2d_matcher * m, const double * scan_pose_xyt,
                                 const double * points_xy, size_t n_points, size_t * n_th_out,
                                 size_t * n_lin_out, size_t * n_beams_out)
{
  NDT2D_C_TRY
  if (m != nullptr) discard_ahead(m);
  return prepare_search_impl(m, scan_pose_xyt, points_xy, n_points, n_th_out, n_lin_out, n_beams_out,
                             nullptr);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_finish_match(ndt2d_matcher * m, const double * record, double * pose_inout,
                               double * covariance_out, double * score_out)
{
  NDT2D_C_TRY
  if (m == nullptr || record == nullptr || score_out == nullptr) return NDT2D_ERR_INVALID;
  // n_use is the N of the search prepared last (prepare_search / match_laser_scan);
  // scoring calls in between replace the device beams but leave it alone
  const size_t use = m->n_use;
  const size_t n_lin = m->dlin.size();
  const double best_score = record[0];
  if (record[1] >= 0.0 && pose_inout != nullptr && n_lin > 0)
  {
    // reference src/scan_matcher_ndt.cpp:128-134: pose = the accumulated
    // offsets of the winning candidate
    const uint64_t best_index = static_cast<uint64_t>(record[1]);   // (truncates a near-tie mark, index + 0.5)
    const uint64_t per_th = static_cast<uint64_t>(n_lin) * n_lin;
    const uint64_t ith = best_index / per_th;
    const uint64_t rem = best_index % per_th;
    if (ith >= m->dth.size()) return mfail(m, NDT2D_ERR_INVALID, "finish_match: index out of range");
    pose_inout[0] = m->dlin[rem / n_lin];
    pose_inout[1] = m->dlin[rem % n_lin];
    pose_inout[2] = m->dth[ith];
  }
  // :146 covariance = (1 / s) * k + (1 / (s * s) * u * u^T)
  if (covariance_out != nullptr)
  {
    const double * k = record + 2;
    const double * u = record + 8;
    const double s = record[11];
    const double kk[9] = {k[0], k[1], k[2], k[1], k[3], k[4], k[2], k[4], k[5]};
    const double inv_s = 1 / s;
    const double inv_s2 = 1 / (s * s);
    for (int r = 0; r < 3; ++r)
    {
      for (int c = 0; c < 3; ++c)
      {
        covariance_out[r * 3 + c] = inv_s * kk[r * 3 + c] + (inv_s2 * u[r]) * u[c];
      }
    }
  }
  // :148
  *score_out = best_score / use;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_match_scan_ex(ndt2d_matcher * m, const double * scan_pose_xyt,
                                const double * points_xy, size_t n_points,
                                double * pose_inout, double * covariance_out,
                                double * score_out, double * all_scores,
                                size_t all_scores_cap, size_t * n_candidates_out,
                                uint64_t * best_index_out)
{
  NDT2D_C_TRY
  if (m == nullptr || score_out == nullptr || scan_pose_xyt == nullptr)
  {
    return NDT2D_ERR_INVALID;
  }
  if (n_candidates_out != nullptr) *n_candidates_out = 0;
  if (best_index_out != nullptr) *best_index_out = NDT2D_NO_INDEX;
  // `if (!ndt_) return 0.0;` (reference src/scan_matcher_ndt.cpp:80): outputs untouched
  if (!m->have_ndt)
  {
    *score_out = 0.0;
    return NDT2D_OK;
  }
  m->last_multi = false;
  if (m->ahead)
  {
    // scoreScan launched a search ahead: is this the matchScan it was launched for -- the same
    // pose, the same subsampled beams, no per-candidate scores wanted?
    bool hit = all_scores == nullptr && (n_points == 0 || points_xy != nullptr) &&
               std::memcmp(m->ahead_pose, scan_pose_xyt, sizeof(m->ahead_pose)) == 0;
    if (hit)
    {
      subsample_into(m->scratch_beams, points_xy, n_points, m->laser_max_beams);
      hit = m->scratch_beams.size() == m->beams.size() && !m->beams.empty() &&
            std::memcmp(m->beams.data(), m->scratch_beams.data(), m->beams.size() * sizeof(double)) == 0;
    }
    if (hit)
    {
      // ... and still the search that is pending on the context?  (A caller may have launched or
      // fetched on ndt2d_matcher_device(m) itself in between: then the record is not ours.)
      uint64_t launched = 0, fetched = 0;
      hit = ndt2d_match_status(m->dev, &launched, &fetched) == NDT2D_OK && launched == m->ahead_launch_id &&
            fetched == m->ahead_fetch_id;
    }
    if (hit)
    {
      m->ahead = false;
      ++m->ahead_collected;
      m->score_scan_last = false;
      const size_t n_lin_a = m->dlin.size();
      if (n_candidates_out != nullptr) *n_candidates_out = m->ahead_n_th * n_lin_a * n_lin_a;
      ndt2d_match_result res;
      const int frc = ndt2d_match_fetch(m->dev, &res);
      if (frc != NDT2D_OK) return dev_fail(m, frc, "ndt2d_match_fetch");
      double rec[NDT2D_MATCH_RECORD_DOUBLES];
      rec[0] = res.best_score;
      rec[1] = res.best_index == NDT2D_NO_INDEX ? -1.0 : static_cast<double>(res.best_index) + (res.near_tie ? 0.5 : 0.0);
      for (int i = 0; i < 10; ++i) rec[2 + i] = res.acc[i];
      const int src = settle_near_tie(m, scan_pose_xyt, m->ahead_n_th, n_lin_a, m->n_use, rec);
      if (src != NDT2D_OK) return src;
      if (best_index_out != nullptr) *best_index_out = rec[1] < 0.0 ? NDT2D_NO_INDEX : static_cast<uint64_t>(rec[1]);
      return ndt2d_matcher_finish_match(m, rec, pose_inout, covariance_out, score_out);
    }
    discard_ahead(m);
  }
  // (the pair: this matchScan is of the scan and pose the scoreScan just before it scored)
  const bool after_score_scan =
    m->score_scan_last && std::memcmp(m->score_scan_pose, scan_pose_xyt, sizeof(m->score_scan_pose)) == 0;
  m->score_scan_last = false;
  size_t n_th = 0, n_lin = 0, use = 0;
  bool same_scan = false;
  int rc = prepare_search_impl(m, scan_pose_xyt, points_xy, n_points, &n_th, &n_lin, &use, &same_scan);
  if (rc != NDT2D_OK) return rc;
  // (m->beams now holds this scan's subsampled beams, whether they were uploaded or found in place)
  (void)same_scan;
  if (after_score_scan && !m->beams.empty() && m->beams.size() == m->scored.size() &&
      std::memcmp(m->beams.data(), m->scored.data(), m->beams.size() * sizeof(double)) == 0)
  {
    m->pair_seen = true;
  }
  const size_t n_cand = n_th * n_lin * n_lin;
  if (n_candidates_out != nullptr) *n_candidates_out = n_cand;

  // record = {best_score, best_index or -1, k00,k01,k02,k11,k12,k22, u0,u1,u2, s}
  double record[NDT2D_MATCH_RECORD_DOUBLES] = {0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!m->search_ready)
  {
    // No points: every candidate scores -0.0 and none is < 0 (:127-128); no
    // candidates: the loops do not run.  Either way k = u = s = 0.
    if (all_scores != nullptr)
    {
      for (size_t i = 0; i < n_cand && i < all_scores_cap; ++i) all_scores[i] = -0.0;
    }
  }
  else
  {
    std::vector<double> tmp;
    double * scores_ptr = nullptr;
    if (all_scores != nullptr)
    {
      if (all_scores_cap >= n_cand)
      {
        scores_ptr = all_scores;
      }
      else
      {
        tmp.resize(n_cand);
        scores_ptr = tmp.data();
      }
    }
    if (multi_search_wanted(m, n_th, n_lin, use))
    {
      // the lattice dealt to all devices of the matcher (ndt2d_matcher_create_multi)
      rc = multi_match(m, scan_pose_xyt, n_th, n_lin, use, false, scores_ptr, record);
      if (rc != NDT2D_OK) return rc;
    }
    else
    {
      ndt2d_match_result res;
      rc = ndt2d_match(m->dev, 0, n_th, scores_ptr, &res);
      if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_match");
      record[0] = res.best_score;
      record[1] = res.best_index == NDT2D_NO_INDEX ? -1.0 : static_cast<double>(res.best_index) + (res.near_tie ? 0.5 : 0.0);
      for (int i = 0; i < 10; ++i) record[2 + i] = res.acc[i];
    }
    if ((rc = settle_near_tie(m, scan_pose_xyt, n_th, n_lin, use, record)) != NDT2D_OK) return rc;
    if (!tmp.empty()) std::memcpy(all_scores, tmp.data(), all_scores_cap * sizeof(double));
    if (best_index_out != nullptr) *best_index_out = record[1] < 0.0 ? NDT2D_NO_INDEX : static_cast<uint64_t>(record[1]);
  }
  return ndt2d_matcher_finish_match(m, record, pose_inout, covariance_out, score_out);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_match_scan(ndt2d_matcher * m, const double * scan_pose_xyt,
                             const double * points_xy, size_t n_points, double * pose_inout,
                             double * covariance_out, double * score_out)
{
  NDT2D_C_TRY
  return ndt2d_matcher_match_scan_ex(m, scan_pose_xyt, points_xy, n_points, pose_inout,
                                     covariance_out, score_out, nullptr, 0, nullptr, nullptr);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_match_laser_scan(ndt2d_matcher * m, const double * scan_pose_xyt,
                                   const float * ranges, size_t n_ranges,
                                   const ndt2d_laser_scan * scan, double * pose_inout,
                                   double * covariance_out, double * score_out,
                                   size_t * n_points_out)
{
  NDT2D_C_TRY
  if (m == nullptr || score_out == nullptr || scan_pose_xyt == nullptr || scan == nullptr)
  {
    return NDT2D_ERR_INVALID;
  }
  if (n_ranges > 0 && ranges == nullptr) return mfail(m, NDT2D_ERR_INVALID, "null ranges");
  if (n_points_out != nullptr) *n_points_out = 0;
  // `if (!ndt_) return 0.0;` (reference src/scan_matcher_ndt.cpp:80): outputs untouched
  if (!m->have_ndt)
  {
    *score_out = 0.0;
    return NDT2D_OK;
  }
  discard_ahead(m);
  size_t n_points = 0, use = 0;
  int rc = ndt2d_set_beams_from_ranges(m->dev, ranges, n_ranges, scan, m->laser_max_beams,
                                       &n_points, &use);
  if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_set_beams_from_ranges");
  if (n_points_out != nullptr) *n_points_out = n_points;
  m->beams.clear();
  ++m->beams_epoch;
  m->beams_on_device = false;   // the device holds beams the host has no copy of
  m->n_use = use;
  size_t n_th = 0, n_lin = 0;
  rc = prepare_tables(m, scan_pose_xyt, use, nullptr, false, &n_th, &n_lin);
  if (rc != NDT2D_OK) return rc;
  double record[NDT2D_MATCH_RECORD_DOUBLES] = {0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  m->last_multi = false;
  if (m->search_ready && multi_search_wanted(m, n_th, n_lin, use))
  {
    // every device converts the ranges itself (4 B/beam to each), then takes its theta steps
    for (size_t r = 1; r < m->devs.size(); ++r)
    {
      size_t np = 0, nu = 0;
      rc = ndt2d_set_beams_from_ranges(m->devs[r], ranges, n_ranges, scan, m->laser_max_beams, &np, &nu);
      if (rc != NDT2D_OK) return dev_fail_at(m, r, rc, "ndt2d_set_beams_from_ranges");
      if (np != n_points || nu != use) return mfail(m, NDT2D_ERR_HIP, "match_laser_scan: the devices disagree on the conversion");
    }
    rc = multi_match(m, scan_pose_xyt, n_th, n_lin, use, true, nullptr, record);
    if (rc != NDT2D_OK) return rc;
    // (no host copy of the converted beams: a near-tie mark is counted, not settled)
    if (record[1] >= 0.0 && record[1] != std::floor(record[1])) ++m->adj_marked;
    if (record[1] >= 0.0) record[1] = std::floor(record[1]);
  }
  else if (m->search_ready)
  {
    ndt2d_match_result res;
    rc = ndt2d_match(m->dev, 0, n_th, nullptr, &res);
    if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_match");
    record[0] = res.best_score;
    record[1] = res.best_index == NDT2D_NO_INDEX ? -1.0 : static_cast<double>(res.best_index);
    for (int i = 0; i < 10; ++i) record[2 + i] = res.acc[i];
    if (res.near_tie) ++m->adj_marked;
  }
  return ndt2d_matcher_finish_match(m, record, pose_inout, covariance_out, score_out);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_score_poses(ndt2d_matcher * m, const double * points_xy, size_t n_points,
                              const double * poses_xyt, size_t n_poses, double * scores_out)
{
  NDT2D_C_TRY
  if (m == nullptr || scores_out == nullptr || (n_poses > 0 && poses_xyt == nullptr))
  {
    return NDT2D_ERR_INVALID;
  }
  if (n_poses == 0) return NDT2D_OK;
  discard_ahead(m);
  // `if (!ndt_) return 0.0;` (reference src/scan_matcher_ndt.cpp:159)
  if (!m->have_ndt)
  {
    for (size_t i = 0; i < n_poses; ++i) scores_out[i] = 0.0;
    return NDT2D_OK;
  }
  if (n_points > 0 && points_xy == nullptr) return mfail(m, NDT2D_ERR_INVALID, "null points");
  size_t use = 0;
  bool pending = false;
  int rc = stage_beams(m, points_xy, n_points, &use, &pending);
  if (rc != NDT2D_OK) return rc;
  if (use == 0)
  {
    // score = 0.0 / 0 (:177)
    for (size_t i = 0; i < n_poses; ++i) scores_out[i] = std::numeric_limits<double>::quiet_NaN();
    return NDT2D_OK;
  }
  m->last_multi = false;
  if (multi_poses_wanted(m, n_poses, use))
  {
    // contiguous ranges of the batch on all devices of the matcher
    if (pending)
    {
      rc = ndt2d_set_beams(m->dev, m->beams.data(), use);
      if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_set_beams");
      m->beams_on_device = true;
    }
    return multi_score_poses(m, poses_xyt, n_poses, use, scores_out, nullptr);
  }
  if (pending)
  {
    // a new scan: its beams go to the device with the scoring call itself
    rc = ndt2d_score_poses_beams(m->dev, m->beams.data(), use, poses_xyt, n_poses, scores_out);
    if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_score_poses_beams");
    m->beams_on_device = true;
    return NDT2D_OK;
  }
  rc = ndt2d_score_poses(m->dev, poses_xyt, n_poses, scores_out, nullptr);
  if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_score_poses");
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_prepare_beams(ndt2d_matcher * m, const double * points_xy, size_t n_points,
                                size_t * n_beams_out)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (n_points > 0 && points_xy == nullptr) return mfail(m, NDT2D_ERR_INVALID, "null points");
  discard_ahead(m);
  size_t use = 0;
  int rc = stage_beams(m, points_xy, n_points, &use);
  if (n_beams_out != nullptr) *n_beams_out = use;
  return rc;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_score_points(ndt2d_matcher * m, const double * points_xy, size_t n_points,
                               const double * pose_xyt, double * score_out)
{
  NDT2D_C_TRY
  if (pose_xyt == nullptr || score_out == nullptr) return NDT2D_ERR_INVALID;
  if (m != nullptr && m->single_pose_host && m->have_ndt && n_points > 0 && points_xy != nullptr &&
      m->laser_max_beams > 0 && std::min(m->laser_max_beams, n_points) <= m->single_pose_max_beams)
  {
    // One pose of a short scan -- the unchanged ParticleFilter::measure calls this once per
    // particle (reference src/particle_filter.cpp:81-87): a kernel launch and a round trip over
    // PCIe per call would cost several times the arithmetic, so the host scores it, from the host
    // NDT, in the reference's order ("the unchanged node + unchanged ParticleFilter
    // keep working via per-pose scorePoints").  Nothing on the device is touched: a search
    // launched ahead by scoreScan stays pending.
    if (const HostNdt * ndt = host_ndt(m, true))
    {
      *score_out = host_score_points(*ndt, points_xy, n_points, m->laser_max_beams, pose_xyt);
      return NDT2D_OK;
    }
  }
  return ndt2d_matcher_score_poses(m, points_xy, n_points, pose_xyt, 1, score_out);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_score_scan(ndt2d_matcher * m, const double * scan_pose_xyt,
                             const double * points_xy, size_t n_points, double * score_out)
{
  NDT2D_C_TRY
  // scoreScan(scan) = scorePoints(scan->getPoints(), scan->getPose()) (:151-154)
  if (m == nullptr || scan_pose_xyt == nullptr || score_out == nullptr) return NDT2D_ERR_INVALID;
  discard_ahead(m);
  m->last_multi = false;
  const bool ahead_wanted =
    m->ahead_enabled && m->pair_seen && m->have_ndt && (n_points == 0 || points_xy != nullptr) && !m->dth.empty() &&
    !m->dlin.empty() && !multi_search_wanted(m, m->dth.size(), m->dlin.size(), std::min(m->laser_max_beams, n_points));
  if (m->single_pose_host && m->have_ndt && n_points > 0 && points_xy != nullptr && m->laser_max_beams > 0 &&
      std::min(m->laser_max_beams, n_points) <= m->single_pose_max_beams)
  {
    if (const HostNdt * ndt = host_ndt(m, true))
    {
      // The host scores the pose (see ndt2d_matcher_score_points) -- and when the matchScan of
      // this scan is coming (`ahead`), its search is launched first and runs meanwhile.
      subsample_into(m->scored, points_xy, n_points, m->laser_max_beams);
      if (ahead_wanted)
      {
        size_t n_th = 0, n_lin = 0, use = 0;
        if (prepare_search_impl(m, scan_pose_xyt, points_xy, n_points, &n_th, &n_lin, &use, nullptr) == NDT2D_OK &&
            m->search_ready && ndt2d_match_launch(m->dev, 0, n_th, nullptr, nullptr) == NDT2D_OK)
        {
          m->ahead = true;
          (void)ndt2d_match_status(m->dev, &m->ahead_launch_id, &m->ahead_fetch_id);
          ++m->ahead_launched;
          std::memcpy(m->ahead_pose, scan_pose_xyt, sizeof(m->ahead_pose));
          m->ahead_n_th = n_th;
        }
      }
      *score_out = host_score_points(*ndt, points_xy, n_points, m->laser_max_beams, scan_pose_xyt);
      m->score_scan_last = true;
      std::memcpy(m->score_scan_pose, scan_pose_xyt, sizeof(m->score_scan_pose));
      return NDT2D_OK;
    }
  }
  if (ahead_wanted)
  {
    // The matchScan of this scan is coming (see `ahead`): its search goes onto the stream
    // behind the scoring kernel, then the score is waited for.
    size_t use = 0;
    bool pending = false;
    int rc = stage_beams(m, points_xy, n_points, &use, &pending);
    if (rc != NDT2D_OK) return rc;
    if (use > 0)
    {
      rc = ndt2d_score_poses_beams_launch(m->dev, pending ? m->beams.data() : nullptr, use, scan_pose_xyt, 1);
      if (rc == NDT2D_ERR_STATE && pending)
      {
        // more beams than travel as kernel arguments: one staged upload, then the launch on
        // the beams the device holds
        rc = ndt2d_set_beams(m->dev, m->beams.data(), use);
        if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_set_beams");
        m->beams_on_device = true;
        pending = false;
        rc = ndt2d_score_poses_beams_launch(m->dev, nullptr, use, scan_pose_xyt, 1);
      }
      if (rc == NDT2D_OK)
      {
        m->beams_on_device = true;
        size_t n_th = 0, n_lin = 0;
        // (a search that cannot be launched ahead is not scoreScan's failure: matchScan will say)
        if (prepare_tables(m, scan_pose_xyt, use, nullptr, true, &n_th, &n_lin) == NDT2D_OK && m->search_ready &&
            ndt2d_match_launch(m->dev, 0, n_th, nullptr, nullptr) == NDT2D_OK)
        {
          m->ahead = true;
          (void)ndt2d_match_status(m->dev, &m->ahead_launch_id, &m->ahead_fetch_id);
          ++m->ahead_launched;
          std::memcpy(m->ahead_pose, scan_pose_xyt, sizeof(m->ahead_pose));
          m->ahead_n_th = n_th;
          m->n_use = use;
        }
        rc = ndt2d_score_fetch(m->dev, score_out);
        if (rc != NDT2D_OK)
        {
          const int frc = dev_fail(m, rc, "ndt2d_score_fetch");   // (the message, before anything else talks to the device)
          discard_ahead(m);
          m->beams_on_device = false;
          return frc;
        }
        m->score_scan_last = true;
        m->scored = m->beams;
        std::memcpy(m->score_scan_pose, scan_pose_xyt, sizeof(m->score_scan_pose));
        return NDT2D_OK;
      }
      if (rc != NDT2D_ERR_STATE) return dev_fail(m, rc, "ndt2d_score_poses_beams_launch");
      // (not a kernel-argument launch -- more beams than travel as arguments: the ordinary call)
    }
  }
  const int rc = ndt2d_matcher_score_poses(m, points_xy, n_points, scan_pose_xyt, 1, score_out);
  if (rc == NDT2D_OK && m->have_ndt)
  {
    m->score_scan_last = true;
    m->scored = m->beams;
    std::memcpy(m->score_scan_pose, scan_pose_xyt, sizeof(m->score_scan_pose));
  }
  return rc;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_settle_near_tie(ndt2d_matcher * m, const double * scan_pose_xyt, double * record_inout)
{
  NDT2D_C_TRY
  if (m == nullptr || scan_pose_xyt == nullptr || record_inout == nullptr) return NDT2D_ERR_INVALID;
  if (!m->search_ready) return mfail(m, NDT2D_ERR_STATE, "settle_near_tie: ndt2d_matcher_prepare_search first");
  discard_ahead(m);
  return settle_near_tie(m, scan_pose_xyt, m->dth.size(), m->dlin.size(), m->n_use, record_inout);
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_adjudication(ndt2d_matcher * m, int enabled)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  m->adjudicate = enabled != 0;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_adjudication_stats(ndt2d_matcher * m, uint64_t * marked, uint64_t * changed, uint64_t * truncated)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (marked != nullptr) *marked = m->adj_marked;
  if (changed != nullptr) *changed = m->adj_changed;
  if (truncated != nullptr) *truncated = m->adj_truncated;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_single_pose_path(ndt2d_matcher * m, const char * where, size_t max_beams)
{
  NDT2D_C_TRY
  if (m == nullptr || where == nullptr) return NDT2D_ERR_INVALID;
  if (std::strcmp(where, "host") == 0) m->single_pose_host = true;
  else if (std::strcmp(where, "device") == 0) m->single_pose_host = false;
  else return mfail(m, NDT2D_ERR_INVALID, "set_single_pose_path: unknown path (host, device)");
  if (max_beams > 0) m->single_pose_max_beams = max_beams;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_search_ahead_stats(ndt2d_matcher * m, uint64_t * launched, uint64_t * collected)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (launched != nullptr) *launched = m->ahead_launched;
  if (collected != nullptr) *collected = m->ahead_collected;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_set_search_ahead(ndt2d_matcher * m, int enabled)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  discard_ahead(m);
  m->ahead_enabled = enabled != 0 ? 1 : 0;
  m->pair_seen = false;
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_pf_measure(ndt2d_matcher * m, const double * particles_xyt,
                             size_t n_particles, const double * points_xy, size_t n_points,
                             double * weights_out, double * mean_out, double * cov_inout)
{
  NDT2D_C_TRY
  if (m == nullptr || weights_out == nullptr || mean_out == nullptr || cov_inout == nullptr ||
      (n_particles > 0 && particles_xyt == nullptr))
  {
    return NDT2D_ERR_INVALID;
  }
  discard_ahead(m);
  if (n_particles > 0 && m->have_ndt && n_points > 0 && m->laser_max_beams > 0)
  {
    // weights_[i] = scorePoints(points, particle_i) (particle_filter.cpp:81-87), then
    // updateStatistics (:163-218), all on the device
    if (points_xy == nullptr) return mfail(m, NDT2D_ERR_INVALID, "null points");
    size_t use = 0;
    int rc = stage_beams(m, points_xy, n_points, &use);
    if (rc != NDT2D_OK) return rc;
    double out[NDT2D_PF_RESULT_DOUBLES];
    m->last_multi = false;
    if (multi_poses_wanted(m, n_particles, use))
    {
      rc = multi_score_poses(m, particles_xyt, n_particles, use, weights_out, out);
      if (rc != NDT2D_OK) return rc;
    }
    else
    {
      rc = ndt2d_pf_measure(m->dev, particles_xyt, n_particles, weights_out, out);
      if (rc != NDT2D_OK) return dev_fail(m, rc, "ndt2d_pf_measure");
    }
    mean_out[0] = out[1];
    mean_out[1] = out[2];
    mean_out[2] = out[3];
    cov_inout[0] = out[4];
    cov_inout[1] = out[5];
    cov_inout[3] = out[5];
    cov_inout[4] = out[6];
    cov_inout[8] += out[7];  // cov_(2,2) accumulates (:216)
    return NDT2D_OK;
  }

  // Degenerate inputs (no map / no points / no particles): every weight is the
  // constant scorePoints returns (0.0 or NaN); the statistics of those constants.
  if (n_particles > 0)
  {
    int rc = ndt2d_matcher_score_poses(m, points_xy, n_points, particles_xyt, n_particles,
                                       weights_out);
    if (rc != NDT2D_OK) return rc;
  }
  double stats[NDT2D_POSE_STATS_DOUBLES] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < n_particles; ++i)
  {
    const double w = weights_out[i];
    const double * p = particles_xyt + 3 * i;
    stats[0] += w;
    stats[1] += w * p[0];
    stats[2] += w * p[1];
    double cos_p, sin_p;
    ndt2d_cos_sin(p[2], &cos_p, &sin_p);
    stats[3] += w * cos_p;
    stats[4] += w * sin_p;
    stats[5] += w * p[0] * p[0];
    stats[6] += w * p[0] * p[1];
    stats[7] += w * p[1] * p[1];
  }
  const double sum_weight = stats[0];
  for (size_t i = 0; i < n_particles; ++i) weights_out[i] /= sum_weight;
  const double mean_x = stats[1] / sum_weight;
  const double mean_y = stats[2] / sum_weight;
  mean_out[0] = mean_x;
  mean_out[1] = mean_y;
  mean_out[2] = std::atan2(stats[4] / sum_weight, stats[3] / sum_weight);
  cov_inout[0] = stats[5] / sum_weight - mean_x * mean_x;
  cov_inout[1] = stats[6] / sum_weight - mean_x * mean_y;
  cov_inout[3] = cov_inout[1];
  cov_inout[4] = stats[7] / sum_weight - mean_y * mean_y;
  for (size_t i = 0; i < n_particles; ++i)
  {
    const double d = normalize_angle(mean_out[2] - particles_xyt[3 * i + 2]);
    cov_inout[8] += weights_out[i] * d * d;
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_grid_info(ndt2d_matcher * m, uint32_t * size_x, uint32_t * size_y,
                            double * cell_size, double * origin_x, double * origin_y)
{
  NDT2D_C_TRY
  if (m == nullptr) return NDT2D_ERR_INVALID;
  if (!m->have_ndt) return mfail(m, NDT2D_ERR_NO_GRID, "no NDT");
  int rc = ndt2d_get_grid(m->dev, nullptr, 0, size_x, size_y, cell_size, origin_x, origin_y);
  return rc == NDT2D_OK ? rc : dev_fail(m, rc, "ndt2d_get_grid");
  NDT2D_C_CATCH(m)
}

int ndt2d_matcher_grid_cells6(ndt2d_matcher * m, double * cells6_out, size_t capacity_cells)
{
  NDT2D_C_TRY
  if (m == nullptr || cells6_out == nullptr) return NDT2D_ERR_INVALID;
  if (!m->have_ndt) return mfail(m, NDT2D_ERR_NO_GRID, "no NDT");
  discard_ahead(m);
  if (m->ndt)
  {
    if (capacity_cells < m->ndt->ncell()) return mfail(m, NDT2D_ERR_INVALID, "capacity too small");
    m->ndt->pack6(cells6_out);
    return NDT2D_OK;
  }
  int rc = ndt2d_get_grid(m->dev, cells6_out, capacity_cells, nullptr, nullptr, nullptr, nullptr,
                          nullptr);
  return rc == NDT2D_OK ? rc : dev_fail(m, rc, "ndt2d_get_grid");
  NDT2D_C_CATCH(m)
}

int ndt2d_search_offsets(double size, double res, double * out, size_t cap, size_t * n_out)
{
  NDT2D_C_TRY
  if (!offsets_fit(size, res, 1u << 24)) return NDT2D_ERR_INVALID;   // (a loop that would not end, or only fill memory)
  const std::vector<double> v = search_offsets(size, res);
  if (n_out != nullptr) *n_out = v.size();
  if (out != nullptr)
  {
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_kld_resample(const double * particles_xyt, const double * weights, size_t n,
                       size_t min_particles, size_t max_particles, double kld_err, double kld_z,
                       const double * leaf_size3, const double * uniforms, size_t n_uniforms,
                       uint32_t * indices_out, size_t * n_out)
{
  NDT2D_C_TRY
  if (n_out == nullptr) return NDT2D_ERR_INVALID;
  *n_out = 0;
  if (max_particles == 0) return NDT2D_OK;
  if (n == 0 || n > 0xffffffffull || particles_xyt == nullptr || weights == nullptr ||
      leaf_size3 == nullptr || uniforms == nullptr || indices_out == nullptr ||
      n_uniforms < max_particles)
  {
    return NDT2D_ERR_INVALID;
  }
  // cumulative weights; a draw u picks the first particle with cdf > u * total
  std::vector<double> cdf(n);
  double total = 0.0;
  for (size_t i = 0; i < n; ++i)
  {
    total += weights[i];
    cdf[i] = total;
  }
  // KDTree::insert's key (kd_tree.hpp:95-98); one leaf per distinct key
  struct Key
  {
    int32_t k[3];
    bool operator==(const Key & o) const { return k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2]; }
  };
  struct KeyHash
  {
    size_t operator()(const Key & key) const
    {
      uint64_t h = 0x9e3779b97f4a7c15ull;
      for (int d = 0; d < 3; ++d)
      {
        h ^= static_cast<uint32_t>(key.k[d]);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 32;
      }
      return static_cast<size_t>(h);
    }
  };
  std::unordered_set<Key, KeyHash> leaves;   // kd_tree_.clear() (:97)
  leaves.reserve(1024);
  size_t Mx = max_particles;                 // (:105)
  size_t count = 0;
  while (count < std::max(min_particles, Mx))
  {
    const double u = uniforms[count] * total;
    size_t p = static_cast<size_t>(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
    if (p >= n) p = n - 1;   // u rounded up to the total (or a NaN weight): the last particle
    Key key;
    for (int d = 0; d < 3; ++d)
    {
      const double q = particles_xyt[3 * p + d] / leaf_size3[d];
      // static_cast<int> of the reference; out-of-range values are pinned to the ends
      key.k[d] = q >= 2147483647.0 ? 2147483647
                 : (q <= -2147483648.0 ? (-2147483647 - 1) : (q == q ? static_cast<int32_t>(q) : 0));
    }
    leaves.insert(key);
    indices_out[count++] = static_cast<uint32_t>(p);
    const size_t k = leaves.size();
    if (k > 1)
    {
      const double a = (k - 1) / (2.0 * kld_err);
      const double b = 2.0 / (9.0 * (k - 1));
      const double c = 1.0 - b + std::sqrt(b) * kld_z;
      const double mx = a * c * c * c;       // size_t Mx = double (:125)
      Mx = mx >= 1.8446744073709552e19 ? ~static_cast<size_t>(0)
                                        : (mx > 0.0 ? static_cast<size_t>(mx) : 0);
    }
    if (count >= max_particles) break;       // (:129-132)
  }
  *n_out = count;
  return NDT2D_OK;
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_host_build_grid(double ndt_resolution, double range_max, const double * poses_xyt,
                          const double * points_xy, const size_t * offsets, size_t n_scans,
                          double * cells6_out, size_t capacity_cells, uint32_t * size_x,
                          uint32_t * size_y, double * origin_x, double * origin_y)
{
  NDT2D_C_TRY
  return ndt2d_host_build_grid_ex(ndt_resolution, range_max, poses_xyt, points_xy, offsets, n_scans, 0u, cells6_out,
                                  capacity_cells, size_x, size_y, origin_x, origin_y);
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_host_build_grid_ex(double ndt_resolution, double range_max, const double * poses_xyt,
                             const double * points_xy, const size_t * offsets, size_t n_scans, unsigned flags,
                             double * cells6_out, size_t capacity_cells, uint32_t * size_x,
                             uint32_t * size_y, double * origin_x, double * origin_y)
{
  NDT2D_C_TRY
  if (!(ndt_resolution > 0.0) || (n_scans > 0 && (poses_xyt == nullptr || offsets == nullptr)))
  {
    return NDT2D_ERR_INVALID;
  }
  static const double no_points[2] = {0.0, 0.0};
  static const size_t no_offsets[1] = {0};
  std::unique_ptr<HostNdt> ndt = build_ndt(ndt_resolution, range_max, poses_xyt,
                                           points_xy ? points_xy : no_points,
                                           offsets ? offsets : no_offsets, n_scans, nullptr,
                                           (flags & NDT2D_BUILD_CLOSED_FORM) ? ndt2d::kEigenFormClosed
                                                                             : ndt2d::kEigenFormSchur,
                                           (flags & NDT2D_BUILD_SEQUENTIAL) == 0);
  if (!ndt) return NDT2D_ERR_INVALID;   // (degenerate extent: non-finite poses / range_max, >= 2^31 cells)
  if (size_x) *size_x = static_cast<uint32_t>(ndt->size_x());
  if (size_y) *size_y = static_cast<uint32_t>(ndt->size_y());
  if (origin_x) *origin_x = ndt->origin_x();
  if (origin_y) *origin_y = ndt->origin_y();
  if (cells6_out != nullptr)
  {
    if (capacity_cells < ndt->ncell()) return NDT2D_ERR_INVALID;
    ndt->pack6(cells6_out);
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(nullptr)
}

// ---------------------------------------------------------------------------
// Synthetic workload generator
// ---------------------------------------------------------------------------

}  // extern "C"

namespace
{

struct SplitMix64
{
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next()
  {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return static_cast<double>(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal()
  {
    // Box-Muller, one value per two uniforms
    const double u1 = 1.0 - uniform();
    const double u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
};

bool pillar_in_cell(const ndt2d_world & w, long ci, long cj, double * cx, double * cy)
{
  *cx = w.pillar_pitch * static_cast<double>(ci) + 0.5 * w.pillar_pitch;
  *cy = w.pillar_pitch * static_cast<double>(cj) + 0.5 * w.pillar_pitch;
  return std::fabs(*cx) + w.pillar_half < w.room_half && std::fabs(*cy) + w.pillar_half < w.room_half;
}

// Distance along (dx, dy) from (ox, oy) to the nearest surface.
double raycast(const ndt2d_world & w, double ox, double oy, double dx, double dy)
{
  const double inf = std::numeric_limits<double>::infinity();
  // room walls (origin inside the room)
  double t_wall = inf;
  if (dx > 0) t_wall = std::min(t_wall, (w.room_half - ox) / dx);
  if (dx < 0) t_wall = std::min(t_wall, (-w.room_half - ox) / dx);
  if (dy > 0) t_wall = std::min(t_wall, (w.room_half - oy) / dy);
  if (dy < 0) t_wall = std::min(t_wall, (-w.room_half - oy) / dy);
  if (!(w.pillar_pitch > 0.0) || !(w.pillar_half > 0.0)) return t_wall;

  // walk the pillar lattice cells the ray crosses (one pillar per cell)
  const double pitch = w.pillar_pitch;
  long ci = static_cast<long>(std::floor(ox / pitch));
  long cj = static_cast<long>(std::floor(oy / pitch));
  const long step_i = dx > 0 ? 1 : -1, step_j = dy > 0 ? 1 : -1;
  double t_max_x = dx != 0 ? ((dx > 0 ? (ci + 1) * pitch : ci * pitch) - ox) / dx : inf;
  double t_max_y = dy != 0 ? ((dy > 0 ? (cj + 1) * pitch : cj * pitch) - oy) / dy : inf;
  const double t_dx = dx != 0 ? pitch / std::fabs(dx) : inf;
  const double t_dy = dy != 0 ? pitch / std::fabs(dy) : inf;
  double t_enter = 0.0;
  while (t_enter <= t_wall)
  {
    double cx, cy;
    if (pillar_in_cell(w, ci, cj, &cx, &cy))
    {
      // slab test against [cx - h, cx + h] x [cy - h, cy + h]
      double t0 = 0.0, t1 = inf;
      bool hit = true;
      const double lo[2] = {cx - w.pillar_half, cy - w.pillar_half};
      const double hi[2] = {cx + w.pillar_half, cy + w.pillar_half};
      const double o[2] = {ox, oy}, d[2] = {dx, dy};
      for (int a = 0; a < 2 && hit; ++a)
      {
        if (d[a] == 0.0)
        {
          if (o[a] < lo[a] || o[a] > hi[a]) hit = false;
        }
        else
        {
          double ta = (lo[a] - o[a]) / d[a], tb = (hi[a] - o[a]) / d[a];
          if (ta > tb) std::swap(ta, tb);
          t0 = std::max(t0, ta);
          t1 = std::min(t1, tb);
          if (t0 > t1) hit = false;
        }
      }
      if (hit && t0 > 0.0 && t0 < t_wall) return t0;
    }
    if (t_max_x < t_max_y)
    {
      t_enter = t_max_x;
      t_max_x += t_dx;
      ci += step_i;
    }
    else
    {
      t_enter = t_max_y;
      t_max_y += t_dy;
      cj += step_j;
    }
  }
  return t_wall;
}

}  // namespace

extern "C" {

int ndt2d_synth_scan(const ndt2d_world * world, const double * pose_xyt, size_t n_beams,
                     double noise_sigma, uint64_t seed, double * points_xy)
{
  NDT2D_C_TRY
  if (world == nullptr || pose_xyt == nullptr || points_xy == nullptr || n_beams == 0)
  {
    return NDT2D_ERR_INVALID;
  }
  if (std::fabs(pose_xyt[0]) >= world->room_half || std::fabs(pose_xyt[1]) >= world->room_half)
  {
    return NDT2D_ERR_INVALID;
  }
  SplitMix64 rng(seed);
  const double step = 2.0 * M_PI / static_cast<double>(n_beams);
  for (size_t k = 0; k < n_beams; ++k)
  {
    const double ang = -M_PI + static_cast<double>(k) * step;
    const double wa = pose_xyt[2] + ang;
    double r = raycast(*world, pose_xyt[0], pose_xyt[1], std::cos(wa), std::sin(wa));
    r += noise_sigma * rng.normal();
    points_xy[2 * k] = r * std::cos(ang);
    points_xy[2 * k + 1] = r * std::sin(ang);
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_synth_pose_blocked(const ndt2d_world * world, double x, double y, double margin)
{
  NDT2D_C_TRY
  if (world == nullptr || !(world->pillar_pitch > 0.0)) return 0;
  const long ci = static_cast<long>(std::floor(x / world->pillar_pitch));
  const long cj = static_cast<long>(std::floor(y / world->pillar_pitch));
  double cx, cy;
  if (!pillar_in_cell(*world, ci, cj, &cx, &cy)) return 0;
  return (std::fabs(x - cx) <= world->pillar_half + margin &&
          std::fabs(y - cy) <= world->pillar_half + margin)
           ? 1
           : 0;
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_synth_uniform(uint64_t seed, size_t n, double * out)
{
  NDT2D_C_TRY
  if (out == nullptr) return NDT2D_ERR_INVALID;
  SplitMix64 rng(seed);
  for (size_t i = 0; i < n; ++i) out[i] = rng.uniform();
  return NDT2D_OK;
  NDT2D_C_CATCH(nullptr)
}

}  // extern "C"
