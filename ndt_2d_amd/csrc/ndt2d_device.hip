// Device layer of the C-ABI (include/ndt2d_hip.h, section 1): one context per
// (plugin instance, GPU) holding the HBM-resident NDT grid, beams and search
// tables, and launching the kernels of ndt2d_kernels.hip on its own stream.
// No CPU compute path exists here: every compute entry point needs the GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "ndt2d_guard.h"
#include "ndt2d_hip.h"
#include "ndt2d_kernels.h"

using ndt2d::GridDesc;
using ndt2d::kCellDoubles;
using ndt2d::kCellStrideGlobal;

struct DeviceBuffer
{
  double * ptr = nullptr;
  size_t cap = 0;  // doubles
};

// Pinned host staging for small uploads: the copy is truly asynchronous and the
// caller's buffer is free as soon as the call returns.
struct PinnedStage
{
  double * ptr = nullptr;
  double * dev = nullptr;   // the same memory as the device addresses it (kernels may read it in place)
  size_t cap = 0;  // doubles
  // the stream position (ndt2d_context::queued) behind the buffer's last queued reader: the
  // buffer may be written again once the stream is known to have passed it
  bool pending = false;
  uint64_t needed = 0;
};

// Pieces a large pose batch from host memory is cut into (see pipelined_pose_batch)
#define NDT2D_PIPELINE_PIECES 16

struct ndt2d_context
{
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  bool has_grid = false;
  GridDesc grid{};
  DeviceBuffer cells_lds_image;
  DeviceBuffer cells_global;
  DeviceBuffer occ_bits;  // uint32 words stored in a double buffer
  DeviceBuffer cell_bytes;  // per-cell occupancy-map bytes of the extended grid
  // what a list install left to do (bytes_job.n > 0): the map bytes around the listed cells.
  // Only the small-lattice search reads them: the job rides along in the next few-pose launch
  // (the mapper's scoreScan) or runs ahead of the next search, whichever comes first.
  ndt2d::SparseBytesJob bytes_job{};
  // ndt2d_grid_stage_begin ... _commit: capacity (0: none open) and grid size of the open list
  size_t stage_cap = 0;
  uint32_t stage_sx = 0, stage_sy = 0;
  bool stage_may_compact = false;
  DeviceBuffer compact;     // [cells6 | compact records | cell ranks]: ndt2d_set_grid's upload
  DeviceBuffer cells6;    // raw {mean, information, n} records of a device-built grid
  const double * cells6_ptr = nullptr;  // ... of the installed grid, wherever they live
  // a grid installed as the list of its touched cells (ndt2d_set_grid_sparse): the list on the
  // device (inside `compact`), from which ndt2d_get_grid makes the dense records on demand
  DeviceBuffer ranks;       // cell -> compact record table of such a grid (uint16 per cell)
  // particle scoring on a grid whose occupancy bitmap does not fit LDS: one bit per block of
  // cells, made from the bitmap on the first launch that needs it (-1: not made for this grid)
  DeviceBuffer coarse_bits;
  int coarse_log2 = -1;
  // ... and the small-lattice search on a window wider than 256 cells: map bytes per block of cells
  DeviceBuffer block_bytes;
  int block_bytes_log2 = -1;
  uint32_t sparse_n = 0;
  const uint32_t * sparse_index = nullptr;
  const double * sparse_cells6 = nullptr;
  // device NDT build scratch + host staging that must outlive the async copies
  DeviceBuffer b_points, b_scans, b_offsets, b_world, b_keys, b_vals, b_temp, b_seg;
  std::vector<double> stage_scans;
  std::vector<uint32_t> stage_offsets;
  std::vector<uint32_t> stage_seen;   // list install: one bit per cell already listed

  DeviceBuffer beams;
  size_t n_beams = 0;
  PinnedStage stage_beams, stage_tables, stage_grid, stage_call;
  // where the launches read beams and tables: the buffers of ndt2d_set_beams /
  // ndt2d_set_search, or one buffer filled by a single copy (ndt2d_set_search_beams)
  const double * beams_ptr = nullptr;
  const double * tables_ptr = nullptr;
  DeviceBuffer call_dev;

  // Results of small calls go straight into host-coherent pinned memory, followed by a
  // sequence number the host spins on (a stream synchronisation costs ~4 us more):
  //   [0..11] match record   [16] match flag   [24] score flag   [32..2079] scores / weights
  //   [2080..2087] particle statistics
  // Stream positions: `queued` counts the marks handed out behind queued work
  // (stage_mark), `reached` is the newest mark the in-order stream is known to have
  // passed -- a result flag seen by the host proves everything queued before that kernel done.
  uint64_t queued = 0, reached = 0;
  uint64_t match_pos = 0, few_pos = 0;   // `queued` when the pending search / few-pose launch went out
  double * host_res = nullptr;      // host address
  double * host_res_dev = nullptr;  // the same memory as the GPU addresses it
  unsigned long long seq = 0;       // last sequence number handed out
  unsigned long long match_seq = 0; // ... to the pending match launch
  unsigned long long * done_words = nullptr;   // score_few launches: one word per pose

  DeviceBuffer tables;  // dth | cos | sin | dlin
  // ndt2d_set_search keeps the tables in its pinned staging buffer and leaves the upload
  // to the launch: the small-lattice search takes them as kernel arguments (no copy)
  bool tables_uploaded = false;
  size_t n_th = 0, n_lin = 0;
  double pose_x = 0.0, pose_y = 0.0;
  double dlin_absmax = 0.0;
  double patch_span = -1.0;   // see MatchArgs::patch_span
  double beam_rmax = 0.0;
  bool has_search = false;

  DeviceBuffer ws_match, ws_poses, record, stats, outer;
  DeviceBuffer tmp_scores, tmp_poses, tmp_noise;
  // Pipelined pose batches (pipelined_pose_batch): the uploads of a large batch from host memory
  // run on their own stream, piece by piece, under the scoring of the piece before
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  hipEvent_t pipe_up[NDT2D_PIPELINE_PIECES] = {}, pipe_scored[NDT2D_PIPELINE_PIECES] = {}, pipe_idle = nullptr;
  DeviceBuffer piece_stats;   // [pieces][8] moment sums
  int pipeline_pieces = 0;    // 0: default (ndt2d_set_pipeline_pieces)
  int last_pieces = 1;        // pieces of the last host pose batch (ndt2d_last_pipeline_pieces)
  DeviceBuffer near_list;   // ndt2d_match_near_best: {count, indices}
  // LaserScan conversion: ranges (floats), points, {n, rmax | n, use, rmax}
  DeviceBuffer scan_ranges, scan_points, scan_info;
  size_t n_scan_points = 0;

  // HIP events around the dominant kernel of each launch: a ring, so that a caller
  // can queue launches back to back and read their durations afterwards
  hipEvent_t ring0[NDT2D_TIMING_HISTORY] = {}, ring1[NDT2D_TIMING_HISTORY] = {};
  uint64_t n_timed_launches = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // the current launch's pair
  bool timing = true;   // record the event pairs (ndt2d_set_timing)
  bool timed = false;
  int last_kernels = 0;
  const char * last_variant = "";
  int force_variant = ndt2d::kVariantAuto;
  bool batched_only = false;   // "batched": small pose batches stay on the batched kernels
  int eigen_form = 0;          // ndt2d_set_eigenvalue_form (ndt2d_eigen2.h): 0 Eigen's EigenSolver, 1 closed form

  bool match_pending = false;
  uint64_t match_launches = 0, match_fetches = 0;   // searches launched / fetched so far (ndt2d_match_status)
  // a few-pose launch whose results have not been collected (ndt2d_score_poses_beams_launch)
  bool few_pending = false;
  unsigned long long few_seq = 0;
  size_t few_n_poses = 0;
  bool few_stats = false;
  // the moment sums of a sharded particle launch on their way to the host (ndt2d_pose_sums_launch)
  bool sums_pending = false;
  unsigned long long sums_seq = 0;
  uint64_t sums_pos = 0;
  uint64_t last_candidates = 0;
};

namespace
{

int fail(ndt2d_context * h, int code, const std::string & msg)
{
  if (h != nullptr) h->err = msg;
  return code;
}

#ifdef NDT2D_TEST_HOOKS
// Test builds only (libndt2d_hip_hooks.so): the k-th search launch / particle launch from now on
// fails with NDT2D_ERR_HIP (ndt2d_test_fail_launch(k); 0: none) -- how tests/test_gpu_multi_failure.py
// makes ONE device of a multi-device call fail while the others are in flight.
std::atomic<int> g_test_fail_launch{0};
bool test_launch_fails()
{
  int v = g_test_fail_launch.load();
  while (v > 0)
  {
    if (g_test_fail_launch.compare_exchange_weak(v, v - 1)) return v == 1;
  }
  return false;
}
#define NDT2D_TEST_MAYBE_FAIL(h, what) do { if (test_launch_fails()) return fail(h, NDT2D_ERR_HIP, what ": failure injected by the test hook"); } while (0)
#else
#define NDT2D_TEST_MAYBE_FAIL(h, what) do {} while (0)
#endif

// ndt2d_guard.h: where the text of an exception caught at the C boundary goes
void guard_note(ndt2d_context * h, const char * what) noexcept
{
  if (h == nullptr) return;
  try
  {
    h->err = what;
  }
  catch (...)
  {
  }
}
void guard_note(std::nullptr_t, const char *) noexcept {}

int fail_hip(ndt2d_context * h, hipError_t e, const char * what)
{
  std::string msg = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();  // clear sticky state
  return fail(h, NDT2D_ERR_HIP, msg);
}

#define NDT2D_HIP(h, call)                                 \
  do                                                       \
  {                                                        \
    hipError_t e__ = (call);                               \
    if (e__ != hipSuccess) return fail_hip(h, e__, #call); \
  } while (0)

// hipStreamSynchronize, noting how far the stream has come (see ndt2d_context::reached)
#define NDT2D_SYNC(h)                                          \
  do {                                                         \
    const uint64_t q_ = (h)->queued;                           \
    NDT2D_HIP(h, hipStreamSynchronize((h)->stream));           \
    if ((h)->reached < q_) (h)->reached = q_;                  \
  } while (0)

int ensure(ndt2d_context * h, DeviceBuffer & b, size_t doubles)
{
  if (doubles <= b.cap && b.ptr != nullptr) return NDT2D_OK;
  if (b.ptr != nullptr)
  {
    NDT2D_SYNC(h);
    NDT2D_HIP(h, hipFree(b.ptr));
    b.ptr = nullptr;
    b.cap = 0;
  }
  // (an eighth of headroom: the mapper's grid follows the scan poses and grows by a row or
  // a column of cells now and then -- not a free + malloc + stream synchronisation each time)
  size_t cap = doubles < 64 ? 64 : doubles + doubles / 8;
  NDT2D_HIP(h, hipMalloc(reinterpret_cast<void **>(&b.ptr), cap * sizeof(double)));
  b.cap = cap;
  return NDT2D_OK;
}

// Wait until nothing queued reads `st` any more.  After a sequence of synchronous calls the
// stream is already known to be past the buffer's mark (the call that used it ended with a result
// flag); only a caller that queues launch upon launch without fetching waits here.
int stage_wait(ndt2d_context * h, PinnedStage & st)
{
  if (st.pending)
  {
    // (nobody has fetched a result since the mark -- back-to-back addScans, say: the stream is
    // asked.  A word the reader raises in host memory when it has read the buffer was tried: it
    // costs the install kernel 1.3 us in every mapper cycle to save 8 us in a sequence no node
    // produces; hipStreamQuery is slower than the synchronisation.)
    if (h->reached < st.needed) NDT2D_SYNC(h);
    st.pending = false;
  }
  return NDT2D_OK;
}

// Make `st` ready to take `doubles` values: wait for the previous copy out of it.
int stage_acquire(ndt2d_context * h, PinnedStage & st, size_t doubles)
{
  const int wrc = stage_wait(h, st);
  if (wrc != NDT2D_OK) return wrc;
  if (doubles > st.cap || st.ptr == nullptr)
  {
    if (st.ptr != nullptr) NDT2D_HIP(h, hipHostFree(st.ptr));
    st.ptr = nullptr;
    st.cap = 0;
    const size_t cap = doubles < 4096 ? 4096 : doubles;
    NDT2D_HIP(h, hipHostMalloc(reinterpret_cast<void **>(&st.ptr), cap * sizeof(double), hipHostMallocDefault));
    st.cap = cap;
    st.dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&st.dev), st.ptr, 0) != hipSuccess)
    {
      (void)hipGetLastError();
      st.dev = nullptr;
    }
  }
  return NDT2D_OK;
}

// The copy out of a staging buffer, and the mark that tells when the buffer may be written
// again: a POSITION in the in-order stream, not an event.  A recorded event holds the stream
// up for ~5 us in front of whatever is queued next (experiments/kernel_gaps.py; in the mapper's
// cycle 5.8 us between the install kernel and the scoreScan behind it,
// experiments/r03_cycle.sh), and the question it answers is already answered: every
// synchronous call ends with a result flag (or a stream synchronisation) seen by the host,
// which proves everything queued before that kernel done (ndt2d_context::reached).
int stage_copy(ndt2d_context * h, PinnedStage & st, double * dst, size_t doubles)
{
  NDT2D_HIP(h, hipMemcpyAsync(dst, st.ptr, doubles * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return NDT2D_OK;
}

int stage_mark(ndt2d_context * h, PinnedStage & st)
{
  st.needed = ++h->queued;
  st.pending = true;
  return NDT2D_OK;
}

int stage_submit(ndt2d_context * h, PinnedStage & st, double * dst, size_t doubles)
{
  const int rc = stage_copy(h, st, dst, doubles);
  return rc != NDT2D_OK ? rc : stage_mark(h, st);
}

void release(PinnedStage & st)
{
  if (st.ptr != nullptr) (void)hipHostFree(st.ptr);
  st = PinnedStage{};
}

void release(DeviceBuffer & b)
{
  if (b.ptr != nullptr) (void)hipFree(b.ptr);
  b.ptr = nullptr;
  b.cap = 0;
}

// angles::normalize_angle / shortest_angular_distance (ROS `angles`)
double normalize_angle(double a)
{
  const double r = std::fmod(a + M_PI, 2.0 * M_PI);
  return r <= 0.0 ? r + M_PI : r - M_PI;
}
double angle_diff(double from, double to) { return normalize_angle(to - from); }

// The per-call part of MotionModel::sample (reference src/motion_model.cpp:48-72):
// decompose the odometry delta and derive the three (mean, sigma) pairs, narrowed
// to float as std::normal_distribution<float> holds them.
ndt2d::MotionParams motion_params(double dx, double dy, double dth, const double * a)
{
  const double trans = std::hypot(dx, dy);
  const double rot1 = (trans > 0.01) ? std::atan2(dy, dx) : 0.0;
  const double rot2 = angle_diff(rot1, dth);
  const double rot1_ = std::min(std::fabs(angle_diff(rot1, 0.0)), std::fabs(angle_diff(rot1, M_PI)));
  const double rot2_ = std::min(std::fabs(angle_diff(rot2, 0.0)), std::fabs(angle_diff(rot2, M_PI)));
  const double sigma_rot1 = std::sqrt(a[0] * rot1_ * rot1_ + a[1] * trans * trans);
  const double sigma_trans =
    std::sqrt(a[2] * trans * trans + a[3] * rot1_ * rot1_ + a[3] * rot2_ * rot2_);
  const double sigma_rot2 = std::sqrt(a[0] * rot2_ * rot2_ + a[1] * trans * trans);
  ndt2d::MotionParams p;
  p.rot1 = static_cast<float>(rot1);
  p.trans = static_cast<float>(trans);
  p.rot2 = static_cast<float>(rot2);
  p.sigma_rot1 = static_cast<float>(sigma_rot1);
  p.sigma_trans = static_cast<float>(sigma_trans);
  p.sigma_rot2 = static_cast<float>(sigma_rot2);
  return p;
}

constexpr int kScoreFlagSlot = 24;
constexpr int kScoreSlot = 32;                                             // scores / weights of a small batch
constexpr int kPfOutSlot = kScoreSlot + static_cast<int>(ndt2d::kFewPosesMax);   // its statistics
constexpr int kHostResDoubles = kPfOutSlot + 32;

int ensure_host_res(ndt2d_context * h)
{
  if (h->host_res != nullptr) return NDT2D_OK;
  void * p = nullptr;
  NDT2D_HIP(h, hipHostMalloc(&p, kHostResDoubles * sizeof(double),
                             hipHostMallocCoherent | hipHostMallocMapped));
  std::memset(p, 0, kHostResDoubles * sizeof(double));
  void * d = nullptr;
  hipError_t e = hipHostGetDevicePointer(&d, p, 0);
  if (e != hipSuccess)
  {
    (void)hipHostFree(p);
    return fail_hip(h, e, "hipHostGetDevicePointer");
  }
  void * counter = nullptr;
  const size_t done_bytes = (static_cast<size_t>(ndt2d::kFewPosesMax) + 8) * sizeof(unsigned long long);
  e = hipMalloc(&counter, done_bytes);
  if (e != hipSuccess)
  {
    (void)hipHostFree(p);
    return fail_hip(h, e, "hipMalloc");
  }
  e = hipMemsetAsync(counter, 0, done_bytes, h->stream);
  if (e != hipSuccess)
  {
    (void)hipHostFree(p);
    (void)hipFree(counter);
    return fail_hip(h, e, "hipMemsetAsync");
  }
  h->host_res = static_cast<double *>(p);
  h->host_res_dev = static_cast<double *>(d);
  h->done_words = static_cast<unsigned long long *>(counter);
  return NDT2D_OK;
}

// Wait until the kernel that was given `seq` has published its results at `slot`:
// spin on the flag for a while (the GPU writes it over PCIe into coherent host memory;
// ~4 us sooner than a stream synchronisation returns), then fall back to the stream.
int wait_host_flag(ndt2d_context * h, int slot, unsigned long long seq, uint64_t position)
{
  volatile unsigned long long * flag =
    reinterpret_cast<volatile unsigned long long *>(h->host_res + slot);
  for (int outer = 0; outer < 64; ++outer)
  {
    for (int i = 0; i < 2048; ++i)
    {
      const unsigned long long seen = *flag;
      if (seen == seq)
      {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (h->reached < position) h->reached = position;   // the stream is in order
        return NDT2D_OK;
      }
      if (seen == (seq | ndt2d::kHostFlagGaveUp))
      {
        // the launch's last block waited in vain for another block's result (its bounded poll):
        // this call has no result; the kernel ended normally and the context stays usable
        if (h->reached < position) h->reached = position;
        return fail(h, NDT2D_ERR_HIP, "the kernel gave up waiting for one of its blocks' results (bounded poll)");
      }
      __builtin_ia32_pause();
    }
    // a failed launch / faulted kernel never raises the flag: ask the stream now and then
    const hipError_t q = hipStreamQuery(h->stream);
    if (q != hipErrorNotReady && q != hipSuccess) return fail_hip(h, q, "hipStreamQuery");
  }
  NDT2D_SYNC(h);
  if (*flag == (seq | ndt2d::kHostFlagGaveUp))
  {
    return fail(h, NDT2D_ERR_HIP, "the kernel gave up waiting for one of its blocks' results (bounded poll)");
  }
  if (*flag != seq) return fail(h, NDT2D_ERR_HIP, "result flag was not raised");
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (h->reached < position) h->reached = position;
  return NDT2D_OK;
}

// Host-side figures of the translation lattice: its extent and the widest 8-step patch.
void lattice_extent(ndt2d_context * h, const double * dlin, size_t n_lin)
{
  h->dlin_absmax = 0.0;
  h->patch_span = 0.0;
  for (size_t i = 0; i < n_lin; ++i)
  {
    if (std::fabs(dlin[i]) > h->dlin_absmax) h->dlin_absmax = std::fabs(dlin[i]);
  }
  for (size_t i = 0; i < n_lin && h->patch_span >= 0.0; i += 8)
  {
    const size_t last = i + 7 < n_lin ? i + 7 : n_lin - 1;
    for (size_t k = i; k < last; ++k)
    {
      if (!(dlin[k + 1] >= dlin[k])) h->patch_span = -1.0;   // not ascending (or NaN): no claim
    }
    if (h->patch_span >= 0.0 && dlin[last] - dlin[i] > h->patch_span) h->patch_span = dlin[last] - dlin[i];
  }
}

// A small batch of poses (<= kFewPosesMax) through the block-per-pose kernel: one launch,
// results through host-coherent memory.  arg_beams (optional): the context's new beams,
// handed to the kernel as arguments (<= kArgBeams) -- the caller has made h->beams large
// enough.  stats: the whole of ParticleFilter::measure (normalised weights into h_scores,
// NDT2D_PF_RESULT_DOUBLES into h_out).
int run_few(ndt2d_context * h, const double * arg_beams, size_t n_beams, const double * h_poses,
            size_t n_poses, bool stats, double * h_scores, double * h_out);
// ... and in two steps: the launch, and the wait for its results (one launch may be pending)
int run_few_launch(ndt2d_context * h, const double * arg_beams, size_t n_beams, const double * h_poses,
                   size_t n_poses, bool stats);
int run_few_fetch(ndt2d_context * h, double * h_scores, double * h_out);

// Take the next pair of timing events (created on first use) as h->ev0 / h->ev1.
int next_timing_slot(ndt2d_context * h)
{
  const size_t slot = static_cast<size_t>(h->n_timed_launches % NDT2D_TIMING_HISTORY);
  if (h->ring0[slot] == nullptr) NDT2D_HIP(h, hipEventCreate(&h->ring0[slot]));
  if (h->ring1[slot] == nullptr) NDT2D_HIP(h, hipEventCreate(&h->ring1[slot]));
  h->ev0 = h->ring0[slot];
  h->ev1 = h->ring1[slot];
  ++h->n_timed_launches;
  return NDT2D_OK;
}

// The address a kernel can use for host memory allocated with ndt2d_host_alloc
// (pinned, mapped); nullptr for ordinary (pageable) host memory.
template <class T>
T * device_view(T * host_ptr)
{
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, host_ptr) != hipSuccess)
  {
    (void)hipGetLastError();   // pageable memory: "invalid value", not an error here
    return nullptr;
  }
  if (attr.type != hipMemoryTypeHost || attr.devicePointer == nullptr) return nullptr;
  return static_cast<T *>(attr.devicePointer);
}

// max |beam|; an infinite reach (NaN / inf beam) disables the windowed lane mapping
double beam_reach(const double * beams_xy, size_t n_beams)
{
  double rmax = 0.0;
  for (size_t i = 0; i < n_beams; ++i)
  {
    const double r = std::hypot(beams_xy[2 * i], beams_xy[2 * i + 1]);
    if (r > rmax) rmax = r;
    if (std::isnan(r)) rmax = std::numeric_limits<double>::infinity();
  }
  return rmax;
}

bool is_pow2(double v)
{
  if (!(v > 0.0) || !std::isfinite(v)) return false;
  int e = 0;
  return std::frexp(v, &e) == 0.5 && std::fpclassify(v) == FP_NORMAL &&
         std::fpclassify(1.0 / v) == FP_NORMAL;
}

}  // namespace

extern "C" {

int ndt2d_abi_version(void) { return NDT2D_ABI_VERSION; }

int ndt2d_create(ndt2d_handle * out, int device_id)
{
  NDT2D_C_TRY
  if (out == nullptr) return NDT2D_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
  {
    (void)hipGetLastError();
    return NDT2D_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= count) return NDT2D_ERR_INVALID;
  ndt2d_context * h = new (std::nothrow) ndt2d_context();
  if (h == nullptr) return NDT2D_ERR_INVALID;
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess)
  {
    (void)hipGetLastError();
    delete h;
    return NDT2D_ERR_HIP;
  }
  h->stream = h->own_stream;
  *out = h;
  return NDT2D_OK;
  NDT2D_C_CATCH(nullptr)
}

int ndt2d_destroy(ndt2d_handle h)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  release(h->cells_lds_image);
  release(h->cells_global);
  release(h->occ_bits);
  release(h->cell_bytes);
  release(h->compact);
  release(h->cells6);
  release(h->ranks);
  release(h->coarse_bits);
  release(h->block_bytes);
  release(h->call_dev);
  release(h->stage_grid);
  release(h->stage_call);
  if (h->host_res != nullptr) (void)hipHostFree(h->host_res);
  if (h->done_words != nullptr) (void)hipFree(h->done_words);
  release(h->b_points);
  release(h->b_scans);
  release(h->b_offsets);
  release(h->b_world);
  release(h->b_keys);
  release(h->b_vals);
  release(h->b_temp);
  release(h->b_seg);
  release(h->beams);
  release(h->stage_beams);
  release(h->stage_tables);
  release(h->tables);
  release(h->ws_match);
  release(h->outer);
  release(h->ws_poses);
  release(h->record);
  release(h->stats);
  release(h->tmp_scores);
  release(h->tmp_poses);
  release(h->tmp_noise);
  release(h->piece_stats);
  for (int i = 0; i < NDT2D_PIPELINE_PIECES; ++i)
  {
    if (h->pipe_up[i]) (void)hipEventDestroy(h->pipe_up[i]);
    if (h->pipe_scored[i]) (void)hipEventDestroy(h->pipe_scored[i]);
  }
  if (h->pipe_idle) (void)hipEventDestroy(h->pipe_idle);
  if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
  if (h->down_stream) (void)hipStreamDestroy(h->down_stream);
  release(h->near_list);
  release(h->scan_ranges);
  release(h->scan_points);
  release(h->scan_info);
  for (int i = 0; i < NDT2D_TIMING_HISTORY; ++i)
  {
    if (h->ring0[i]) (void)hipEventDestroy(h->ring0[i]);
    if (h->ring1[i]) (void)hipEventDestroy(h->ring1[i]);
  }
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

const char * ndt2d_last_error(ndt2d_handle h)
{
  return h != nullptr ? h->err.c_str() : "null handle";
}

int ndt2d_set_stream(ndt2d_handle h, void * hip_stream)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->stream = hip_stream != nullptr ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  h->stage_cap = 0;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

void * ndt2d_get_stream(ndt2d_handle h) { return h != nullptr ? h->stream : nullptr; }

int ndt2d_device_id(ndt2d_handle h) { return h != nullptr ? h->device : -1; }

int ndt2d_set_grid(ndt2d_handle h, const double * cells6, uint32_t size_x, uint32_t size_y,
                   double cell_size, double origin_x, double origin_y)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (cells6 == nullptr || size_x == 0 || size_y == 0 || !(cell_size > 0.0))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_grid: bad argument");
  }
  const uint64_t ncell64 = static_cast<uint64_t>(size_x) * size_y;
  if (ncell64 >= (1ull << 31)) return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_grid: grid too large");
  const uint32_t ncell = static_cast<uint32_t>(ncell64);
  NDT2D_HIP(h, hipSetDevice(h->device));
  // The old grid is gone from here on (ensure() may free its buffers): a failure
  // below leaves the context without a grid, never with dangling pointers.
  h->has_grid = false;
  h->bytes_job.n = 0;
  h->stage_cap = 0;   // an open ndt2d_grid_stage_begin is void: this call takes the staging buffer

  // The cells6 records travel once, through pinned staging (the caller's buffer is
  // free on return); the device derives the layouts the scorers read -- packed records
  // h = -0.5 * information with the sentinel for cells that cannot score (n < 5,
  // reference src/ndt_model.cpp:107), their 64-byte-stride copy, the occupancy bitmap
  // and the per-cell map bytes -- in ndt2d_build.hip.  Asynchronous on the stream.
  int rc;
  const size_t n6 = static_cast<size_t>(ncell) * 6;
  // Small maps also get the records of the cells that can score in compacted form, with
  // a cell -> record table (the small-lattice search keeps both in LDS): built here, on
  // the pass over the cells that copies them into the staging buffer anyway.
  uint32_t n_occ = 0;
  const bool compactable = ncell < 65535u;
  if (compactable)
  {
    for (uint32_t i = 0; i < ncell; ++i) n_occ += !(cells6[6 * static_cast<size_t>(i) + 5] < 5.0) ? 1u : 0u;
  }
  const size_t n_compact = compactable ? static_cast<size_t>(n_occ + 1) * kCellDoubles : 0;
  // (uint16 in doubles, whole 16-byte pieces: the kernel copies it with 16-byte loads)
  const size_t n_rank = compactable ? (static_cast<size_t>(ncell) + 1 + 7) / 8 * 2 : 0;
  const size_t n_upload = n6 + n_compact + n_rank;
  if ((rc = ensure(h, h->compact, n_upload)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells_lds_image, static_cast<size_t>(ncell + 1) * kCellDoubles)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells_global, static_cast<size_t>(ncell + 1) * kCellStrideGlobal)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->occ_bits, ((static_cast<size_t>(ncell) + 1 + 31) / 32 + 2) / 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cell_bytes, (static_cast<size_t>(size_x) + 2) * (size_y + 2) / 8 + 1)) != NDT2D_OK) return rc;
  if ((rc = stage_acquire(h, h->stage_grid, n_upload)) != NDT2D_OK) return rc;
  std::memcpy(h->stage_grid.ptr, cells6, n6 * sizeof(double));
  if (compactable)
  {
    double * recs = h->stage_grid.ptr + n6;
    uint16_t * rank = reinterpret_cast<uint16_t *>(recs + n_compact);
    uint32_t k = 0;
    for (uint32_t i = 0; i < ncell; ++i)
    {
      const double * c = cells6 + 6 * static_cast<size_t>(i);
      if (!(c[5] < 5.0))
      {
        double * r = recs + static_cast<size_t>(k) * kCellDoubles;
        r[0] = c[0];
        r[1] = c[1];
        r[2] = -0.5 * c[2];
        r[3] = -0.5 * c[3];
        r[4] = -0.5 * c[4];
        r[5] = 1.0;
        rank[i] = static_cast<uint16_t>(k++);
      }
      else
      {
        rank[i] = static_cast<uint16_t>(n_occ);
      }
    }
    rank[ncell] = static_cast<uint16_t>(n_occ);
    static const double sentinel[kCellDoubles] = {1.0e300, 0.0, -1.0, 0.0, -1.0, 0.0};
    std::memcpy(recs + static_cast<size_t>(n_occ) * kCellDoubles, sentinel, sizeof(sentinel));
  }
  if ((rc = stage_copy(h, h->stage_grid, h->compact.ptr, n_upload)) != NDT2D_OK) return rc;

  GridDesc g{};
  g.size_x = size_x;
  g.size_y = size_y;
  g.ncell = ncell;
  g.cell_size = cell_size;
  g.pow2 = is_pow2(cell_size) ? 1 : 0;
  g.inv_cell_size = 1.0 / cell_size;
  g.origin_x = origin_x;
  g.origin_y = origin_y;
  h->cells6_ptr = h->compact.ptr;
  h->sparse_n = 0;
  h->sparse_index = nullptr;
  h->sparse_cells6 = nullptr;
  hipError_t e = ndt2d::launch_pack_grid(g, h->cells6_ptr, h->cells_lds_image.ptr, h->cells_global.ptr,
                                         reinterpret_cast<uint32_t *>(h->occ_bits.ptr),
                                         reinterpret_cast<uint8_t *>(h->cell_bytes.ptr), h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pack_grid");
  if ((rc = stage_mark(h, h->stage_grid)) != NDT2D_OK) return rc;   // (behind the kernel, see stage_copy)
  g.cells_lds_image = h->cells_lds_image.ptr;
  g.cells_global = h->cells_global.ptr;
  g.occ_bits = reinterpret_cast<const uint32_t *>(h->occ_bits.ptr);
  g.cell_bytes = reinterpret_cast<const uint8_t *>(h->cell_bytes.ptr);
  if (compactable)
  {
    g.compact_records = h->compact.ptr + n6;
    g.cell_rank = reinterpret_cast<const uint16_t *>(h->compact.ptr + n6 + n_compact);
    g.n_occ = n_occ;
  }
  h->grid = g;
  h->coarse_log2 = -1;
  h->block_bytes_log2 = -1;
  h->has_grid = true;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_build_grid(ndt2d_handle h, double ndt_resolution, double range_max,
                     const double * poses_xyt, const double * points_xy, const size_t * offsets,
                     size_t n_scans)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (!(ndt_resolution > 0.0) || n_scans == 0 || poses_xyt == nullptr || offsets == nullptr ||
      n_scans > (1u << 30))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_build_grid: bad argument");
  }
  const size_t n_points = offsets[n_scans];
  if (n_points >= (1ull << 31) || (n_points > 0 && points_xy == nullptr))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_build_grid: bad points");
  }
  // Extent: ScanMatcherNDT::addScans (reference src/scan_matcher_ndt.cpp:52-66); max_x_ /
  // max_y_ start at numeric_limits<double>::min() as the reference has it.
  double min_x = std::numeric_limits<double>::max(), max_x = std::numeric_limits<double>::min();
  double min_y = std::numeric_limits<double>::max(), max_y = std::numeric_limits<double>::min();
  for (size_t k = 0; k < n_scans; ++k)
  {
    min_x = std::min(poses_xyt[3 * k] - range_max, min_x);
    max_x = std::max(poses_xyt[3 * k] + range_max, max_x);
    min_y = std::min(poses_xyt[3 * k + 1] - range_max, min_y);
    max_y = std::max(poses_xyt[3 * k + 1] + range_max, max_y);
  }
  // NDT::NDT (src/ndt_model.cpp:118-126): size_x_ = (size_t)(size_x / cell_size + 1)
  const double fsx = ((max_x - min_x) / ndt_resolution) + 1;
  const double fsy = ((max_y - min_y) / ndt_resolution) + 1;
  if (!(fsx >= 1.0) || !(fsy >= 1.0) || fsx * fsy >= 2147483648.0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_build_grid: degenerate grid extent");
  }
  const uint32_t size_x = static_cast<uint32_t>(static_cast<size_t>(fsx));
  const uint32_t size_y = static_cast<uint32_t>(static_cast<size_t>(fsy));
  const uint32_t ncell = size_x * size_y;
  NDT2D_HIP(h, hipSetDevice(h->device));
  // as in ndt2d_set_grid: no grid until the new one is complete
  h->has_grid = false;
  h->bytes_job.n = 0;
  h->stage_cap = 0;

  // staging that must stay alive until the copies are done: kept in the context
  NDT2D_SYNC(h);
  h->stage_scans.resize(4 * n_scans);
  h->stage_offsets.resize(((n_scans + 1) + 1) & ~size_t(1));
  for (size_t k = 0; k < n_scans; ++k)
  {
    h->stage_scans[4 * k] = poses_xyt[3 * k];
    h->stage_scans[4 * k + 1] = poses_xyt[3 * k + 1];
    ndt2d_cos_sin(poses_xyt[3 * k + 2], &h->stage_scans[4 * k + 2], &h->stage_scans[4 * k + 3]);  // :135-136, host libm
    h->stage_offsets[k] = static_cast<uint32_t>(offsets[k]);
  }
  h->stage_offsets[n_scans] = static_cast<uint32_t>(n_points);

  const size_t temp_bytes = ndt2d::build_sort_temp_bytes(static_cast<uint32_t>(n_points), ncell);
  int rc;
  if ((rc = ensure(h, h->b_points, 2 * n_points + 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_scans, 4 * n_scans)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_offsets, h->stage_offsets.size() / 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_world, 2 * n_points + 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_keys, n_points + 2)) != NDT2D_OK) return rc;   // 2 x u32 arrays
  if ((rc = ensure(h, h->b_vals, n_points + 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_temp, temp_bytes / 8 + 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_seg, static_cast<size_t>(ncell) + 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells6, static_cast<size_t>(ncell) * 6)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells_lds_image, static_cast<size_t>(ncell + 1) * kCellDoubles)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells_global, static_cast<size_t>(ncell + 1) * kCellStrideGlobal)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->occ_bits, ((static_cast<size_t>(ncell) + 1 + 31) / 32 + 2) / 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cell_bytes, (static_cast<size_t>(size_x) + 2) * (size_y + 2) / 8 + 1)) != NDT2D_OK) return rc;

  if (n_points > 0)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h->b_points.ptr, points_xy, 2 * n_points * sizeof(double),
                                hipMemcpyHostToDevice, h->stream));
  }
  NDT2D_HIP(h, hipMemcpyAsync(h->b_scans.ptr, h->stage_scans.data(), 4 * n_scans * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
  NDT2D_HIP(h, hipMemcpyAsync(h->b_offsets.ptr, h->stage_offsets.data(),
                              (n_scans + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));

  GridDesc g{};
  g.size_x = size_x;
  g.size_y = size_y;
  g.ncell = ncell;
  g.cell_size = ndt_resolution;
  g.pow2 = is_pow2(ndt_resolution) ? 1 : 0;
  g.inv_cell_size = 1.0 / ndt_resolution;
  g.origin_x = min_x;
  g.origin_y = min_y;

  ndt2d::BuildArgs a{};
  a.eigen_form = h->eigen_form;
  a.grid = g;
  a.points_xy = h->b_points.ptr;
  a.n_points = static_cast<uint32_t>(n_points);
  a.scans = h->b_scans.ptr;
  a.offsets = reinterpret_cast<const uint32_t *>(h->b_offsets.ptr);
  a.n_scans = static_cast<uint32_t>(n_scans);
  a.world_xy = h->b_world.ptr;
  a.keys_in = reinterpret_cast<uint32_t *>(h->b_keys.ptr);
  a.keys_out = a.keys_in + n_points + 1;
  a.vals_in = reinterpret_cast<uint32_t *>(h->b_vals.ptr);
  a.vals_out = a.vals_in + n_points + 1;
  a.sort_temp = h->b_temp.ptr;
  a.sort_temp_bytes = temp_bytes;
  a.seg_begin = reinterpret_cast<uint32_t *>(h->b_seg.ptr);
  a.cells6 = h->cells6.ptr;
  a.cells_lds_image = h->cells_lds_image.ptr;
  a.cells_global = h->cells_global.ptr;
  a.occ_bits = reinterpret_cast<uint32_t *>(h->occ_bits.ptr);
  a.cell_bytes = reinterpret_cast<uint8_t *>(h->cell_bytes.ptr);
  if (int trc = next_timing_slot(h); trc != NDT2D_OK) return trc;
  NDT2D_HIP(h, hipEventRecord(h->ev0, h->stream));
  hipError_t e = ndt2d::launch_build_grid(a, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_build_grid");
  NDT2D_HIP(h, hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  h->last_kernels = 7;
  h->last_variant = "build/sort-by-cell+in-order-recurrence";

  g.cells_lds_image = h->cells_lds_image.ptr;
  g.cells_global = h->cells_global.ptr;
  g.occ_bits = reinterpret_cast<const uint32_t *>(h->occ_bits.ptr);
  g.cell_bytes = reinterpret_cast<const uint8_t *>(h->cell_bytes.ptr);
  h->cells6_ptr = h->cells6.ptr;
  h->sparse_n = 0;
  h->sparse_index = nullptr;
  h->sparse_cells6 = nullptr;
  if (ncell < 65535u)
  {
    // Small grids also get the compacted records + cell -> record table the searches keep in LDS
    // (as a host-installed grid does, ndt2d_set_grid): a loop closure's map is built here, and its
    // search was 8-20 % slower without them.  The count of scoring cells comes back to the host (the
    // kernels' LDS images are sized by it): one small copy behind the build.
    const size_t n_rec = static_cast<size_t>(ncell + 1) * kCellDoubles;
    const size_t n_rank = (static_cast<size_t>(ncell) + 1 + 7) / 8 * 2;
    if ((rc = ensure(h, h->compact, n_rec + n_rank + 2)) != NDT2D_OK) return rc;
    double * records = h->compact.ptr;
    uint16_t * ranks = reinterpret_cast<uint16_t *>(h->compact.ptr + n_rec);
    uint32_t * d_n_occ = reinterpret_cast<uint32_t *>(h->compact.ptr + n_rec + n_rank);
    e = ndt2d::launch_compact_grid(ncell, h->cells_lds_image.ptr, reinterpret_cast<const uint32_t *>(h->occ_bits.ptr),
                                   records, ranks, d_n_occ, h->stream);
    if (e != hipSuccess) return fail_hip(h, e, "launch_compact_grid");
    uint32_t n_occ = 0;
    NDT2D_HIP(h, hipMemcpyAsync(&n_occ, d_n_occ, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    NDT2D_SYNC(h);
    if (n_occ > 0)
    {
      g.compact_records = records;
      g.cell_rank = ranks;
      g.n_occ = n_occ;
    }
  }
  h->grid = g;
  h->coarse_log2 = -1;
  h->block_bytes_log2 = -1;
  h->has_grid = true;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

// Largest [rank table | compacted records] image any search kernel would keep in LDS
// (ndt2d_match_small.hip: 72 KB together with its map and rows; ndt2d_match_lane.hip: half a
// CU's LDS minus its map): beyond it the compacted form is not prepared.
static constexpr size_t kCompactImageBudget = 80 * 1024;

// Layout of the staged image of a list install for a capacity of `cap` listed cells (doubles):
// [cells6: 6 cap][cell indices: cap u32][ranks of the listed cells: cap u16][compact records:
// (cap + 1) x 6][occupancy words of the grid]
struct StageLayout
{
  size_t n6, n_idx, n_rk, n_compact_max, off_occ, n_words, n_upload;
};
static StageLayout stage_layout(size_t cap, uint32_t ncell, bool with_compact)
{
  StageLayout l{};
  l.n6 = cap * 6;
  l.n_idx = (cap + 1) / 2;
  l.n_rk = (cap + 3) / 4;
  if ((l.n6 + l.n_idx + l.n_rk) & 1) ++l.n_rk;   // the compact records are read with 16-byte loads
  l.n_compact_max = with_compact ? (cap + 1) * kCellDoubles : 0;
  // (occupancy words of the cells that can score: they tell the install kernel which cells
  // the list will write, ndt2d_build.hip)
  l.n_words = (static_cast<size_t>(ncell) + 1 + 31) / 32;
  l.off_occ = l.n6 + l.n_idx + l.n_rk + l.n_compact_max;
  l.n_upload = l.off_occ + (l.n_words + 1) / 2 + 2;
  return l;
}

int ndt2d_grid_stage_begin(ndt2d_handle h, uint32_t size_x, uint32_t size_y, size_t capacity,
                           uint32_t ** cell_index_out, double ** cells6_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  h->stage_cap = 0;
  if (cell_index_out == nullptr || cells6_out == nullptr || size_x == 0 || size_y == 0 || capacity >= (1ull << 31))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_grid_stage_begin: bad argument");
  }
  const uint64_t ncell64 = static_cast<uint64_t>(size_x) * size_y;
  if (ncell64 >= (1ull << 31)) return fail(h, NDT2D_ERR_INVALID, "ndt2d_grid_stage_begin: grid too large");
  const uint32_t ncell = static_cast<uint32_t>(ncell64);
  NDT2D_HIP(h, hipSetDevice(h->device));
  h->has_grid = false;   // (see ndt2d_set_grid)
  h->bytes_job.n = 0;
  const size_t cap = capacity > 0 ? capacity : 1;
  // the compacted records are prepared only when their image could fit a search kernel's LDS
  const size_t rank_table_bytes = (static_cast<size_t>(ncell) + 1 + 7) / 8 * 16;
  const bool may_compact = rank_table_bytes + 2 * kCellDoubles * sizeof(double) <= kCompactImageBudget;
  const StageLayout l = stage_layout(cap, ncell, may_compact);
  int rc;
  if ((rc = ensure(h, h->compact, l.n_upload)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells_lds_image, static_cast<size_t>(ncell + 1) * kCellDoubles)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cells_global, static_cast<size_t>(ncell + 1) * kCellStrideGlobal)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->occ_bits, ((static_cast<size_t>(ncell) + 1 + 31) / 32 + 2) / 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->cell_bytes, (static_cast<size_t>(size_x) + 2) * (size_y + 2) / 8 + 1)) != NDT2D_OK) return rc;
  if (may_compact && (rc = ensure(h, h->ranks, rank_table_bytes / sizeof(double) + 2)) != NDT2D_OK) return rc;
  if ((rc = stage_acquire(h, h->stage_grid, l.n_upload)) != NDT2D_OK) return rc;
  h->stage_cap = cap;
  h->stage_sx = size_x;
  h->stage_sy = size_y;
  h->stage_may_compact = may_compact;
  *cells6_out = h->stage_grid.ptr;
  *cell_index_out = reinterpret_cast<uint32_t *>(h->stage_grid.ptr + l.n6);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_grid_stage_commit(ndt2d_handle h, size_t n_listed, double cell_size, double origin_x, double origin_y)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (h->stage_cap == 0) return fail(h, NDT2D_ERR_STATE, "ndt2d_grid_stage_commit: no ndt2d_grid_stage_begin before it");
  const size_t cap = h->stage_cap;
  h->stage_cap = 0;
  if (n_listed > cap || !(cell_size > 0.0)) return fail(h, NDT2D_ERR_INVALID, "ndt2d_grid_stage_commit: bad argument");
  const uint32_t size_x = h->stage_sx, size_y = h->stage_sy;
  const uint32_t ncell = size_x * size_y;
  const uint32_t n = static_cast<uint32_t>(n_listed);
  const StageLayout l = stage_layout(cap, ncell, h->stage_may_compact);
  const size_t n6 = l.n6, n_idx = l.n_idx, n_rk = l.n_rk, off_occ = l.off_occ, n_words = l.n_words, n_upload = l.n_upload;
  double * st = h->stage_grid.ptr;
  const double * cells6 = st;
  uint32_t * st_idx = reinterpret_cast<uint32_t *>(st + n6);
  const uint32_t * cell_index = st_idx;
  for (uint32_t k = 0; k < n; ++k)
  {
    if (cell_index[k] >= ncell) return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_grid_sparse: cell index out of range");
  }
  uint32_t n_occ = 0;
  for (uint32_t k = 0; k < n; ++k) n_occ += !(cells6[6 * static_cast<size_t>(k) + 5] < 5.0) ? 1u : 0u;
  const size_t rank_table_bytes = (static_cast<size_t>(ncell) + 1 + 7) / 8 * 16;
  const bool compactable = h->stage_may_compact && n_occ > 0 && n_occ < 65535u &&
                           rank_table_bytes + (static_cast<size_t>(n_occ) + 1) * kCellDoubles * sizeof(double) <=
                             kCompactImageBudget;
  const size_t n_compact = compactable ? static_cast<size_t>(n_occ + 1) * kCellDoubles : 0;
  int rc;
  uint16_t * st_rk = reinterpret_cast<uint16_t *>(st + n6 + n_idx);
  double * st_rec = st + n6 + n_idx + n_rk;
  uint32_t * st_occ = reinterpret_cast<uint32_t *>(st + off_occ);
  std::memset(st_occ, 0, (n_words + 1) / 2 * sizeof(double));
  uint32_t k_occ = 0;
  // every cell at most once, whether it can score or not (two records for one cell: whichever
  // thread stored last would win, in the install kernel or in ndt2d_get_grid's scatter)
  h->stage_seen.assign(n_words, 0u);
  for (uint32_t k = 0; k < n; ++k)
  {
    const double * c = cells6 + 6 * static_cast<size_t>(k);
    {
      const uint32_t bit = 1u << (cell_index[k] & 31u);
      if (h->stage_seen[cell_index[k] >> 5] & bit)
      {
        return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_grid_sparse: a cell is listed twice");
      }
      h->stage_seen[cell_index[k] >> 5] |= bit;
    }
    if (!(c[5] < 5.0))
    {
      const uint32_t bit = 1u << (cell_index[k] & 31u);
      st_occ[cell_index[k] >> 5] |= bit;
      if (compactable)
      {
        double * r = st_rec + static_cast<size_t>(k_occ) * kCellDoubles;
        r[0] = c[0];
        r[1] = c[1];
        r[2] = -0.5 * c[2];
        r[3] = -0.5 * c[3];
        r[4] = -0.5 * c[4];
        r[5] = 1.0;
      }
      st_rk[k] = static_cast<uint16_t>(k_occ++);
    }
    else
    {
      st_rk[k] = static_cast<uint16_t>(n_occ);
    }
  }
  if (compactable)
  {
    static const double sentinel[kCellDoubles] = {1.0e300, 0.0, -1.0, 0.0, -1.0, 0.0};
    std::memcpy(st_rec + static_cast<size_t>(n_occ) * kCellDoubles, sentinel, sizeof(sentinel));
  }
  // the install kernel reads the staged image in place and leaves the device copy behind; a
  // staging buffer the device cannot address is copied first
  ndt2d::SparseImage image{};
  image.n = n;
  image.off_idx = n6;
  image.off_rk = n6 + n_idx;
  image.off_compact = n6 + n_idx + n_rk;
  image.n_compact = n_compact;
  image.off_occ = off_occ;
  if (h->stage_grid.dev != nullptr)
  {
    image.src = h->stage_grid.dev;
    image.dst = h->compact.ptr;
  }
  else
  {
    if ((rc = stage_copy(h, h->stage_grid, h->compact.ptr, n_upload)) != NDT2D_OK) return rc;
    image.src = h->compact.ptr;
    image.dst = nullptr;
  }

  GridDesc g{};
  g.size_x = size_x;
  g.size_y = size_y;
  g.ncell = ncell;
  g.cell_size = cell_size;
  g.pow2 = is_pow2(cell_size) ? 1 : 0;
  g.inv_cell_size = 1.0 / cell_size;
  g.origin_x = origin_x;
  g.origin_y = origin_y;
  const double * d_cells6 = h->compact.ptr;
  const uint32_t * d_idx = reinterpret_cast<const uint32_t *>(h->compact.ptr + n6);
  hipError_t e = ndt2d::launch_grid_install(
    g, image, h->cells_lds_image.ptr, h->cells_global.ptr, reinterpret_cast<uint32_t *>(h->occ_bits.ptr),
    reinterpret_cast<uint8_t *>(h->cell_bytes.ptr),
    compactable ? reinterpret_cast<uint16_t *>(h->ranks.ptr) : nullptr, n_occ, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_grid_install");
  // The map bytes around the listed cells (they read the records the install has just written):
  // queued right behind it.  Until round 6 the job waited for the next few-pose launch to carry
  // it -- the mapper's scoreScan -- or ran ahead of the next search; since a short scan's single
  // pose is scored on the host (round 4) it always ended up in front of the search, on the
  // critical path of the mapper's cycle.  Here it runs while the host prepares that search.
  g.cells_lds_image = h->cells_lds_image.ptr;
  g.cells_global = h->cells_global.ptr;
  g.occ_bits = reinterpret_cast<const uint32_t *>(h->occ_bits.ptr);
  g.cell_bytes = reinterpret_cast<const uint8_t *>(h->cell_bytes.ptr);
  {
    ndt2d::SparseBytesJob job{};
    job.cell_index = d_idx;
    job.cells6 = d_cells6;
    job.n = n;
    job.bytes = reinterpret_cast<uint8_t *>(h->cell_bytes.ptr);
    e = ndt2d::launch_sparse_bytes(g, job, h->stream);
    if (e != hipSuccess) return fail_hip(h, e, "launch_sparse_bytes");
    h->bytes_job.n = 0;
  }
  // (behind the install kernel, which reads the buffer in place)
  if ((rc = stage_mark(h, h->stage_grid)) != NDT2D_OK) return rc;
  if (compactable)
  {
    g.compact_records = h->compact.ptr + n6 + n_idx + n_rk;
    g.cell_rank = reinterpret_cast<const uint16_t *>(h->ranks.ptr);
    g.n_occ = n_occ;
  }
  h->cells6_ptr = nullptr;          // made from the list when ndt2d_get_grid asks for them
  h->sparse_n = n;
  h->sparse_index = d_idx;
  h->sparse_cells6 = d_cells6;
  h->grid = g;
  h->coarse_log2 = -1;
  h->block_bytes_log2 = -1;
  h->has_grid = true;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_set_grid_sparse(ndt2d_handle h, const uint32_t * cell_index, const double * cells6,
                          size_t n_listed, uint32_t size_x, uint32_t size_y, double cell_size,
                          double origin_x, double origin_y)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if ((n_listed > 0 && (cell_index == nullptr || cells6 == nullptr)) || size_x == 0 || size_y == 0 ||
      !(cell_size > 0.0) || n_listed >= (1ull << 31))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_grid_sparse: bad argument");
  }
  uint32_t * st_idx = nullptr;
  double * st_cells6 = nullptr;
  const int rc = ndt2d_grid_stage_begin(h, size_x, size_y, n_listed, &st_idx, &st_cells6);
  if (rc != NDT2D_OK) return rc;
  if (n_listed > 0)
  {
    std::memcpy(st_cells6, cells6, n_listed * 6 * sizeof(double));
    std::memcpy(st_idx, cell_index, n_listed * sizeof(uint32_t));
  }
  return ndt2d_grid_stage_commit(h, n_listed, cell_size, origin_x, origin_y);
  NDT2D_C_CATCH(h)
}

int ndt2d_set_eigenvalue_form(ndt2d_handle h, const char * form)
{
  NDT2D_C_TRY
  if (h == nullptr || form == nullptr) return NDT2D_ERR_INVALID;
  if (std::strcmp(form, "eigen") == 0) h->eigen_form = 0;
  else if (std::strcmp(form, "closed") == 0) h->eigen_form = 1;
  else return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_eigenvalue_form: unknown form (eigen, closed)");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_get_grid(ndt2d_handle h, double * cells6_out, size_t capacity_cells, uint32_t * size_x,
                   uint32_t * size_y, double * cell_size, double * origin_x, double * origin_y)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_get_grid: no grid");
  if (size_x) *size_x = h->grid.size_x;
  if (size_y) *size_y = h->grid.size_y;
  if (cell_size) *cell_size = h->grid.cell_size;
  if (origin_x) *origin_x = h->grid.origin_x;
  if (origin_y) *origin_y = h->grid.origin_y;
  if (cells6_out != nullptr)
  {
    if (capacity_cells < h->grid.ncell) return fail(h, NDT2D_ERR_INVALID, "ndt2d_get_grid: capacity");
    NDT2D_HIP(h, hipSetDevice(h->device));
    if (h->cells6_ptr == nullptr)
    {
      // a grid installed as a list: the dense records from the list, once
      int rc = ensure(h, h->cells6, static_cast<size_t>(h->grid.ncell) * 6);
      if (rc != NDT2D_OK) return rc;
      hipError_t e = ndt2d::launch_grid_sparse_to_dense(h->sparse_index, h->sparse_cells6, h->sparse_n,
                                                        h->grid.ncell, h->cells6.ptr, h->stream);
      if (e != hipSuccess) return fail_hip(h, e, "launch_grid_sparse_to_dense");
      h->cells6_ptr = h->cells6.ptr;
    }
    NDT2D_HIP(h, hipMemcpyAsync(cells6_out, h->cells6_ptr,
                                static_cast<size_t>(h->grid.ncell) * 6 * sizeof(double),
                                hipMemcpyDeviceToHost, h->stream));
    NDT2D_SYNC(h);
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_clear_grid(ndt2d_handle h)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  h->has_grid = false;
  h->bytes_job.n = 0;
  h->stage_cap = 0;   // (an open list is dropped with the grid it was for)
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_has_grid(ndt2d_handle h) { return (h != nullptr && h->has_grid) ? 1 : 0; }

int ndt2d_set_beams(ndt2d_handle h, const double * beams_xy, size_t n_beams)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (beams_xy == nullptr || n_beams == 0 || n_beams > (1u << 20))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_beams: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->beams, 2 * n_beams + 2);
  if (rc != NDT2D_OK) return rc;
  rc = stage_acquire(h, h->stage_beams, 2 * n_beams);
  if (rc != NDT2D_OK) return rc;
  std::memcpy(h->stage_beams.ptr, beams_xy, 2 * n_beams * sizeof(double));
  rc = stage_submit(h, h->stage_beams, h->beams.ptr, 2 * n_beams);
  if (rc != NDT2D_OK) return rc;
  h->n_beams = n_beams;
  h->beams_ptr = h->beams.ptr;
  // a search prepared for other beams must be set up again (ndt2d_set_search) before
  // ndt2d_match_launch: its caller pairs the tables with THESE beams
  h->has_search = false;
  h->beam_rmax = beam_reach(beams_xy, n_beams);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_set_search(ndt2d_handle h, double pose_x, double pose_y, const double * dth,
                     const double * cos_th, const double * sin_th, size_t n_th,
                     const double * dlin, size_t n_lin)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (dth == nullptr || cos_th == nullptr || sin_th == nullptr || dlin == nullptr || n_th == 0 ||
      n_lin == 0 || n_th > (1u << 24) || n_lin > 46340)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_search: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  const size_t n_tab = 3 * n_th + n_lin;
  int rc = ensure(h, h->tables, n_tab);
  if (rc != NDT2D_OK) return rc;
  rc = stage_acquire(h, h->stage_tables, n_tab);
  if (rc != NDT2D_OK) return rc;
  double * t = h->stage_tables.ptr;
  std::memcpy(t, dth, n_th * sizeof(double));
  std::memcpy(t + n_th, cos_th, n_th * sizeof(double));
  std::memcpy(t + 2 * n_th, sin_th, n_th * sizeof(double));
  std::memcpy(t + 3 * n_th, dlin, n_lin * sizeof(double));
  h->tables_uploaded = false;   // ndt2d_match_launch uploads them if its kernel reads HBM
  h->n_th = n_th;
  h->n_lin = n_lin;
  lattice_extent(h, dlin, n_lin);
  h->pose_x = pose_x;
  h->pose_y = pose_y;
  h->tables_ptr = h->tables.ptr;
  h->has_search = true;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_set_search_beams(ndt2d_handle h, const double * beams_xy, size_t n_beams, double pose_x,
                           double pose_y, const double * dth, const double * cos_th,
                           const double * sin_th, size_t n_th, const double * dlin, size_t n_lin)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (beams_xy == nullptr)
  {
    // the beams the context holds stay: only the tables change (and need no copy yet)
    if (n_beams == 0 || n_beams != h->n_beams || h->beams_ptr == nullptr)
    {
      return fail(h, NDT2D_ERR_STATE, "ndt2d_set_search_beams: no such beams on the device");
    }
    return ndt2d_set_search(h, pose_x, pose_y, dth, cos_th, sin_th, n_th, dlin, n_lin);
  }
  if (n_beams == 0 || n_beams > (1u << 20) || dth == nullptr ||
      cos_th == nullptr || sin_th == nullptr || dlin == nullptr || n_th == 0 || n_lin == 0 ||
      n_th > (1u << 24) || n_lin > 46340)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_search_beams: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  // one staging buffer, one copy: [beams 2n][dth | cos | sin  3 n_th][dlin n_lin]
  const size_t n_tab = 3 * n_th + n_lin;
  const size_t n_all = 2 * n_beams + n_tab;
  int rc = ensure(h, h->call_dev, n_all);
  if (rc != NDT2D_OK) return rc;
  rc = stage_acquire(h, h->stage_call, n_all);
  if (rc != NDT2D_OK) return rc;
  double * t = h->stage_call.ptr;
  std::memcpy(t, beams_xy, 2 * n_beams * sizeof(double));
  t += 2 * n_beams;
  std::memcpy(t, dth, n_th * sizeof(double));
  std::memcpy(t + n_th, cos_th, n_th * sizeof(double));
  std::memcpy(t + 2 * n_th, sin_th, n_th * sizeof(double));
  std::memcpy(t + 3 * n_th, dlin, n_lin * sizeof(double));
  rc = stage_submit(h, h->stage_call, h->call_dev.ptr, n_all);
  if (rc != NDT2D_OK) return rc;
  h->n_beams = n_beams;
  h->beam_rmax = beam_reach(beams_xy, n_beams);
  h->beams_ptr = h->call_dev.ptr;
  h->tables_ptr = h->call_dev.ptr + 2 * n_beams;
  h->tables_uploaded = true;
  h->n_th = n_th;
  h->n_lin = n_lin;
  lattice_extent(h, dlin, n_lin);
  h->pose_x = pose_x;
  h->pose_y = pose_y;
  h->has_search = true;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_match_launch(ndt2d_handle h, size_t th_begin, size_t th_end, double * d_scores,
                       double * d_record)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (th_begin >= th_end) return fail(h, NDT2D_ERR_INVALID, "ndt2d_match_launch: bad theta range");
  return ndt2d_match_launch_strided(h, th_begin, 1, th_end - th_begin, d_scores, d_record);
  NDT2D_C_CATCH(h)
}

int ndt2d_match_launch_strided(ndt2d_handle h, size_t th_first, size_t th_stride, size_t th_count,
                               double * d_scores, double * d_record)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_match_launch: no grid");
  if (!h->has_search || h->n_beams == 0)
  {
    return fail(h, NDT2D_ERR_STATE, "ndt2d_match_launch: set_beams/set_search first");
  }
  if (th_count == 0 || th_stride == 0 || th_first >= h->n_th ||
      (th_count - 1) > (h->n_th - 1 - th_first) / th_stride)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_match_launch: bad theta range");
  }
  const size_t th_begin = th_first, th_end = th_first + th_count;
  NDT2D_TEST_MAYBE_FAIL(h, "ndt2d_match_launch");
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->record, NDT2D_MATCH_RECORD_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  if (h->bytes_job.n > 0)
  {
    // the map bytes of a list install that no scoreScan has carried yet
    hipError_t be = ndt2d::launch_sparse_bytes(h->grid, h->bytes_job, h->stream);
    if (be != hipSuccess) return fail_hip(h, be, "launch_sparse_bytes");
    h->bytes_job.n = 0;
  }

  ndt2d::MatchArgs a{};
  a.grid = h->grid;
  a.beams_xy = h->beams_ptr;
  a.n_beams = static_cast<uint32_t>(h->n_beams);
  a.dth = h->tables_ptr;
  a.cos_th = h->tables_ptr + h->n_th;
  a.sin_th = h->tables_ptr + 2 * h->n_th;
  a.dlin = h->tables_ptr + 3 * h->n_th;
  a.dlin_absmax = h->dlin_absmax;
  a.patch_span = h->patch_span;
  a.beam_rmax = h->beam_rmax;
  a.n_th = static_cast<uint32_t>(h->n_th);
  a.n_lin = static_cast<uint32_t>(h->n_lin);
  a.th_begin = static_cast<uint32_t>(th_begin);
  a.th_end = static_cast<uint32_t>(th_end);
  a.th_stride = static_cast<uint32_t>(th_stride);
  a.pose_x = h->pose_x;
  a.pose_y = h->pose_y;
  a.scores = d_scores;
  if (!(h->force_variant & (ndt2d::kVariantWave | ndt2d::kVariantLane)))
  {
    // a small lattice on a window wider than 256 cells: the small-lattice search wants the
    // grid's map bytes per block of cells (made once per grid and block size)
    const int k = ndt2d::match_small_block_log2(a, ndt2d::poses_lds_per_block());
    if (k > 0)
    {
      if (h->block_bytes_log2 != k)
      {
        rc = ensure(h, h->block_bytes, ndt2d::grid_block_bytes_size(h->grid, static_cast<uint32_t>(k)) / 8 + 2);
        if (rc != NDT2D_OK) return rc;
        hipError_t be = ndt2d::grid_block_bytes_launch(h->grid, static_cast<uint32_t>(k),
                                                       reinterpret_cast<uint8_t *>(h->block_bytes.ptr), h->stream);
        if (be != hipSuccess) return fail_hip(h, be, "grid_block_bytes_launch");
        h->block_bytes_log2 = k;
      }
      a.grid.block_bytes = reinterpret_cast<const uint8_t *>(h->block_bytes.ptr);
      a.grid.block_bytes_log2 = static_cast<uint32_t>(k);
    }
  }
  if (!h->tables_uploaded)
  {
    a.host_tables = h->stage_tables.ptr;
    if (ndt2d::match_needs_device_tables(a, !(h->force_variant & ndt2d::kVariantWave), h->force_variant))
    {
      rc = stage_submit(h, h->stage_tables, h->tables.ptr, 3 * h->n_th + h->n_lin);
      if (rc != NDT2D_OK) return rc;
      h->tables_uploaded = true;
      a.host_tables = nullptr;
    }
  }
  {
    // the head of the workspace holds counters the kernels expect to find at zero (they
    // leave them so): a fresh allocation is cleared once
    const size_t cap_before = h->ws_match.cap;
    rc = ensure(h, h->ws_match, ndt2d::match_workspace_doubles(a));
    if (rc != NDT2D_OK) return rc;
    // (a fresh allocation is cleared once: the counters at its head, and the `done` words of the
    // small-lattice search behind its record table)
    if (h->ws_match.cap != cap_before)
    {
      const size_t head_and_small = (ndt2d::match_workspace_head_doubles() +
                                     ndt2d::kSmallMaxItems * (NDT2D_MATCH_RECORD_DOUBLES + 1)) * sizeof(double);
      const size_t all = h->ws_match.cap * sizeof(double);
      NDT2D_HIP(h, hipMemsetAsync(h->ws_match.ptr, 0, head_and_small < all ? head_and_small : all, h->stream));
    }
  }

  // scratch for the rotated-beam table of the lane-per-candidate mapping
  double * outer = nullptr;
  if (!(h->force_variant & ndt2d::kVariantWave))
  {
    rc = ensure(h, h->outer, ndt2d::match_lane_outer_doubles(a));
    if (rc != NDT2D_OK) return rc;
    outer = h->outer.ptr;
  }

  rc = ensure_host_res(h);
  if (rc != NDT2D_OK) return rc;
  ndt2d::LaunchInfo info{"", 0};
  if (h->timing)
  {
    if (int trc = next_timing_slot(h); trc != NDT2D_OK) return trc;
  }
  h->match_seq = ++h->seq;
  h->match_pos = h->queued;   // (every mark so far is behind work queued before this search)
  // (the event pair brackets the search kernel itself)
  hipError_t e = ndt2d::launch_match(a, h->ws_match.ptr, outer, h->record.ptr, d_record,
                                     h->host_res_dev, h->match_seq, h->force_variant, h->stream,
                                     h->timing ? h->ev0 : nullptr, h->timing ? h->ev1 : nullptr, &info);
  if (e != hipSuccess) return fail_hip(h, e, "launch_match");
  h->timed = h->timing;
  h->last_kernels = info.n_kernels;
  h->last_variant = info.variant;
  h->match_pending = true;
  ++h->match_launches;
  h->last_candidates = static_cast<uint64_t>(th_end - th_begin) * h->n_lin * h->n_lin;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_match_fetch(ndt2d_handle h, ndt2d_match_result * out)
{
  NDT2D_C_TRY
  if (h == nullptr || out == nullptr) return NDT2D_ERR_INVALID;
  if (!h->match_pending) return fail(h, NDT2D_ERR_STATE, "ndt2d_match_fetch: nothing launched");
  NDT2D_HIP(h, hipSetDevice(h->device));
  // the final reduction wrote the record into host-coherent memory, then its flag
  // (whatever the wait says, the search is no longer pending: a call that failed leaves the
  // context ready for the next one)
  h->match_pending = false;
  ++h->match_fetches;
  int rc = wait_host_flag(h, ndt2d::kHostFlagSlot, h->match_seq, h->match_pos);
  if (rc != NDT2D_OK) return rc;
  double rec[NDT2D_MATCH_RECORD_DOUBLES];
  for (int k = 0; k < NDT2D_MATCH_RECORD_DOUBLES; ++k) rec[k] = h->host_res[k];
  out->best_score = rec[0];
  out->best_index = rec[1] < 0.0 ? NDT2D_NO_INDEX : static_cast<uint64_t>(rec[1]);
  // (index + 0.5: another candidate lies within the near-tie tolerance of the winner, ndt2d_device_fn.h merge_best)
  out->near_tie = (rec[1] >= 0.0 && rec[1] != std::floor(rec[1])) ? 1u : 0u;
  for (int k = 0; k < 10; ++k) out->acc[k] = rec[2 + k];
  out->n_candidates = h->last_candidates;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_match_near_best(ndt2d_handle h, size_t th_begin, size_t th_end, double rel, uint64_t * index_out,
                          size_t capacity, size_t * n_out, ndt2d_match_result * result_out)
{
  NDT2D_C_TRY
  if (h == nullptr || n_out == nullptr || (capacity > 0 && index_out == nullptr)) return NDT2D_ERR_INVALID;
  *n_out = 0;
  if (!(rel >= 0.0) || th_begin >= th_end || th_end > h->n_th || capacity > 65536)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_match_near_best: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  const size_t n_scores = (th_end - th_begin) * h->n_lin * h->n_lin;
  int rc = ensure(h, h->tmp_scores, n_scores);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->near_list, capacity + 2);
  if (rc != NDT2D_OK) return rc;
  rc = ndt2d_match_launch(h, th_begin, th_end, h->tmp_scores.ptr, nullptr);
  if (rc != NDT2D_OK) return rc;
  unsigned long long * list = reinterpret_cast<unsigned long long *>(h->near_list.ptr);
  std::vector<unsigned long long> host(capacity + 1);
  // one pass over all scores; should more candidates qualify than the list holds, further passes
  // over a shrinking index range until the list holds exactly the FIRST ones in visiting order
  // (bisection on the upper index; a plateau of equal scores is the only realistic way there)
  auto collect = [&](uint64_t hi) -> int {
    hipError_t e = ndt2d::launch_collect_near(h->tmp_scores.ptr, n_scores, hi, h->record.ptr, rel, 0.0, list,
                                              static_cast<uint32_t>(capacity), h->stream);
    if (e != hipSuccess) return fail_hip(h, e, "launch_collect_near");
    NDT2D_HIP(h, hipMemcpyAsync(host.data(), list, (capacity + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                h->stream));
    NDT2D_SYNC(h);
    return NDT2D_OK;
  };
  if ((rc = collect(n_scores)) != NDT2D_OK) return rc;
  const size_t total = static_cast<size_t>(host[0]);
  if (total > capacity && capacity > 0)
  {
    uint64_t lo = 0, hi = n_scores;   // count(lo) <= capacity < count(hi)
    while (hi - lo > 1)
    {
      const uint64_t mid = lo + (hi - lo) / 2;
      if ((rc = collect(mid)) != NDT2D_OK) return rc;
      if (host[0] <= capacity) lo = mid; else hi = mid;
    }
    if ((rc = collect(lo)) != NDT2D_OK) return rc;
  }
  ndt2d_match_result res;
  rc = ndt2d_match_fetch(h, &res);
  if (rc != NDT2D_OK) return rc;
  if (result_out != nullptr) *result_out = res;
  *n_out = total;
  const size_t kept = std::min<size_t>(static_cast<size_t>(host[0]), capacity);
  const uint64_t base = static_cast<uint64_t>(th_begin) * h->n_lin * h->n_lin;
  for (size_t k = 0; k < kept; ++k) index_out[k] = base + host[1 + k];
  std::sort(index_out, index_out + kept);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_match_status(ndt2d_handle h, uint64_t * n_launched, uint64_t * n_fetched)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (n_launched != nullptr) *n_launched = h->match_launches;
  if (n_fetched != nullptr) *n_fetched = h->match_fetches;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_match(ndt2d_handle h, size_t th_begin, size_t th_end, double * h_scores,
                ndt2d_match_result * out)
{
  NDT2D_C_TRY
  if (h == nullptr || out == nullptr) return NDT2D_ERR_INVALID;
  double * d_scores = nullptr;
  size_t n_scores = 0;
  if (h_scores != nullptr)
  {
    if (th_begin >= th_end || th_end > h->n_th)
    {
      return fail(h, NDT2D_ERR_INVALID, "ndt2d_match: bad theta range");
    }
    NDT2D_HIP(h, hipSetDevice(h->device));
    n_scores = (th_end - th_begin) * h->n_lin * h->n_lin;
    int rc = ensure(h, h->tmp_scores, n_scores);
    if (rc != NDT2D_OK) return rc;
    d_scores = h->tmp_scores.ptr;
  }
  int rc = ndt2d_match_launch(h, th_begin, th_end, d_scores, nullptr);
  if (rc != NDT2D_OK) return rc;
  if (h_scores != nullptr)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h_scores, d_scores, n_scores * sizeof(double),
                                hipMemcpyDeviceToHost, h->stream));
    NDT2D_SYNC(h);
  }
  return ndt2d_match_fetch(h, out);
  NDT2D_C_CATCH(h)
}

// slots of the host-coherent block the sharded particle path uses (behind the few-pose path's
// statistics): a device's eight moment sums, their flag, and its updateStatistics result
constexpr int kPoseSumsSlot = kPfOutSlot + 8;
constexpr int kPoseSumsFlagSlot = kPoseSumsSlot + ndt2d::kPoseSumsFlagOffset;
constexpr int kPfShardOutSlot = kPfOutSlot + 24;
static_assert(kPfShardOutSlot + 8 <= kHostResDoubles, "host block too small");

static int score_poses_launch_impl(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                                   double * d_scores, double * d_stats, bool publish_sums);

int ndt2d_score_poses_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                             double * d_scores, double * d_stats)
{
  NDT2D_C_TRY
  return score_poses_launch_impl(h, d_poses_xyt, n_poses, d_scores, d_stats, false);
  NDT2D_C_CATCH(h)
}

int ndt2d_pose_sums_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses, double * d_scores)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  NDT2D_TEST_MAYBE_FAIL(h, "ndt2d_pose_sums_launch");
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  if ((rc = ensure_host_res(h)) != NDT2D_OK) return rc;
  return score_poses_launch_impl(h, d_poses_xyt, n_poses, d_scores, h->stats.ptr, true);
  NDT2D_C_CATCH(h)
}

int ndt2d_pose_sums_fetch(ndt2d_handle h, double * sums_out)
{
  NDT2D_C_TRY
  if (h == nullptr || sums_out == nullptr) return NDT2D_ERR_INVALID;
  if (!h->sums_pending) return fail(h, NDT2D_ERR_STATE, "ndt2d_pose_sums_fetch: nothing launched");
  NDT2D_HIP(h, hipSetDevice(h->device));
  h->sums_pending = false;
  const int rc = wait_host_flag(h, kPoseSumsFlagSlot, h->sums_seq, h->sums_pos);
  if (rc != NDT2D_OK) return rc;
  for (int k = 0; k < NDT2D_POSE_STATS_DOUBLES; ++k) sums_out[k] = h->host_res[kPoseSumsSlot + k];
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_finalize_totals_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                                    double * d_weights, const double * totals)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_poses_xyt == nullptr || d_weights == nullptr || totals == nullptr || n_poses == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_finalize_totals_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->ws_poses, ndt2d::poses_workspace_doubles(n_poses));
  if (rc != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES)) != NDT2D_OK) return rc;
  if ((rc = ensure_host_res(h)) != NDT2D_OK) return rc;
  hipError_t e = ndt2d::launch_pf_finalize(d_poses_xyt, n_poses, d_weights, nullptr, totals, h->ws_poses.ptr,
                                           h->stats.ptr + NDT2D_POSE_STATS_DOUBLES,
                                           h->host_res_dev + kPfShardOutSlot, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pf_finalize");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_result_read(ndt2d_handle h, double * out)
{
  NDT2D_C_TRY
  if (h == nullptr || out == nullptr) return NDT2D_ERR_INVALID;
  if (h->host_res == nullptr) return fail(h, NDT2D_ERR_STATE, "ndt2d_pf_result_read: nothing launched");
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int k = 0; k < NDT2D_PF_RESULT_DOUBLES; ++k) out[k] = h->host_res[kPfShardOutSlot + k];
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

static int score_poses_launch_impl(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                                   double * d_scores, double * d_stats, bool publish_sums)
{
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_score_poses_launch: no grid");
  if (h->n_beams == 0) return fail(h, NDT2D_ERR_STATE, "ndt2d_score_poses_launch: set_beams first");
  if (d_poses_xyt == nullptr || d_scores == nullptr || n_poses == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_score_poses_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->ws_poses, ndt2d::poses_workspace_doubles(n_poses));
  if (rc != NDT2D_OK) return rc;

  ndt2d::PosesArgs a{};
  a.grid = h->grid;
  a.beams_xy = h->beams_ptr;
  a.n_beams = static_cast<uint32_t>(h->n_beams);
  a.poses_xyt = d_poses_xyt;
  a.n_poses = n_poses;
  a.scores = d_scores;
  a.beam_rmax = h->beam_rmax;
  {
    // an occupancy bitmap too large for LDS: screen with one bit per block of cells
    const int k = ndt2d::poses_coarse_log2(a, ndt2d::poses_lds_per_block());
    if (k > 0)
    {
      if (h->coarse_log2 != k)
      {
        rc = ensure(h, h->coarse_bits, ndt2d::poses_coarse_words(h->grid, static_cast<uint32_t>(k)) / 2 + 2);
        if (rc != NDT2D_OK) return rc;
        hipError_t ce = ndt2d::poses_coarse_bits_launch(h->grid, static_cast<uint32_t>(k),
                                                        reinterpret_cast<uint32_t *>(h->coarse_bits.ptr), h->stream);
        if (ce != hipSuccess) return fail_hip(h, ce, "poses_coarse_bits_launch");
        h->coarse_log2 = k;
      }
      a.coarse_bits = reinterpret_cast<const uint32_t *>(h->coarse_bits.ptr);
      a.coarse_log2 = static_cast<uint32_t>(k);
    }
  }

  ndt2d::LaunchInfo info{"", 0};
  if (h->timing)
  {
    if (int trc = next_timing_slot(h); trc != NDT2D_OK) return trc;
    NDT2D_HIP(h, hipEventRecord(h->ev0, h->stream));
  }
  unsigned long long sums_seq = 0;
  if (publish_sums)
  {
    sums_seq = ++h->seq;
    h->sums_pos = h->queued;
  }
  hipError_t e = ndt2d::launch_score_poses(a, h->ws_poses.ptr, d_stats, h->force_variant,
                                           h->stream, h->timing ? h->ev1 : nullptr, &info,
                                           publish_sums ? h->host_res_dev + kPoseSumsSlot : nullptr, sums_seq);
  if (e != hipSuccess) return fail_hip(h, e, "launch_score_poses");
  if (publish_sums)
  {
    h->sums_seq = sums_seq;
    h->sums_pending = true;
  }
  h->timed = h->timing;
  h->last_kernels = info.n_kernels;
  h->last_variant = info.variant;
  return NDT2D_OK;
}

int ndt2d_score_poses_beams(ndt2d_handle h, const double * beams_xy, size_t n_beams,
                            const double * h_poses_xyt, size_t n_poses, double * h_scores)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (beams_xy == nullptr || n_beams == 0 || n_beams > (1u << 20) || h_poses_xyt == nullptr ||
      h_scores == nullptr || n_poses == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_score_poses_beams: bad argument");
  }
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_score_poses_beams: no grid");
  NDT2D_HIP(h, hipSetDevice(h->device));
  if (n_poses <= ndt2d::kFewPoses && n_beams <= ndt2d::kArgBeams &&
      h->force_variant == ndt2d::kVariantAuto && !h->batched_only)
  {
    // beams AND poses as kernel arguments: one launch, no copy; the kernel leaves the beams
    // in the context's beam buffer for the calls that follow on this scan
    int rc = ensure(h, h->beams, 2 * n_beams + 2);
    if (rc != NDT2D_OK) return rc;
    ndt2d::PosesArgs probe{};
    probe.n_beams = static_cast<uint32_t>(n_beams);
    probe.n_poses = n_poses;
    if (ndt2d::score_few_supported(probe, 64 * 1024))
    {
      // the kernel overwrites the context's beam buffer: until it has reported back the
      // context holds no beams (a failed wait must not leave new beams under the old
      // count and reach)
      h->n_beams = 0;
      h->beams_ptr = nullptr;
      h->has_search = false;
      rc = run_few(h, beams_xy, n_beams, h_poses_xyt, n_poses, false, h_scores, nullptr);
      if (rc != NDT2D_OK) return rc;
      h->n_beams = n_beams;
      h->beams_ptr = h->beams.ptr;
      h->has_search = false;   // see ndt2d_set_beams
      h->beam_rmax = beam_reach(beams_xy, n_beams);
      return NDT2D_OK;
    }
  }
  int rc = ndt2d_set_beams(h, beams_xy, n_beams);
  if (rc != NDT2D_OK) return rc;
  return ndt2d_score_poses(h, h_poses_xyt, n_poses, h_scores, nullptr);
  NDT2D_C_CATCH(h)
}

int ndt2d_score_poses_beams_launch(ndt2d_handle h, const double * beams_xy, size_t n_beams,
                                   const double * h_poses_xyt, size_t n_poses)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (h_poses_xyt == nullptr || n_poses == 0 || (beams_xy != nullptr && n_beams == 0))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_score_poses_beams_launch: bad argument");
  }
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_score_poses_beams_launch: no grid");
  if (beams_xy == nullptr) n_beams = h->n_beams;
  ndt2d::PosesArgs probe{};
  probe.n_beams = static_cast<uint32_t>(n_beams);
  probe.n_poses = n_poses;
  // only what travels as kernel arguments: nothing of the caller's is read after the return
  if (n_poses > ndt2d::kFewPoses || n_beams == 0 || (beams_xy != nullptr && n_beams > ndt2d::kArgBeams) ||
      h->force_variant != ndt2d::kVariantAuto || h->batched_only || !ndt2d::score_few_supported(probe, 64 * 1024))
  {
    return fail(h, NDT2D_ERR_STATE, "ndt2d_score_poses_beams_launch: not a kernel-argument launch (use ndt2d_score_poses_beams)");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  if (beams_xy == nullptr) return run_few_launch(h, nullptr, n_beams, h_poses_xyt, n_poses, false);
  int rc = ensure(h, h->beams, 2 * n_beams + 2);
  if (rc != NDT2D_OK) return rc;
  h->n_beams = 0;
  h->beams_ptr = nullptr;
  h->has_search = false;
  rc = run_few_launch(h, beams_xy, n_beams, h_poses_xyt, n_poses, false);
  if (rc != NDT2D_OK) return rc;
  // (the kernel is queued: what follows on the stream finds the beams in place; a failed
  // ndt2d_score_fetch takes them away again)
  h->n_beams = n_beams;
  h->beams_ptr = h->beams.ptr;
  h->beam_rmax = beam_reach(beams_xy, n_beams);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_score_fetch(ndt2d_handle h, double * h_scores)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (h_scores == nullptr) return fail(h, NDT2D_ERR_INVALID, "ndt2d_score_fetch: bad argument");
  const int rc = run_few_fetch(h, h_scores, nullptr);
  if (rc != NDT2D_OK && rc != NDT2D_ERR_STATE)
  {
    h->n_beams = 0;
    h->beams_ptr = nullptr;
    h->has_search = false;
  }
  return rc;
  NDT2D_C_CATCH(h)
}

// A LARGE batch of poses from (pageable or pinned) host memory -- ParticleFilter::measure of a
// big filter through the drop-in boundary (reference src/particle_filter.cpp:78-89; BASELINE
// configs[4]: 10^6 particles = 24 MB in, 8 MB out): the batch is cut into pieces, piece k + 1 is
// uploaded on a stream of its own while piece k is scored, and (scores only) piece k - 1 travels
// back on a third.  Sharding a pose set leaves every raw score bit-identical (a lane's score
// does not depend on the launch geometry), the pieces' moment sums are added in piece order.
//   measure: weights = scores normalised by the total, h_out = the statistics (updateStatistics)
//   else:    h_scores = raw scores, h_stats (optional) = the eight moment sums
constexpr size_t kPipelineFromPoses = 1u << 17;   // 3 MB of poses: below, the pieces cost more than they hide

static int pipeline_pieces_for(const ndt2d_context * h, size_t n_poses)
{
  if (h->pipeline_pieces == 1 || n_poses < kPipelineFromPoses) return 1;
  int pieces = h->pipeline_pieces > 0 ? h->pipeline_pieces : 4;
  // (no piece below a quarter of the threshold)
  while (pieces > 1 && n_poses / static_cast<size_t>(pieces) < kPipelineFromPoses / 4) --pieces;
  return pieces > NDT2D_PIPELINE_PIECES ? NDT2D_PIPELINE_PIECES : pieces;
}

static int pipelined_pose_batch(ndt2d_context * h, const double * h_poses_xyt, size_t n_poses, int pieces,
                                bool measure, double * h_scores, double * h_stats_or_out)
{
  int rc = ensure(h, h->tmp_poses, 3 * n_poses);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->tmp_scores, n_poses);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->piece_stats, static_cast<size_t>(NDT2D_PIPELINE_PIECES) * NDT2D_POSE_STATS_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  if (h->up_stream == nullptr) NDT2D_HIP(h, hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
  if (h->down_stream == nullptr) NDT2D_HIP(h, hipStreamCreateWithFlags(&h->down_stream, hipStreamNonBlocking));
  if (h->pipe_idle == nullptr) NDT2D_HIP(h, hipEventCreateWithFlags(&h->pipe_idle, hipEventDisableTiming));
  for (int k = 0; k < pieces; ++k)
  {
    if (h->pipe_up[k] == nullptr) NDT2D_HIP(h, hipEventCreateWithFlags(&h->pipe_up[k], hipEventDisableTiming));
    if (h->pipe_scored[k] == nullptr) NDT2D_HIP(h, hipEventCreateWithFlags(&h->pipe_scored[k], hipEventDisableTiming));
  }
  // Whatever fails from here on, nothing of this call may still be reading the caller's poses
  // or writing the caller's scores when it returns.
  auto drain = [&](int code) -> int {
    (void)hipStreamSynchronize(h->up_stream);
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->down_stream);
    return code;
  };
  // (what the main stream still has queued may read the staging buffers: the uploads wait for it)
  NDT2D_HIP(h, hipEventRecord(h->pipe_idle, h->stream));
  NDT2D_HIP(h, hipStreamWaitEvent(h->up_stream, h->pipe_idle, 0));
  const bool want_sums = measure || h_stats_or_out != nullptr;
  // Equal pieces of whole groups of 64 poses.  (A smaller first piece -- nothing runs under its
  // upload -- was measured and lost: the batched kernel's blocks walk their groups in rounds, and
  // pieces of 1/7 and 2/7 of cfg-5 fall between two round counts, 240 us for what should take 165.)
  // No event pairs around the pieces' kernels (ndt2d_launch_history_ms): a pair costs the stream
  // 11 us per piece.
  const size_t unit = ((n_poses + static_cast<size_t>(pieces) - 1) / static_cast<size_t>(pieces) + 63) / 64 * 64;
  const bool timing_was = h->timing;
  h->timing = false;
  auto finish = [&](int code) -> int {
    h->timing = timing_was;
    return code == NDT2D_OK ? code : drain(code);
  };
  int n_pieces = 0;
  hipError_t e = hipSuccess;
  size_t prev_off = 0, prev_cnt = 0;
  // raw scores: piece k - 1 travels back once piece k has been queued behind it (a download
  // into ordinary host memory holds the calling thread until the piece has been scored)
  auto download = [&](size_t off, size_t cnt, int k) -> hipError_t {
    hipError_t de = hipStreamWaitEvent(h->down_stream, h->pipe_scored[k], 0);
    if (de == hipSuccess)
    {
      de = hipMemcpyAsync(h_scores + off, h->tmp_scores.ptr + off, cnt * sizeof(double), hipMemcpyDeviceToHost,
                          h->down_stream);
    }
    return de;
  };
  for (size_t off = 0; off < n_poses; ++n_pieces)
  {
    const size_t cnt = n_poses - off < unit ? n_poses - off : unit;
    const int k = n_pieces;
    e = hipMemcpyAsync(h->tmp_poses.ptr + 3 * off, h_poses_xyt + 3 * off, 3 * cnt * sizeof(double),
                       hipMemcpyHostToDevice, h->up_stream);
    if (e == hipSuccess) e = hipEventRecord(h->pipe_up[k], h->up_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, h->pipe_up[k], 0);
    if (e != hipSuccess) return finish(fail_hip(h, e, "pipelined_pose_batch: upload"));
    rc = score_poses_launch_impl(h, h->tmp_poses.ptr + 3 * off, cnt, h->tmp_scores.ptr + off,
                                 want_sums ? h->piece_stats.ptr + static_cast<size_t>(k) * NDT2D_POSE_STATS_DOUBLES : nullptr,
                                 false);
    if (rc != NDT2D_OK) return finish(rc);
    if (!measure)
    {
      e = hipEventRecord(h->pipe_scored[k], h->stream);
      if (e == hipSuccess && k > 0) e = download(prev_off, prev_cnt, k - 1);
      if (e != hipSuccess) return finish(fail_hip(h, e, "pipelined_pose_batch: download"));
      prev_off = off;
      prev_cnt = cnt;
    }
    off += cnt;
  }
  if (!measure)
  {
    e = download(prev_off, prev_cnt, n_pieces - 1);
    if (e != hipSuccess) return finish(fail_hip(h, e, "pipelined_pose_batch: download"));
  }
  if (want_sums)
  {
    e = ndt2d::launch_sum_moment_rows(h->piece_stats.ptr, static_cast<uint32_t>(n_pieces), h->stats.ptr, h->stream);
    if (e != hipSuccess) return finish(fail_hip(h, e, "launch_sum_moment_rows"));
  }
  if (measure)
  {
    double * d_out = h->stats.ptr + NDT2D_POSE_STATS_DOUBLES;
    rc = ndt2d_pf_finalize_launch(h, h->tmp_poses.ptr, n_poses, h->tmp_scores.ptr, h->stats.ptr, d_out);
    if (rc != NDT2D_OK) return finish(rc);
    e = hipMemcpyAsync(h_scores, h->tmp_scores.ptr, n_poses * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess)
    {
      e = hipMemcpyAsync(h_stats_or_out, d_out, NDT2D_PF_RESULT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost,
                         h->stream);
    }
  }
  else if (h_stats_or_out != nullptr)
  {
    e = hipMemcpyAsync(h_stats_or_out, h->stats.ptr, NDT2D_POSE_STATS_DOUBLES * sizeof(double),
                       hipMemcpyDeviceToHost, h->stream);
  }
  if (e != hipSuccess) return finish(fail_hip(h, e, "pipelined_pose_batch: results"));
  e = hipStreamSynchronize(h->down_stream);
  if (e != hipSuccess) return finish(fail_hip(h, e, "hipStreamSynchronize"));
  {
    const uint64_t q_ = h->queued;
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return finish(fail_hip(h, e, "hipStreamSynchronize"));
    if (h->reached < q_) h->reached = q_;
  }
  h->timing = timing_was;
  h->last_pieces = n_pieces;
  return NDT2D_OK;
}

int ndt2d_score_poses(ndt2d_handle h, const double * h_poses_xyt, size_t n_poses,
                      double * h_scores, double * h_stats)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (h_poses_xyt == nullptr || h_scores == nullptr || n_poses == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_score_poses: bad argument");
  }
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_score_poses: no grid");
  if (h->n_beams == 0) return fail(h, NDT2D_ERR_STATE, "ndt2d_score_poses: set_beams first");
  NDT2D_HIP(h, hipSetDevice(h->device));
  h->last_pieces = 1;
  int rc;

  // A small batch (scorePoints / scoreScan: ONE pose; a particle filter of the node's
  // default size: <= 500): a block per pose and a thread per beam, the poses as kernel
  // arguments (<= 8) or through one staged copy, the scores written straight into
  // host-coherent memory behind a flag the host spins on -- one launch.
  if (h_stats == nullptr && n_poses <= ndt2d::kFewPosesMax &&
      h->force_variant == ndt2d::kVariantAuto && !h->batched_only)
  {
    ndt2d::PosesArgs probe{};
    probe.n_beams = static_cast<uint32_t>(h->n_beams);
    probe.n_poses = n_poses;
    if (ndt2d::score_few_supported(probe, 64 * 1024))
    {
      return run_few(h, nullptr, h->n_beams, h_poses_xyt, n_poses, false, h_scores, nullptr);
    }
  }

  rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  // Pinned host memory (ndt2d_host_alloc) is addressed by the kernels directly: the
  // poses are read over PCIe once, the scores written once, no copy is queued.
  const double * d_poses = device_view(h_poses_xyt);
  double * d_scores = device_view(h_scores);
  if (d_poses == nullptr && d_scores == nullptr)
  {
    // a large batch from ordinary host memory: uploads, scoring and downloads piece by piece, overlapped
    const int pieces = pipeline_pieces_for(h, n_poses);
    if (pieces > 1) return pipelined_pose_batch(h, h_poses_xyt, n_poses, pieces, false, h_scores, h_stats);
  }
  if (d_poses == nullptr)
  {
    rc = ensure(h, h->tmp_poses, 3 * n_poses);
    if (rc != NDT2D_OK) return rc;
    NDT2D_HIP(h, hipMemcpyAsync(h->tmp_poses.ptr, h_poses_xyt, 3 * n_poses * sizeof(double),
                                hipMemcpyHostToDevice, h->stream));
    d_poses = h->tmp_poses.ptr;
  }
  const bool copy_scores = d_scores == nullptr;
  if (copy_scores)
  {
    rc = ensure(h, h->tmp_scores, n_poses);
    if (rc != NDT2D_OK) return rc;
    d_scores = h->tmp_scores.ptr;
  }
  rc = ndt2d_score_poses_launch(h, d_poses, n_poses, d_scores, h_stats != nullptr ? h->stats.ptr : nullptr);
  if (rc != NDT2D_OK) return rc;
  if (copy_scores)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h_scores, d_scores, n_poses * sizeof(double), hipMemcpyDeviceToHost,
                                h->stream));
  }
  if (h_stats != nullptr)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h_stats, h->stats.ptr, NDT2D_POSE_STATS_DOUBLES * sizeof(double),
                                hipMemcpyDeviceToHost, h->stream));
  }
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_finalize_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n_poses,
                             double * d_weights, const double * d_stats, double * d_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_poses_xyt == nullptr || d_weights == nullptr || d_stats == nullptr || d_out == nullptr ||
      n_poses == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_finalize_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->ws_poses, ndt2d::poses_workspace_doubles(n_poses));
  if (rc != NDT2D_OK) return rc;
  hipError_t e = ndt2d::launch_pf_finalize(d_poses_xyt, n_poses, d_weights, d_stats, nullptr,
                                           h->ws_poses.ptr, d_out, nullptr, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pf_finalize");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_measure(ndt2d_handle h, const double * h_poses_xyt, size_t n_poses,
                     double * h_weights, double * h_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (h_poses_xyt == nullptr || h_weights == nullptr || h_out == nullptr || n_poses == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_measure: bad argument");
  }
  if (!h->has_grid) return fail(h, NDT2D_ERR_NO_GRID, "ndt2d_pf_measure: no grid");
  if (h->n_beams == 0) return fail(h, NDT2D_ERR_STATE, "ndt2d_pf_measure: set_beams first");
  NDT2D_HIP(h, hipSetDevice(h->device));
  h->last_pieces = 1;
  if (n_poses <= ndt2d::kFewPosesMax && h->force_variant == ndt2d::kVariantAuto && !h->batched_only)
  {
    // a filter of the node's size (<= 500 particles by default): scoring and
    // updateStatistics in ONE launch, weights and result through host-coherent memory
    ndt2d::PosesArgs probe{};
    probe.n_beams = static_cast<uint32_t>(h->n_beams);
    probe.n_poses = n_poses;
    if (ndt2d::score_few_supported(probe, 64 * 1024))
    {
      return run_few(h, nullptr, h->n_beams, h_poses_xyt, n_poses, true, h_weights, h_out);
    }
  }
  {
    // a large filter: the particles go up piece by piece under the scoring of the piece before
    const int pieces = pipeline_pieces_for(h, n_poses);
    if (pieces > 1) return pipelined_pose_batch(h, h_poses_xyt, n_poses, pieces, true, h_weights, h_out);
  }
  int rc = ensure(h, h->tmp_poses, 3 * n_poses);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->tmp_scores, n_poses);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  NDT2D_HIP(h, hipMemcpyAsync(h->tmp_poses.ptr, h_poses_xyt, 3 * n_poses * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
  rc = ndt2d_score_poses_launch(h, h->tmp_poses.ptr, n_poses, h->tmp_scores.ptr, h->stats.ptr);
  if (rc != NDT2D_OK) return rc;
  double * d_out = h->stats.ptr + NDT2D_POSE_STATS_DOUBLES;
  rc = ndt2d_pf_finalize_launch(h, h->tmp_poses.ptr, n_poses, h->tmp_scores.ptr, h->stats.ptr, d_out);
  if (rc != NDT2D_OK) return rc;
  NDT2D_HIP(h, hipMemcpyAsync(h_weights, h->tmp_scores.ptr, n_poses * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
  NDT2D_HIP(h, hipMemcpyAsync(h_out, d_out, NDT2D_PF_RESULT_DOUBLES * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_noise_launch(ndt2d_handle h, uint64_t seed, uint64_t step, uint64_t first_index,
                          size_t n, float * d_noise_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_noise_out == nullptr || n == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_noise_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  hipError_t e = ndt2d::launch_pf_noise(d_noise_out, n, seed, first_index, step, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pf_noise");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_motion_launch(ndt2d_handle h, double * d_poses_xyt, size_t n, double dx, double dy,
                           double dth, const double * alphas5, const float * d_noise,
                           uint64_t seed, uint64_t step, uint64_t first_index)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_poses_xyt == nullptr || alphas5 == nullptr || n == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_motion_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  const ndt2d::MotionParams p = motion_params(dx, dy, dth, alphas5);
  hipError_t e =
    ndt2d::launch_pf_motion(d_poses_xyt, n, p, d_noise, seed, first_index, step, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pf_motion");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_init_launch(ndt2d_handle h, double * d_poses_xyt, size_t n, double x, double y,
                         double theta, double sigma_x, double sigma_y, double sigma_theta,
                         const float * d_noise, uint64_t seed, uint64_t step,
                         uint64_t first_index)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_poses_xyt == nullptr || n == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_init_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  ndt2d::InitParams p;
  p.x = static_cast<float>(x);
  p.y = static_cast<float>(y);
  p.theta = static_cast<float>(theta);
  p.sigma_x = static_cast<float>(sigma_x);
  p.sigma_y = static_cast<float>(sigma_y);
  p.sigma_theta = static_cast<float>(sigma_theta);
  hipError_t e =
    ndt2d::launch_pf_init(d_poses_xyt, n, p, d_noise, seed, first_index, step, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pf_init");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pose_moments_launch(ndt2d_handle h, const double * d_poses_xyt, size_t n,
                              const double * d_weights, double * d_stats)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_poses_xyt == nullptr || d_stats == nullptr || n == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pose_moments_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->ws_poses, ndt2d::poses_workspace_doubles(n));
  if (rc != NDT2D_OK) return rc;
  hipError_t e =
    ndt2d::launch_pose_moments(d_poses_xyt, n, d_weights, h->ws_poses.ptr, d_stats, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_pose_moments");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_pf_update(ndt2d_handle h, double * h_poses_xyt, size_t n, double dx, double dy,
                    double dth, const double * alphas5, const float * h_noise, uint64_t seed,
                    uint64_t step, double * h_weights, double * h_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (h_poses_xyt == nullptr || alphas5 == nullptr || h_weights == nullptr || h_out == nullptr ||
      n == 0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_pf_update: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = ensure(h, h->tmp_poses, 3 * n);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->tmp_scores, n);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES);
  if (rc != NDT2D_OK) return rc;
  const float * d_noise = nullptr;
  if (h_noise != nullptr)
  {
    rc = ensure(h, h->tmp_noise, (3 * n * sizeof(float) + sizeof(double) - 1) / sizeof(double));
    if (rc != NDT2D_OK) return rc;
    NDT2D_HIP(h, hipMemcpyAsync(h->tmp_noise.ptr, h_noise, 3 * n * sizeof(float),
                                hipMemcpyHostToDevice, h->stream));
    d_noise = reinterpret_cast<const float *>(h->tmp_noise.ptr);
  }
  NDT2D_HIP(h, hipMemcpyAsync(h->tmp_poses.ptr, h_poses_xyt, 3 * n * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
  NDT2D_HIP(h, hipMemcpyAsync(h->tmp_scores.ptr, h_weights, n * sizeof(double),
                              hipMemcpyHostToDevice, h->stream));
  rc = ndt2d_pf_motion_launch(h, h->tmp_poses.ptr, n, dx, dy, dth, alphas5, d_noise, seed, step, 0);
  if (rc != NDT2D_OK) return rc;
  rc = ndt2d_pose_moments_launch(h, h->tmp_poses.ptr, n, h->tmp_scores.ptr, h->stats.ptr);
  if (rc != NDT2D_OK) return rc;
  double * d_out = h->stats.ptr + NDT2D_POSE_STATS_DOUBLES;
  rc = ndt2d_pf_finalize_launch(h, h->tmp_poses.ptr, n, h->tmp_scores.ptr, h->stats.ptr, d_out);
  if (rc != NDT2D_OK) return rc;
  NDT2D_HIP(h, hipMemcpyAsync(h_poses_xyt, h->tmp_poses.ptr, 3 * n * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
  NDT2D_HIP(h, hipMemcpyAsync(h_weights, h->tmp_scores.ptr, n * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
  NDT2D_HIP(h, hipMemcpyAsync(h_out, d_out, NDT2D_PF_RESULT_DOUBLES * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

}  // extern "C"

namespace
{

// The results of the pending few-pose launch: wait for its flag, copy them out.
int run_few_fetch(ndt2d_context * h, double * h_scores, double * h_out)
{
  if (!h->few_pending) return fail(h, NDT2D_ERR_STATE, "no few-pose launch is pending");
  h->few_pending = false;
  const int rc = wait_host_flag(h, kScoreFlagSlot, h->few_seq, h->few_pos);
  if (rc != NDT2D_OK) return rc;
  if (h_scores != nullptr) std::memcpy(h_scores, h->host_res + kScoreSlot, h->few_n_poses * sizeof(double));
  if (h->few_stats && h_out != nullptr)
  {
    for (int k = 0; k < NDT2D_PF_RESULT_DOUBLES; ++k) h_out[k] = h->host_res[kPfOutSlot + k];
  }
  return NDT2D_OK;
}

int run_few_launch(ndt2d_context * h, const double * arg_beams, size_t n_beams, const double * h_poses,
                   size_t n_poses, bool stats)
{
  if (h->few_pending)
  {
    // (results nobody collected: the slots are about to be written again)
    const int drc = run_few_fetch(h, nullptr, nullptr);
    if (drc != NDT2D_OK) return drc;
  }

  int rc = ensure_host_res(h);
  if (rc != NDT2D_OK) return rc;
  ndt2d::PosesArgs a{};
  a.grid = h->grid;
  a.beams_xy = arg_beams != nullptr ? h->beams.ptr : h->beams_ptr;
  a.n_beams = static_cast<uint32_t>(n_beams);
  a.n_poses = n_poses;
  a.scores = h->host_res_dev + kScoreSlot;
  ndt2d::FewPoses few{};
  bool poses_in_place = false;
  if (n_poses <= ndt2d::kFewPoses)
  {
    std::memcpy(few.xyt, h_poses, 3 * n_poses * sizeof(double));
    a.poses_xyt = nullptr;
  }
  else
  {
    // through the pinned staging buffer, which the kernel reads in place over PCIe (24 bytes
    // per block): no copy command, no event; the call returns after the kernel has finished,
    // so the buffer is free for the next one (the caller's buffer is free on return)
    if ((rc = ensure(h, h->tmp_poses, 3 * n_poses)) != NDT2D_OK) return rc;
    if ((rc = stage_acquire(h, h->stage_call, 3 * n_poses)) != NDT2D_OK) return rc;
    std::memcpy(h->stage_call.ptr, h_poses, 3 * n_poses * sizeof(double));
    if (h->stage_call.dev != nullptr)
    {
      a.poses_xyt = h->stage_call.dev;
      poses_in_place = true;
    }
    else
    {
      if ((rc = stage_submit(h, h->stage_call, h->tmp_poses.ptr, 3 * n_poses)) != NDT2D_OK) return rc;
      a.poses_xyt = h->tmp_poses.ptr;
    }
  }
  ndt2d::FewOut out{};
  out.flag = reinterpret_cast<unsigned long long *>(h->host_res_dev + kScoreFlagSlot);
  out.seq = ++h->seq;
  out.done = h->done_words;
  out.beams_out = arg_beams != nullptr ? h->beams.ptr : nullptr;
  out.stats = stats ? 1 : 0;
  if (stats)
  {
    if ((rc = ensure(h, h->tmp_scores, n_poses)) != NDT2D_OK) return rc;
    out.dev_scores = h->tmp_scores.ptr;
    out.host_out = h->host_res_dev + kPfOutSlot;
    out.dev_poses = poses_in_place ? h->tmp_poses.ptr : nullptr;
  }
  out.side = h->bytes_job;   // (n == 0: none)
  hipError_t e = ndt2d::launch_score_few(a, &few, out, arg_beams, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_score_few");
  // the kernel reads the staged poses in place: the buffer is its until it has finished
  if (poses_in_place && (rc = stage_mark(h, h->stage_call)) != NDT2D_OK) return rc;
  // (every mark so far is behind this kernel or behind work queued before it: its flag proves them all)
  const uint64_t flag_pos = h->queued;
  h->bytes_job.n = 0;
  h->timed = false;
  h->last_kernels = 1;
  h->last_variant = h->grid.pow2 ? "poses/block-per-pose/pow2" : "poses/block-per-pose/div";
  h->few_pending = true;
  h->few_seq = out.seq;
  h->few_pos = flag_pos;
  h->few_n_poses = n_poses;
  h->few_stats = stats;
  return NDT2D_OK;
}

int run_few(ndt2d_context * h, const double * arg_beams, size_t n_beams, const double * h_poses,
            size_t n_poses, bool stats, double * h_scores, double * h_out)
{
  const int rc = run_few_launch(h, arg_beams, n_beams, h_poses, n_poses, stats);
  return rc != NDT2D_OK ? rc : run_few_fetch(h, h_scores, h_out);
}

ndt2d::ScanDesc scan_desc(const ndt2d_laser_scan & s)
{
  ndt2d::ScanDesc d;
  d.angle_min = s.angle_min;
  d.angle_increment = s.angle_increment;
  d.range_max = s.range_max;
  d.inverted = s.inverted != 0 ? 1 : 0;
  d.laser_x = s.laser_x;
  d.laser_y = s.laser_y;
  // reference src/ndt_mapper.cpp:403-404 ("minor optimization"): host libm
  ndt2d_cos_sin(s.laser_theta, &d.cos_lt, &d.sin_lt);
  d.motion_x = s.motion_x;
  d.motion_y = s.motion_y;
  d.motion_theta = s.motion_theta;
  return d;
}

// H2D of the ranges into the context's scratch + conversion into scan_points;
// scan_info[0..1] receives {n, rmax}.
int upload_and_convert(ndt2d_context * h, const float * h_ranges, size_t n_ranges,
                       const ndt2d_laser_scan * scan)
{
  int rc = ensure(h, h->scan_ranges, (n_ranges * sizeof(float) + sizeof(double) - 1) / sizeof(double));
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->scan_points, 2 * n_ranges);
  if (rc != NDT2D_OK) return rc;
  rc = ensure(h, h->scan_info, 8);
  if (rc != NDT2D_OK) return rc;
  NDT2D_HIP(h, hipMemcpyAsync(h->scan_ranges.ptr, h_ranges, n_ranges * sizeof(float),
                              hipMemcpyHostToDevice, h->stream));
  return ndt2d_convert_scan_launch(h, reinterpret_cast<const float *>(h->scan_ranges.ptr), n_ranges,
                                   scan, h->scan_points.ptr, h->scan_info.ptr);
}

}  // namespace

extern "C" {

int ndt2d_convert_scan_launch(ndt2d_handle h, const float * d_ranges, size_t n_ranges,
                              const ndt2d_laser_scan * scan, double * d_points_xy_out,
                              double * d_info_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_ranges == nullptr || scan == nullptr || d_points_xy_out == nullptr ||
      d_info_out == nullptr || n_ranges == 0 || n_ranges > (1u << 24))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_convert_scan_launch: bad argument");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  hipError_t e = ndt2d::launch_convert_scan(d_ranges, static_cast<uint32_t>(n_ranges),
                                            scan_desc(*scan), d_points_xy_out, d_info_out,
                                            h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_convert_scan");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_convert_scan(ndt2d_handle h, const float * h_ranges, size_t n_ranges,
                       const ndt2d_laser_scan * scan, double * h_points_xy_out,
                       size_t * n_points_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (scan == nullptr || n_points_out == nullptr || (n_ranges > 0 && h_ranges == nullptr) ||
      (n_ranges > 0 && h_points_xy_out == nullptr))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_convert_scan: bad argument");
  }
  *n_points_out = 0;
  if (n_ranges == 0) return NDT2D_OK;  // an empty message holds no points
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = upload_and_convert(h, h_ranges, n_ranges, scan);
  if (rc != NDT2D_OK) return rc;
  double info[2] = {0.0, 0.0};
  NDT2D_HIP(h, hipMemcpyAsync(info, h->scan_info.ptr, sizeof(info), hipMemcpyDeviceToHost, h->stream));
  NDT2D_SYNC(h);
  const size_t n = static_cast<size_t>(info[0]);
  h->n_scan_points = n;
  if (n > 0)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h_points_xy_out, h->scan_points.ptr, 2 * n * sizeof(double),
                                hipMemcpyDeviceToHost, h->stream));
    NDT2D_SYNC(h);
  }
  *n_points_out = n;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_set_beams_from_ranges(ndt2d_handle h, const float * h_ranges, size_t n_ranges,
                                const ndt2d_laser_scan * scan, size_t laser_max_beams,
                                size_t * n_points_out, size_t * n_beams_out)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (scan == nullptr || n_beams_out == nullptr || (n_ranges > 0 && h_ranges == nullptr) ||
      laser_max_beams > (1u << 20))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_beams_from_ranges: bad argument");
  }
  *n_beams_out = 0;
  if (n_points_out != nullptr) *n_points_out = 0;
  h->n_scan_points = 0;
  if (n_ranges == 0 || laser_max_beams == 0) return NDT2D_OK;
  NDT2D_HIP(h, hipSetDevice(h->device));
  int rc = upload_and_convert(h, h_ranges, n_ranges, scan);
  if (rc != NDT2D_OK) return rc;
  const size_t cap = std::min(laser_max_beams, n_ranges);
  rc = ensure(h, h->beams, 2 * cap + 2);
  if (rc != NDT2D_OK) return rc;
  hipError_t e = ndt2d::launch_subsample(h->scan_points.ptr, h->scan_info.ptr,
                                         static_cast<uint32_t>(laser_max_beams), h->beams.ptr,
                                         h->scan_info.ptr + 2, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_subsample");
  double info[3] = {0.0, 0.0, 0.0};
  NDT2D_HIP(h, hipMemcpyAsync(info, h->scan_info.ptr + 2, sizeof(info), hipMemcpyDeviceToHost,
                              h->stream));
  NDT2D_SYNC(h);
  h->n_scan_points = static_cast<size_t>(info[0]);
  if (n_points_out != nullptr) *n_points_out = h->n_scan_points;
  const size_t use = static_cast<size_t>(info[1]);
  *n_beams_out = use;
  h->has_search = false;  // see ndt2d_set_beams
  if (use > 0)
  {
    h->n_beams = use;
    h->beams_ptr = h->beams.ptr;
    h->beam_rmax = info[2];
  }
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

const double * ndt2d_scan_points(ndt2d_handle h, size_t * n_points_out)
{
  if (h == nullptr) return nullptr;
  if (n_points_out != nullptr) *n_points_out = h->n_scan_points;
  return h->scan_points.ptr;
}

int ndt2d_occupancy_grid(ndt2d_handle h, double resolution, double occ_thresh,
                         const double * poses_xyt, const double * points_xy,
                         const size_t * offsets, size_t n_scans, size_t n_scans_bounded,
                         double * bounds_inout, ndt2d_occupancy_info * info_out,
                         signed char * data_out, size_t data_capacity)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (!(resolution > 0.0) || bounds_inout == nullptr || info_out == nullptr ||
      n_scans_bounded > n_scans || n_scans > (1u << 30) ||
      (n_scans > 0 && (poses_xyt == nullptr || offsets == nullptr)))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_occupancy_grid: bad argument");
  }
  const size_t n_points = n_scans > 0 ? offsets[n_scans] : 0;
  if (n_points >= (1ull << 31) || (n_points > 0 && points_xy == nullptr))
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_occupancy_grid: bad points");
  }
  NDT2D_HIP(h, hipSetDevice(h->device));

  // scans and points to the device (the NDT build's buffers and staging are reused)
  NDT2D_SYNC(h);
  h->stage_scans.resize(4 * n_scans + 4);
  h->stage_offsets.resize(((n_scans + 1) + 1) & ~size_t(1));
  for (size_t k = 0; k < n_scans; ++k)
  {
    h->stage_scans[4 * k] = poses_xyt[3 * k];
    h->stage_scans[4 * k + 1] = poses_xyt[3 * k + 1];
    ndt2d_cos_sin(poses_xyt[3 * k + 2], &h->stage_scans[4 * k + 2], &h->stage_scans[4 * k + 3]);  // :78-79,163-164, host libm
    h->stage_offsets[k] = static_cast<uint32_t>(offsets[k]);
  }
  h->stage_offsets[n_scans] = static_cast<uint32_t>(n_points);
  int rc;
  if ((rc = ensure(h, h->b_points, 2 * n_points + 2)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_scans, 4 * n_scans + 4)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->b_offsets, h->stage_offsets.size() / 2 + 1)) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->ws_poses, ndt2d::poses_workspace_doubles(n_points))) != NDT2D_OK) return rc;
  if ((rc = ensure(h, h->stats, NDT2D_POSE_STATS_DOUBLES + NDT2D_PF_RESULT_DOUBLES)) != NDT2D_OK) return rc;
  if (n_points > 0)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h->b_points.ptr, points_xy, 2 * n_points * sizeof(double),
                                hipMemcpyHostToDevice, h->stream));
  }
  if (n_scans > 0)
  {
    NDT2D_HIP(h, hipMemcpyAsync(h->b_scans.ptr, h->stage_scans.data(), 4 * n_scans * sizeof(double),
                                hipMemcpyHostToDevice, h->stream));
  }
  NDT2D_HIP(h, hipMemcpyAsync(h->b_offsets.ptr, h->stage_offsets.data(),
                              (n_scans + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));

  ndt2d::OccupancyArgs args{};
  args.points_xy = h->b_points.ptr;
  args.n_points = static_cast<uint32_t>(n_points);
  args.scans = h->b_scans.ptr;
  args.offsets = reinterpret_cast<const uint32_t *>(h->b_offsets.ptr);
  args.n_scans = static_cast<uint32_t>(n_scans);
  args.resolution = resolution;

  // updateBounds (:154-185), only when the scan count changed (:51-54)
  if (n_scans != n_scans_bounded)
  {
    double found[4] = {HUGE_VAL, -HUGE_VAL, HUGE_VAL, -HUGE_VAL};
    const size_t first_point = offsets[n_scans_bounded];
    if (n_points > first_point)
    {
      hipError_t e = ndt2d::launch_occupancy_bounds(args, static_cast<uint32_t>(first_point),
                                                    h->ws_poses.ptr, h->stats.ptr, h->stream);
      if (e != hipSuccess) return fail_hip(h, e, "launch_occupancy_bounds");
      NDT2D_HIP(h, hipMemcpyAsync(found, h->stats.ptr, sizeof(found), hipMemcpyDeviceToHost, h->stream));
      NDT2D_SYNC(h);
    }
    const double min_x = std::min(found[0], bounds_inout[0]);
    const double max_x = std::max(found[1], bounds_inout[1]);
    const double min_y = std::min(found[2], bounds_inout[2]);
    const double max_y = std::max(found[3], bounds_inout[3]);
    // :181-184
    bounds_inout[0] = std::floor(min_x / resolution) * resolution;
    bounds_inout[1] = std::ceil(max_x / resolution) * resolution;
    bounds_inout[2] = std::floor(min_y / resolution) * resolution;
    bounds_inout[3] = std::ceil(max_y / resolution) * resolution;
  }

  // :57-65 (info.width / height are uint32: the quotient is truncated)
  const double pad = 5 * resolution;
  const double fw = (bounds_inout[1] - bounds_inout[0] + 2 * pad) / resolution;
  const double fh = (bounds_inout[3] - bounds_inout[2] + 2 * pad) / resolution;
  if (!(fw >= 0.0) || !(fh >= 0.0) || fw >= 2147483648.0 || fh >= 2147483648.0 ||
      fw * fh >= 2147483648.0)
  {
    return fail(h, NDT2D_ERR_INVALID, "ndt2d_occupancy_grid: degenerate map extent");
  }
  args.width = static_cast<uint32_t>(fw);
  args.height = static_cast<uint32_t>(fh);
  args.origin_x = bounds_inout[0] - pad;
  args.origin_y = bounds_inout[2] - pad;
  info_out->resolution = resolution;
  info_out->width = args.width;
  info_out->height = args.height;
  info_out->origin_x = args.origin_x;
  info_out->origin_y = args.origin_y;
  if (data_out == nullptr)
  {
    NDT2D_SYNC(h);  // the staged uploads are done with
    return NDT2D_OK;
  }
  const size_t n_cells = static_cast<size_t>(args.width) * args.height;
  if (data_capacity < n_cells) return fail(h, NDT2D_ERR_INVALID, "ndt2d_occupancy_grid: data_out too small");
  if (n_cells == 0) return NDT2D_OK;
  // counters (8 B per cell) and the int8 map behind them
  if ((rc = ensure(h, h->b_world, n_cells + (n_cells + 7) / 8 + 2)) != NDT2D_OK) return rc;
  unsigned long long * counts = reinterpret_cast<unsigned long long *>(h->b_world.ptr);
  signed char * d_data = reinterpret_cast<signed char *>(h->b_world.ptr + n_cells);
  hipError_t e = ndt2d::launch_occupancy_render(args, occ_thresh, counts, d_data, h->stream);
  if (e != hipSuccess) return fail_hip(h, e, "launch_occupancy_render");
  NDT2D_HIP(h, hipMemcpyAsync(data_out, d_data, n_cells, hipMemcpyDeviceToHost, h->stream));
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_device_alloc(ndt2d_handle h, size_t bytes, void ** d_out)
{
  NDT2D_C_TRY
  if (h == nullptr || d_out == nullptr) return NDT2D_ERR_INVALID;
  *d_out = nullptr;
  if (bytes == 0) return fail(h, NDT2D_ERR_INVALID, "ndt2d_device_alloc: zero size");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipMalloc(d_out, bytes));
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_device_free(ndt2d_handle h, void * d_ptr)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (d_ptr == nullptr) return NDT2D_OK;
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_SYNC(h);  // nothing in flight may still use it
  NDT2D_HIP(h, hipFree(d_ptr));
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_copy_to_device(ndt2d_handle h, void * d_dst, const void * h_src, size_t bytes)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (bytes == 0) return NDT2D_OK;
  if (d_dst == nullptr || h_src == nullptr) return fail(h, NDT2D_ERR_INVALID, "ndt2d_copy_to_device: null pointer");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, h->stream));
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_copy_to_host(ndt2d_handle h, void * h_dst, const void * d_src, size_t bytes)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (bytes == 0) return NDT2D_OK;
  if (h_dst == nullptr || d_src == nullptr) return fail(h, NDT2D_ERR_INVALID, "ndt2d_copy_to_host: null pointer");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_copy_to_device_async(ndt2d_handle h, void * d_dst, const void * h_src, size_t bytes)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (bytes == 0) return NDT2D_OK;
  if (d_dst == nullptr || h_src == nullptr) return fail(h, NDT2D_ERR_INVALID, "ndt2d_copy_to_device_async: null pointer");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, h->stream));
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_copy_to_host_async(ndt2d_handle h, void * h_dst, const void * d_src, size_t bytes)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  if (bytes == 0) return NDT2D_OK;
  if (h_dst == nullptr || d_src == nullptr) return fail(h, NDT2D_ERR_INVALID, "ndt2d_copy_to_host_async: null pointer");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_set_timing(ndt2d_handle h, int enabled)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  h->timing = enabled != 0;
  if (!h->timing) h->timed = false;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_host_alloc(ndt2d_handle h, size_t bytes, void ** out)
{
  NDT2D_C_TRY
  if (h == nullptr || out == nullptr) return NDT2D_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return fail(h, NDT2D_ERR_INVALID, "ndt2d_host_alloc: zero size");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocPortable));
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_host_free(ndt2d_handle h, void * ptr)
{
  NDT2D_C_TRY
  if (ptr == nullptr) return h != nullptr ? NDT2D_OK : NDT2D_ERR_INVALID;
  if (h == nullptr)   // the owning context is gone, and with it everything that could use ptr
  {
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? NDT2D_OK : NDT2D_ERR_HIP;
  }
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_SYNC(h);  // nothing in flight may still use it
  NDT2D_HIP(h, hipHostFree(ptr));
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_synchronize(ndt2d_handle h)
{
  NDT2D_C_TRY
  if (h == nullptr) return NDT2D_ERR_INVALID;
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_SYNC(h);
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_last_launch_ms(ndt2d_handle h, float * ms, int * n_kernels)
{
  NDT2D_C_TRY
  if (h == nullptr || ms == nullptr) return NDT2D_ERR_INVALID;
  if (!h->timed) return fail(h, NDT2D_ERR_STATE, "ndt2d_last_launch_ms: nothing launched");
  NDT2D_HIP(h, hipSetDevice(h->device));
  NDT2D_HIP(h, hipEventSynchronize(h->ev1));
  NDT2D_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  if (n_kernels != nullptr) *n_kernels = h->last_kernels;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_launch_history_ms(ndt2d_handle h, float * ms_out, size_t capacity, size_t * n_out)
{
  NDT2D_C_TRY
  if (h == nullptr || n_out == nullptr || (capacity > 0 && ms_out == nullptr)) return NDT2D_ERR_INVALID;
  *n_out = 0;
  NDT2D_HIP(h, hipSetDevice(h->device));
  size_t n = static_cast<size_t>(std::min<uint64_t>(h->n_timed_launches, NDT2D_TIMING_HISTORY));
  if (n > capacity) n = capacity;
  if (n == 0) return NDT2D_OK;
  NDT2D_HIP(h, hipEventSynchronize(h->ev1));  // the newest; the stream is in order
  for (size_t k = 0; k < n; ++k)
  {
    const size_t slot = static_cast<size_t>((h->n_timed_launches - n + k) % NDT2D_TIMING_HISTORY);
    NDT2D_HIP(h, hipEventElapsedTime(&ms_out[k], h->ring0[slot], h->ring1[slot]));
  }
  *n_out = n;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

const char * ndt2d_last_variant(ndt2d_handle h) { return h != nullptr ? h->last_variant : ""; }

int ndt2d_set_pipeline_pieces(ndt2d_handle h, int pieces)
{
  NDT2D_C_TRY
  if (h == nullptr || pieces < 0 || pieces > NDT2D_PIPELINE_PIECES) return NDT2D_ERR_INVALID;
  h->pipeline_pieces = pieces;
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

int ndt2d_last_pipeline_pieces(ndt2d_handle h)
{
  NDT2D_C_TRY
  return h == nullptr ? 0 : h->last_pieces;
  NDT2D_C_CATCH(h)
}

int ndt2d_set_variant(ndt2d_handle h, const char * name)
{
  NDT2D_C_TRY
  if (h == nullptr || name == nullptr) return NDT2D_ERR_INVALID;
  h->batched_only = false;
  if (std::strcmp(name, "auto") == 0) h->force_variant = ndt2d::kVariantAuto;
  else if (std::strcmp(name, "batched") == 0) h->batched_only = true, h->force_variant = ndt2d::kVariantAuto;
  else if (std::strcmp(name, "lds") == 0) h->force_variant = ndt2d::kVariantLds;
  else if (std::strcmp(name, "global") == 0) h->force_variant = ndt2d::kVariantGlobal;
  else if (std::strcmp(name, "wave") == 0) h->force_variant = ndt2d::kVariantWave;
  else if (std::strcmp(name, "wave-lds") == 0) h->force_variant = ndt2d::kVariantWave | ndt2d::kVariantLds;
  else if (std::strcmp(name, "wave-global") == 0) h->force_variant = ndt2d::kVariantWave | ndt2d::kVariantGlobal;
  else if (std::strcmp(name, "lane") == 0) h->force_variant = ndt2d::kVariantLane;
  else if (std::strcmp(name, "lane-noskip") == 0) h->force_variant = ndt2d::kVariantLane | ndt2d::kVariantNoSkip;
  else if (std::strcmp(name, "small") == 0) h->force_variant = ndt2d::kVariantSmall;
  else if (std::strcmp(name, "small-noskip") == 0) h->force_variant = ndt2d::kVariantSmall | ndt2d::kVariantNoSkip;
  else if (std::strcmp(name, "dense") == 0) h->force_variant = ndt2d::kVariantDense;
  else if (std::strcmp(name, "compact-exact") == 0) h->force_variant = ndt2d::kVariantNoSkip;
  else return fail(h, NDT2D_ERR_INVALID, "ndt2d_set_variant: unknown variant");
  return NDT2D_OK;
  NDT2D_C_CATCH(h)
}

}  // extern "C"

#ifdef NDT2D_TEST_HOOKS
extern "C" int ndt2d_test_fail_launch(int kth)
{
  NDT2D_C_TRY
  g_test_fail_launch.store(kth);
  return 0;
  NDT2D_C_CATCH(nullptr)
}
#endif
