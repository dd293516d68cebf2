"""ndt_2d::ParticleFilter with the particle set resident in HBM.

Mirror of the reference's localisation filter (reference
include/ndt_2d/particle_filter.hpp:49-115, src/particle_filter.cpp) for the part
of it that lies either side of the measurement hot path: `init`, `update`
(MotionModel::sample), `measure` and `updateStatistics` all run as HIP kernels on
one [n, 3] pose array that never leaves the device between calls.  `resample`
(KLD sampling over the KD-tree leaf count, src/particle_filter.cpp:91-140) stays
on the host, as SURVEY.md section 8(f) row N3 scopes it: it is a sequential
stopping rule over weighted draws; only the gather of the chosen particles runs
on the device.

torch is used for device memory and the stream only.
"""
import numpy as np

from . import _capi

# leaf size of the KD-tree the reference bins particles with (particle_filter.cpp:44)
KD_LEAF = (0.5, 0.5, 0.2671)


class MotionModel:
    """ndt_2d::MotionModel (reference include/ndt_2d/motion_model.hpp:44-66): the five
    noise parameters; sampling itself happens in ParticleFilter.update on the device."""

    def __init__(self, a1, a2, a3, a4, a5):
        self.alphas = np.array([a1, a2, a3, a4, a5], dtype=np.float64)


def kld_leaf_keys(particles, leaf=KD_LEAF):
    """Discrete KD-tree key of each pose (reference include/ndt_2d/kd_tree.hpp:95-98):
    static_cast<int>(value / leaf size), i.e. truncation toward zero."""
    p = np.asarray(particles, dtype=np.float64).reshape(-1, 3)
    return np.trunc(p / np.array(leaf, dtype=np.float64)).astype(np.int64)


def kld_leaf_count(keys):
    """KDTree::getLeafCount after inserting `keys`: one leaf per distinct key
    (kd_tree.hpp:125-137 merges equal keys, :152-165 splits otherwise)."""
    return len(np.unique(np.asarray(keys).reshape(-1, 3), axis=0))


def kld_resample_indices(draws, keys, min_particles, max_particles, kld_err, kld_z):
    """The stopping rule of ParticleFilter::resample (reference particle_filter.cpp:
    106-134) applied to a pre-drawn index sequence `draws` (at least max_particles
    long): returns the prefix of `draws` the reference's loop would keep.  `keys`
    are the leaf keys of all particles (kld_leaf_keys)."""
    if max_particles == 0:
        return draws[:0]
    draws = np.asarray(draws[:max_particles])
    k3 = keys[draws]
    # running number of distinct leaves after each insert (kd_tree.hpp: leaf_count_)
    _, first = np.unique(k3, axis=0, return_index=True)
    is_new = np.zeros(len(draws), dtype=np.int64)
    is_new[first] = 1
    k = np.cumsum(is_new)
    # Mx after each insert (:119-126); until a second leaf appears it stays max_particles
    mx = np.full(len(draws), float(max_particles))
    multi = k > 1
    km1 = (k[multi] - 1).astype(np.float64)
    a = km1 / (2.0 * kld_err)
    b = 2.0 / (9.0 * km1)
    c = 1.0 - b + np.sqrt(b) * kld_z
    mx[multi] = np.floor(a * c * c * c)   # size_t Mx = double
    size = np.arange(1, len(draws) + 1)
    stop = (size >= np.maximum(float(min_particles), mx)) | (size >= max_particles)
    n_keep = int(np.argmax(stop)) + 1 if stop.any() else len(draws)
    return draws[:n_keep]


def kld_resample_native(particles, weights, min_particles, max_particles, kld_err, kld_z, uniforms,
                        leaf=KD_LEAF):
    """ParticleFilter::resample's loop (reference particle_filter.cpp:94-134) through the
    library's host entry point ndt2d_kld_resample: returns the indices of the draws kept.
    `uniforms` (at least max_particles values in [0, 1)) stand for the generator."""
    import ctypes as C
    L = _capi.lib()
    pa = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 3)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    lf = np.array(leaf, dtype=np.float64)
    out = np.empty(max(int(max_particles), 1), dtype=np.uint32)
    n_out = C.c_size_t(0)
    rc = L.ndt2d_kld_resample(_capi.dptr(pa), _capi.dptr(w), len(w), int(min_particles),
                              int(max_particles), float(kld_err), float(kld_z), _capi.dptr(lf),
                              _capi.dptr(u), len(u), out.ctypes.data_as(C.POINTER(C.c_uint32)),
                              C.byref(n_out))
    if rc != _capi.OK:
        raise _capi.Ndt2dError(rc, "ndt2d_kld_resample")
    return out[:n_out.value].copy()


class ParticleFilter:
    """ndt_2d::ParticleFilter over the MI355X kernels.  `matcher` supplies the device
    context (an ndt_2d_amd.ScanMatcherNDT); `seed` keys the Philox noise stream
    that stands in for the reference's random_device-seeded mt19937."""

    def __init__(self, min_particles, max_particles, motion_model, matcher, seed=0):
        import torch
        if not torch.cuda.is_available():
            raise _capi.Ndt2dError(_capi.ERR_NO_DEVICE, "ParticleFilter",
                                   "no usable GPU; this library has no CPU fallback")
        self._torch = torch
        self._matcher = matcher
        self._L = _capi.lib()
        self.device = torch.device("cuda", self._L.ndt2d_device_id(matcher.device_handle))
        self._stream = torch.cuda.Stream(device=self.device)
        matcher.set_stream(self._stream.cuda_stream)
        self.min_particles = int(min_particles)
        self.max_particles = int(max_particles)
        self.motion_model = motion_model
        self.seed = int(seed)
        self._step = 0
        self._host_rng = np.random.Generator(np.random.Philox(key=self.seed))
        with torch.cuda.stream(self._stream):
            # reference particle_filter.cpp:48-50
            self.particles = torch.zeros((self.min_particles, 3), dtype=torch.float64,
                                         device=self.device)
            self.weights = torch.full((self.min_particles,), 1.0 / self.min_particles,
                                      dtype=torch.float64, device=self.device)
            self._stats = torch.zeros(_capi.POSE_STATS_DOUBLES + _capi.PF_RESULT_DOUBLES,
                                      dtype=torch.float64, device=self.device)
        self._mean = np.zeros(3)
        self._cov = np.zeros((3, 3))
        self._update_statistics(have_moments=False)

    # -- reference interface ------------------------------------------------------------

    def init(self, x, y, theta, sigma_x, sigma_y, sigma_theta):
        """reference src/particle_filter.cpp:53-69"""
        n = len(self.particles)
        self._matcher.pf_init_launch(self.particles.data_ptr(), n, x, y, theta, sigma_x, sigma_y,
                                     sigma_theta, None, self.seed, self._next_step(), 0)
        with self._torch.cuda.stream(self._stream):
            self.weights.fill_(1.0 / n)
        self._update_statistics(have_moments=False)

    def update(self, dx, dy, dth):
        """reference src/particle_filter.cpp:71-76"""
        self._matcher.pf_motion_launch(self.particles.data_ptr(), len(self.particles), dx, dy, dth,
                                       self.motion_model.alphas, None, self.seed,
                                       self._next_step(), 0)
        self._update_statistics(have_moments=False)

    def measure(self, matcher, points):
        """reference src/particle_filter.cpp:78-89; `points` are the scan's points."""
        if matcher is not self._matcher:
            raise ValueError("measure() must use the matcher whose device holds the particles")
        matcher.prepare_beams(points)
        matcher.score_poses_launch(self.particles.data_ptr(), len(self.particles),
                                   self.weights.data_ptr(), self._stats.data_ptr())
        self._update_statistics(have_moments=True)

    def resample(self, kld_err, kld_z):
        """reference src/particle_filter.cpp:91-140"""
        torch = self._torch
        self._stream.synchronize()
        pa = self.particles.cpu().numpy()
        w = self.weights.cpu().numpy()
        # the draw-and-stop loop itself is native host code shared with the C++ mirror
        keep = kld_resample_native(pa, w, self.min_particles, self.max_particles, kld_err, kld_z,
                                   self._host_rng.random(self.max_particles))
        with torch.cuda.stream(self._stream):
            idx = torch.from_numpy(keep.astype(np.int64)).to(self.device)
            self.particles = self.particles.index_select(0, idx).contiguous()
            self.weights = self.weights.index_select(0, idx).contiguous()
        self._update_statistics(have_moments=False)

    def getMean(self):
        return self._mean.copy()

    def getCovariance(self):
        return self._cov.copy()

    def getMsg(self):
        """Pose array as [n, 4] = {x, y, orientation.z, orientation.w}
        (reference src/particle_filter.cpp:152-161)."""
        self._stream.synchronize()
        pa = self.particles.cpu().numpy()
        return np.stack([pa[:, 0], pa[:, 1], np.sin(pa[:, 2] / 2.0), np.cos(pa[:, 2] / 2.0)], axis=1)

    # -- internals ----------------------------------------------------------------------

    def _next_step(self):
        self._step += 1
        return self._step

    def _update_statistics(self, have_moments):
        """reference src/particle_filter.cpp:163-218 on the device."""
        n = len(self.particles)
        st = self._stats
        if not have_moments:
            self._matcher.pose_moments_launch(self.particles.data_ptr(), n,
                                              self.weights.data_ptr(), st.data_ptr())
        out_ptr = st.data_ptr() + 8 * _capi.POSE_STATS_DOUBLES
        self._matcher.pf_finalize_launch(self.particles.data_ptr(), n, self.weights.data_ptr(),
                                         st.data_ptr(), out_ptr)
        with self._torch.cuda.stream(self._stream):
            out = st[_capi.POSE_STATS_DOUBLES:].cpu().numpy()
        self._stream.synchronize()
        self._mean = out[1:4].copy()
        c = self._cov
        c[0, 0] = out[4]
        c[0, 1] = c[1, 0] = out[5]
        c[1, 1] = out[6]
        c[2, 2] += out[7]   # never zeroed by the reference either (:216)
