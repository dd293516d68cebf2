"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

matchScan: the theta axis of the candidate lattice (the outer loop of the
reference, src/scan_matcher_ndt.cpp:103) is dealt out round-robin: rank r takes
the steps r, r + world, r + 2 world, ...  (Contiguous slabs are supported too,
but the cost of a step varies across the angular range -- the steps around the
scan's own heading hit the map far more often -- and at 8 ranks the busiest
contiguous slab takes 1.19x the mean, experiments/slab_balance.py.)
Every rank reduces its share on its own GPU to one 12-double record
{best_score, best_index, k00,k01,k02,k11,k12,k22, u0,u1,u2, s}, best_index being
the flat index in the WHOLE lattice; the records are exchanged by ONE
all-reduce(sum) of a [world, 12] buffer in which each rank fills only its own
row (x + 0 is exact, so the exchange is bit-deterministic), then combined with
the reference's first-wins rule: the lower score, and between equal scores the
lower flat index = the candidate the reference's loops visit first.

ParticleFilter::measure: particles are split into contiguous ranges; each rank
scores its range and reduces {sum w, sum w*x, ...} (8 doubles); one
all-reduce(sum) of a [world, 8] buffer gives every rank the total particle
weight (reference src/particle_filter.cpp:166-174) and the moment sums -- the ONE
collective of a step.  The theta variance is the reference's second pass
(:213-217: it needs the circular mean, which every rank has after that all-reduce): each
rank sums w d^2 over ITS particles and the share stays with the rank's weights;
cov(2,2) is the shares added in rank order (combine_theta_shares) by whoever collects the
weights -- the resampler, which needs all of them anyway (src/particle_filter.cpp:91-137).
"""
import math

import numpy as np

MATCH_RECORD = 12
POSE_STATS = 8


def shard_range(n, rank, world):
    """Contiguous [begin, end) share of n items for `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_strided(n, rank, world):
    """Round-robin share of n items for `rank`: (first, stride, count)."""
    return rank, world, (n - rank + world - 1) // world if rank < n else 0


def combine_match_records(records):
    """records[world][12] in rank order -> (best_score, best_index or None, acc[10]).

    The reference's first-wins rule (src/scan_matcher_ndt.cpp:128, strict `<` in
    visiting order): the lower score wins, and between equal scores the lower flat
    index -- for contiguous slabs that is the lower rank, for interleaved shares it
    is whoever holds the earlier candidate.  The accumulators are summed in rank order."""
    best_score, best_index, acc, _ = combine_match_records_ex(records)
    return best_score, best_index, acc


NEAR_TIE_REL = 2.0 ** -36      # NDT2D_NEAR_TIE_REL of include/ndt2d_hip.h


def combine_match_records_ex(records):
    """combine_match_records plus the near-tie mark: a record's index ending in .5 says another
    candidate of that rank's share scored within the near-tie tolerance of its winner (include/ndt2d_hip.h,
    ndt2d_match_result.near_tie); two ranks' winners that close mark the result as well.  Returns
    (best_score, best_index, acc, near_tie): with near_tie the caller should settle the winner as
    ndt2d_matcher_match_scan does (ndt2d_match_near_best + the reference's arithmetic)."""
    records = np.asarray(records, dtype=np.float64).reshape(-1, MATCH_RECORD)
    best_score, best_index, marked = 0.0, None, False
    acc = np.zeros(10, dtype=np.float64)
    for rec in records:
        if rec[1] >= 0.0 and rec[0] < 0.0:
            near = best_index is not None and \
                abs(rec[0] - best_score) <= max(abs(rec[0]), abs(best_score)) * NEAR_TIE_REL
            if best_index is None or rec[0] < best_score or \
                    (rec[0] == best_score and int(rec[1]) < best_index):
                best_score, best_index = float(rec[0]), int(rec[1])
                marked = rec[1] != math.floor(rec[1])
            marked = marked or near
        acc += rec[2:]
    return best_score, best_index, acc, marked


def match_scan_sharded(matcher, scan_pose, points, rank, world, dist, pose=None):
    """ScanMatcherNDT::matchScan (reference src/scan_matcher_ndt.cpp:76-149) with the
    lattice dealt to `world` ranks: every rank calls this with its own matcher (one per
    GPU, same map, same scan) and gets the same result as a single-GPU matchScan would
    give -- dict(score, pose, covariance, best_index, n_candidates).

    `dist` is torch.distributed with an initialised process group (nccl = RCCL, or
    gloo); the one collective is the all-reduce of the [world, 12] record table."""
    import torch
    n_th, n_lin, _ = matcher.prepare_search(scan_pose, points)
    first, stride, count = shard_strided(n_th, rank, world)
    device = torch.device("cuda", matcher._L.ndt2d_device_id(matcher.device_handle))
    on_gpu = world == 1 or dist.get_backend() != "gloo"
    stream = torch.cuda.Stream(device=device)
    # the stream the caller had bound (e.g. a ParticleFilter sharing this matcher binds its
    # own for good) comes back afterwards; the context's own stream reads as None
    previous = matcher.get_stream()
    matcher.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            table = torch.zeros((world, MATCH_RECORD), dtype=torch.float64, device=device)
            table[rank, 1] = -1.0   # "no candidate" until the search says otherwise
            if count:
                matcher.match_launch_strided(first, stride, count,
                                             record_ptr=table[rank].data_ptr())
            if world > 1:
                if on_gpu:
                    dist.all_reduce(table, op=dist.ReduceOp.SUM)
                else:
                    host = table.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    table = host
            records = table.cpu().numpy()
        stream.synchronize()
    finally:
        matcher.set_stream(previous)
    best_score, best_index, acc, marked = combine_match_records_ex(records)
    rec = np.concatenate([[best_score, -1.0 if best_index is None else float(best_index) + (0.5 if marked else 0.0)],
                          acc])
    if marked:
        # a rival within the near-tie tolerance (2^-36 relative) of the winner: every rank settles it for itself on the whole lattice
        # (same data, same arithmetic, same verdict -- no further exchange)
        rec = matcher.settle_near_tie(scan_pose, rec)
        best_index = int(rec[1]) if rec[1] >= 0.0 else None
    out = matcher.finish_match(rec, pose=pose)
    out["best_index"] = best_index
    out["near_tie"] = bool(marked)
    out["n_candidates"] = n_th * n_lin * n_lin
    return out


def covariance_from_acc(acc):
    """covariance = (1/s) k + (1/(s*s)) u u^T (reference src/scan_matcher_ndt.cpp:146)."""
    k = np.array([[acc[0], acc[1], acc[2]], [acc[1], acc[3], acc[4]], [acc[2], acc[4], acc[5]]])
    u = np.array(acc[6:9])
    s = acc[9]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_s = np.float64(1.0) / np.float64(s)
        inv_s2 = np.float64(1.0) / (np.float64(s) * np.float64(s))
        return inv_s * k + np.outer(inv_s2 * u, u)


def decode_index(best_index, n_lin):
    """flat index -> (i_theta, i_x, i_y), the reference's loop nesting."""
    per_th = n_lin * n_lin
    ith, rem = divmod(best_index, per_th)
    return ith, rem // n_lin, rem % n_lin


def allreduce_rows(row, rank, world, dist, device=None):
    """Exchange one row per rank with a single all-reduce(sum).

    `row` is a 1-D float64 torch tensor (this rank's record, already on the
    device the backend wants).  Returns the [world, len(row)] table."""
    import torch
    table = torch.zeros((world, row.numel()), dtype=torch.float64,
                        device=row.device if device is None else device)
    table[rank].copy_(row)
    if world > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    return table


def finish_particle_statistics(stats_table, weights_local, particles_local, cov22_prev=0.0):
    """updateStatistics (reference src/particle_filter.cpp:163-218) from the all-reduced
    moment sums.  stats_table[world][8]: un-normalised sums per rank.  Returns (normalised
    local weights, mean[3], cov[3,3], theta_share): cov is complete but for cov[2, 2], which
    holds cov22_prev -- the reference never zeroes it (:216) -- and still lacks the second
    pass (:213-217); theta_share is this rank's part of that pass, sum w d^2 over its own
    particles, d = shortest_angular_distance(theta_i, mean theta).  No collective here: the
    share travels with the weights, and combine_theta_shares() adds the ranks' shares."""
    st = np.asarray(stats_table, dtype=np.float64).reshape(-1, POSE_STATS).sum(axis=0)
    sum_w = st[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.asarray(weights_local, dtype=np.float64) / sum_w
        mean_x, mean_y = st[1] / sum_w, st[2] / sum_w
        mean_th = math.atan2(st[4] / sum_w, st[3] / sum_w)
        cov = np.zeros((3, 3))
        cov[0, 0] = st[5] / sum_w - mean_x * mean_x
        cov[0, 1] = cov[1, 0] = st[6] / sum_w - mean_x * mean_y
        cov[1, 1] = st[7] / sum_w - mean_y * mean_y
    th = np.asarray(particles_local, dtype=np.float64).reshape(-1, 3)[:, 2]
    # angles::shortest_angular_distance(theta_i, mean_theta)
    r = np.fmod((mean_th - th) + math.pi, 2.0 * math.pi)
    d = np.where(r <= 0.0, r + math.pi, r - math.pi)
    share = float(np.sum(w * d * d))
    cov[2, 2] = cov22_prev
    return w, np.array([mean_x, mean_y, mean_th]), cov, share


def combine_theta_shares(cov, shares):
    """cov(2,2) += the ranks' theta-variance shares, added in rank order (every consumer the
    same bits).  `shares`: one value per rank, in rank order.  Returns cov."""
    total = 0.0
    for v in shares:
        total += float(v)
    cov = np.array(cov, dtype=np.float64)
    cov[2, 2] += total
    return cov
