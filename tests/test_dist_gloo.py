"""The N > 1 path on CPU: world_size-2 gloo processes shard the theta axis /
the particle set, exchange their records with the single all-reduce of
ndt_2d_amd.dist and must reproduce the unsharded oracle result.  The per-rank
device step is played by the oracle (there is no GPU here); what is under test
is the sharding, the exchange and the combination rules."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _share_record(scores, dth, dlin, th_first, th_stride, th_count):
    """What one GPU's reduction produces for its share of the theta steps
    (th_first, th_first + th_stride, ...): best_index is the flat index in the WHOLE lattice."""
    n_lin = len(dlin)
    per = n_lin * n_lin
    steps = th_first + th_stride * np.arange(th_count)
    sl = scores.reshape(len(dth), per)[steps].reshape(-1)
    rec = np.zeros(12)
    rec[1] = -1.0
    if len(sl) and sl.min() < 0:
        i = int(np.argmin(sl))  # first occurrence of the minimum
        rec[0], rec[1] = sl[i], steps[i // per] * per + i % per
    th = np.repeat(dth[steps], per)
    dx = np.tile(np.repeat(dlin, n_lin), th_count)
    dy = np.tile(dlin, th_count * n_lin)
    x = np.stack([dx, dy, th])
    terms = [x[0] * x[0], x[0] * x[1], x[0] * x[2], x[1] * x[1], x[1] * x[2], x[2] * x[2],
             x[0], x[1], x[2], np.ones_like(dx)]
    rec[2:] = [float(np.sum(t * sl)) for t in terms]
    return rec


_N_COLLECTIVES = [0]


def _count_collectives(dist):
    """Collectives issued so far in this process (all_reduce is wrapped once, on first use)."""
    if not getattr(dist, "_ndt2d_counted", False):
        real = dist.all_reduce

        def counted(*a, **k):
            _N_COLLECTIVES[0] += 1
            return real(*a, **k)
        dist.all_reduce = counted
        dist._ndt2d_counted = True
    return _N_COLLECTIVES[0]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist

    import oracle_lib as O
    from ndt_2d_amd import dist as shard
    from ndt_2d_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    g = np.load(os.path.join(HERE, "golden", "cfg1_match.npz"))
    p = json.loads(str(g["params_json"]))
    dth = O.search_offsets(p["search_angular_size"], p["search_angular_resolution"])
    dlin = O.search_offsets(p["search_linear_size"], p["search_linear_resolution"])
    row = torch.from_numpy(_share_record(g["scores"], dth, dlin,
                                         *shard.shard_strided(len(dth), rank, world)))
    table = shard.allreduce_rows(row, rank, world, dist).numpy()
    best_score, best_index, acc = shard.combine_match_records(table)
    cov = shard.covariance_from_acc(acc)

    # particles
    gp = np.load(os.path.join(HERE, "golden", "cfg3_poses256.npz"))
    parts, w_raw = gp["particles"], gp["weights_raw"]
    pb, pe = shard.shard_range(len(parts), rank, world)
    pl, wl = parts[pb:pe], w_raw[pb:pe]
    st = np.array([wl.sum(), (wl * pl[:, 0]).sum(), (wl * pl[:, 1]).sum(),
                   (wl * np.cos(pl[:, 2])).sum(), (wl * np.sin(pl[:, 2])).sum(),
                   (wl * pl[:, 0] * pl[:, 0]).sum(), (wl * pl[:, 0] * pl[:, 1]).sum(),
                   (wl * pl[:, 1] * pl[:, 1]).sum()])
    # ONE collective per particle step: the [world, 8] table.  The theta variance
    # share of this rank leaves with its weights (here: into the rank's file)
    n_before = _count_collectives(dist)
    stats = shard.allreduce_rows(torch.from_numpy(st), rank, world, dist).numpy()
    w, mean, pcov, share = shard.finish_particle_statistics(stats, wl, pl, 0.0)
    assert _count_collectives(dist) == n_before + 1

    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), table=table, best_score=best_score,
             best_index=-1 if best_index is None else best_index, acc=acc, cov=cov,
             w=w, mean=mean, pcov=pcov, share=share, pb=pb, pe=pe)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_reproduces_the_unsharded_result(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(HERE, "golden", "cfg1_match.npz"))
    gp = np.load(os.path.join(HERE, "golden", "cfg3_poses256.npz"))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    # every rank holds the same table and reaches the same decision
    assert np.array_equal(outs[0]["table"], outs[1]["table"])
    for o in outs:
        assert int(o["best_index"]) == int(g["best_index"])
        assert float(o["best_score"]) / 720 == float(g["score"])
        assert np.allclose(o["cov"], g["covariance"], rtol=1e-10, atol=0)
        assert np.allclose(o["mean"], gp["mean"], rtol=1e-10, atol=1e-13)
        # whoever collects the weights adds the ranks' theta shares, in rank order
        from ndt_2d_amd import dist as shard
        pcov = shard.combine_theta_shares(o["pcov"], [float(q["share"]) for q in outs])
        assert o["pcov"][2, 2] == 0.0 and pcov[2, 2] > 0.0
        assert np.allclose(pcov, gp["cov"], rtol=1e-9, atol=1e-13)
    w = np.concatenate([o["w"] for o in outs])
    assert np.allclose(w, gp["weights"], rtol=1e-12, atol=0)
