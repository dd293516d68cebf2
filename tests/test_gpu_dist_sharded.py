"""dist.match_scan_sharded: the whole multi-rank matchScan (round-robin theta shares,
one all-reduce of the record table, index-aware combination) with real kernels.  The
GPU box has one GPU, so the ranks share it and exchange through gloo; with RCCL the
only difference is where the all-reduce runs (bench.py covers that path)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from ndt_2d_amd import ScanMatcherNDT, synth
    from ndt_2d_amd import dist as shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = ScanMatcherNDT(0)
    m.initialize("global_scan_matcher", **synth.matcher_params(1, search_angular_resolution=0.002))
    m.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)
    res = shard.match_scan_sharded(m, guess, pts, rank, world, dist)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), score=res["score"], pose=res["pose"],
             covariance=res["covariance"], best_index=res["best_index"],
             n_candidates=res["n_candidates"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_match_scan_equals_single_gpu(tmp_path, world):
    from ndt_2d_amd import ScanMatcherNDT, synth
    ctx = mp.get_context("spawn")
    port = 29600 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    m = ScanMatcherNDT(0)
    m.initialize("single", **synth.matcher_params(1, search_angular_resolution=0.002))
    m.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)
    want = m.matchScan(guess, pts)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(got["best_index"]) == want["best_index"]
        assert int(got["n_candidates"]) == want["n_candidates"]
        assert np.array_equal(got["pose"], want["pose"])
        assert abs(float(got["score"]) - want["score"]) < 1e-12
        assert np.allclose(got["covariance"], want["covariance"], rtol=1e-9, atol=0)
