"""'Select the same best pose' when two candidates score within rounding of each other.

The kernels' scores differ from the CPU reference's in the last bits (device exp vs libm),
so a near-tie could come out in the other order.  Every (score, index) merge on the device
marks its winner when the loser was within 1e-11 (relative) of it (ndt2d_device_fn.h, merge_best);
a marked result makes matchScan list the candidates that close to the best
(ndt2d_match_near_best), rescore them on the host with the reference's arithmetic and
apply its rule: strict `<` in visiting order (reference src/scan_matcher_ndt.cpp:128-134).
The lattices here are built to tie: one beam aimed at the mean of a symmetric cell, offsets
placed symmetrically around it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# a symmetric 9-point cell around (2, 2) in a 4 m cell grid: mean exactly (2, 2)
CELL = np.array([[2.0, 2.0], [3.0, 2.0], [1.0, 2.0], [2.0, 3.0], [2.0, 1.0],
                 [2.5, 2.5], [1.5, 1.5], [2.5, 1.5], [1.5, 2.5]])
PARAMS = dict(ndt_resolution=4.0, range_max=8.0, laser_max_beams=100,
              search_linear_size=0.1875, search_linear_resolution=0.125,      # offsets -0.1875, -0.0625, 0.0625
              search_angular_size=0.001, search_angular_resolution=0.002)     # one step: dth = -0.001
SCAN_POSE = (0.0, 0.0, 0.001)                                                  # theta + dth == 0.0 exactly


def _matchers(variant=None):
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT
    gpu = ScanMatcherNDT(0)
    gpu.initialize("ties", **PARAMS)
    gpu.addScans([((0.0, 0.0, 0.0), CELL)])
    if variant:
        gpu.set_variant(variant)
    ref = O.ScanMatcherNDT()
    ref.initialize(**PARAMS)
    ref.addScans([((0.0, 0.0, 0.0), CELL)])
    return gpu, ref


@pytest.mark.parametrize("variant", [None, "lane", "wave"])
def test_exact_tie_goes_to_the_first_candidate_visited(variant):
    gpu, ref = _matchers(variant)
    beam = np.array([[2.0, 2.0]])
    exp = ref.matchScan(SCAN_POSE, beam, want_scores=True)
    s = np.sort(exp["scores"])
    assert s[0] == s[1] < 0.0                       # the lattice does tie at the top (on the CPU)
    got = gpu.matchScan(SCAN_POSE, beam, want_scores=True)
    assert got["n_candidates"] == exp["n_candidates"] == 9
    assert got["best_index"] == exp["best_index"] == int(np.argmin(exp["scores"]))   # argmin: first minimum
    assert np.array_equal(got["pose"], exp["pose"])
    marked, changed, truncated = gpu.adjudication_stats()
    assert marked == 1 and truncated == 0
    # settled on the host: the returned score is the reference's own bits
    assert got["score"] == exp["score"]
    # the candidates within 1e-9 of the best, as the device lists them
    gpu.prepare_search(SCAN_POSE, beam)
    near, n = gpu.match_near_best(0, 1)
    want = [int(i) for i in np.flatnonzero(exp["scores"] <= exp["scores"].min() * (1.0 - 2.0 ** -36))]
    assert n == len(near) == len(want) and near == want


def test_near_ties_within_rounding_are_settled_as_the_reference_settles_them():
    """Beam end points a few ulps off the symmetric position: the top candidates differ by
    ulps.  With the adjudication the winner is the oracle's in every case; without it the
    device's own order decides (counted, for the record)."""
    gpu, ref = _matchers()
    raw, _ = _matchers()
    raw.set_adjudication(False)
    differ_raw = 0
    cases = 0
    for kx in range(-6, 7):
        for ky in (-3, 0, 2, 5):
            x = 2.0 + kx * np.spacing(2.0)
            y = 2.0 + ky * np.spacing(2.0)
            beam = np.array([[x, y]])
            exp = ref.matchScan(SCAN_POSE, beam, want_scores=True)
            got = gpu.matchScan(SCAN_POSE, beam)
            s = np.sort(exp["scores"])
            assert s[1] - s[0] < 1e-12               # (a near-tie, on the CPU)
            assert got["best_index"] == exp["best_index"], (kx, ky)
            assert np.array_equal(got["pose"], exp["pose"])
            assert got["score"] == exp["score"]
            differ_raw += raw.matchScan(SCAN_POSE, beam)["best_index"] != exp["best_index"]
            cases += 1
    marked, changed, _ = gpu.adjudication_stats()
    assert marked == cases
    assert changed >= differ_raw - 0      # every case the device alone gets wrong was changed
    print("near-tie cases: %d, device order differs from the CPU's in %d, adjudication changed %d"
          % (cases, differ_raw, changed))


def test_unmarked_results_have_no_candidate_within_the_tolerance():
    """cfg-1: the winner is not marked, and indeed no other candidate is within the tolerance."""
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT, synth
    gpu = ScanMatcherNDT(0)
    gpu.initialize("cfg1", **synth.matcher_params(1))
    gpu.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)
    got = gpu.matchScan(guess, pts, want_scores=True)
    assert gpu.adjudication_stats()[0] == 0
    s = np.sort(got["scores"])
    assert s[1] - s[0] > 1e-9 * abs(s[0])
    n_th, _, _ = gpu.prepare_search(guess, pts)
    near, n = gpu.match_near_best(0, n_th)
    assert n == 1 and near == [got["best_index"]]


def test_near_tie_across_devices_of_a_multi_device_matcher():
    """The tied candidates sit in different theta steps, dealt to different device contexts:
    the combination of the per-device records marks the result and it is settled all the same."""
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT
    p = dict(PARAMS, search_angular_size=0.002, search_angular_resolution=0.002)   # dth = -0.002, 0.0
    # the beam AT the mean and theta steps -0.002 and (about) 0 around scan theta 0.001: the two
    # rotations are mirror images, +-0.001
    gpu = ScanMatcherNDT(device_ids=[0, 0])
    gpu.set_multi_min_units(0)
    gpu.initialize("ties2", **p)
    gpu.addScans([((0.0, 0.0, 0.0), CELL)])
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans([((0.0, 0.0, 0.0), CELL)])
    beam = np.array([[2.0, 2.0]])
    exp = ref.matchScan(SCAN_POSE, beam, want_scores=True)
    got = gpu.matchScan(SCAN_POSE, beam)
    assert gpu.matcher_variant().startswith("multi[2]/host/")
    assert got["n_candidates"] == exp["n_candidates"] == 18
    assert got["best_index"] == exp["best_index"]
    assert got["score"] == exp["score"]
    assert gpu.adjudication_stats()[0] == 1


def test_mirrored_beam_sets_tie_up_to_summation_order():
    """24 beams = 12 offsets around the cell's mean and their mirror images: the candidates c and
    -c then add the SAME 24 likelihoods in a different order, so their scores agree up to the
    rounding of the summation -- on the CPU (one running sum, src/ndt_model.cpp:178-187) and on
    the device (the small-lattice search adds a candidate's beams in chunks), differently.
    Settled on the host, the winner is the reference's every time."""
    gpu, ref = _matchers()
    raw, _ = _matchers()
    raw.set_adjudication(False)
    rng = np.random.default_rng(20261002)
    differ_raw = cases = 0
    for _ in range(60):
        d = rng.uniform(-0.4, 0.4, size=(12, 2))
        d = np.round(d * 4096.0) / 4096.0               # dyadic: mean +- d are exact mirror images
        beams = np.concatenate([2.0 + d, 2.0 - d])
        exp = ref.matchScan(SCAN_POSE, beams, want_scores=True)
        s = np.sort(exp["scores"])
        assert s[1] - s[0] < 1e-11
        got = gpu.matchScan(SCAN_POSE, beams)
        assert got["best_index"] == exp["best_index"]
        assert np.array_equal(got["pose"], exp["pose"])
        assert got["score"] == exp["score"]
        differ_raw += raw.matchScan(SCAN_POSE, beams)["best_index"] != exp["best_index"]
        cases += 1
    marked, changed, _ = gpu.adjudication_stats()
    assert marked == cases and changed == differ_raw
    print("mirrored-beam cases: %d, device order differs from the CPU's in %d (all settled)" % (cases, differ_raw))
    assert differ_raw > 0     # the cases exist: without the adjudication the winner would differ


def test_sharded_search_driven_from_outside_settles_its_near_ties_too():
    """ndt_2d_amd/dist.py::match_scan_sharded (one process per GPU): the combined record carries
    the mark, and every rank settles it on its own (ndt2d_matcher_settle_near_tie)."""
    from ndt_2d_amd import dist as shard
    gpu, ref = _matchers()
    rng = np.random.default_rng(77)
    settled = 0
    for _ in range(12):
        d = np.round(rng.uniform(-0.4, 0.4, size=(12, 2)) * 4096.0) / 4096.0
        beams = np.concatenate([2.0 + d, 2.0 - d])
        exp = ref.matchScan(SCAN_POSE, beams)
        got = shard.match_scan_sharded(gpu, SCAN_POSE, beams, 0, 1, None)
        assert got["near_tie"] is True
        assert got["best_index"] == exp["best_index"] and np.array_equal(got["pose"], exp["pose"])
        assert got["score"] == exp["score"]
        settled += 1
    assert gpu.adjudication_stats()[0] == settled


def test_tiny_scores_are_not_ties_and_plateaus_keep_the_first_candidate():
    """(a) A degenerate map whose every score is of the order 1e-150: absolutely they are all
    'within 1e-9', relatively they are far apart -- no mark, the device's own winner stands (the
    tolerance is relative for this reason; found by experiments/fuzz_r04.py).  (b) A plateau: more
    candidates with exactly the best score than the list holds -- the list then carries the FIRST
    ones in visiting order, and the first one wins."""
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT
    # (a) one tight cluster far from where the beams land
    rng = np.random.default_rng(3)
    cluster = np.array([3.0, 3.0]) + 0.004 * rng.standard_normal((40, 2))
    p = dict(ndt_resolution=0.5, range_max=6.0, laser_max_beams=100, search_linear_size=0.3,
             search_linear_resolution=0.02, search_angular_size=0.02, search_angular_resolution=0.01)
    beams = np.array([[3.1, 3.1], [3.12, 3.05], [3.08, 3.14]])
    gpu = ScanMatcherNDT(0)
    gpu.initialize("tiny", **p)
    gpu.addScans([((0.0, 0.0, 0.0), cluster)])
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans([((0.0, 0.0, 0.0), cluster)])
    exp = ref.matchScan((0.0, 0.0, 0.0), beams, want_scores=True)
    s = exp["scores"]
    assert np.sum((s < 0.0) & (s > -1e-9)) > 256            # hundreds of scores an absolute 1e-9 would call ties
    got = gpu.matchScan((0.0, 0.0, 0.0), beams)
    assert got["best_index"] == exp["best_index"] and np.array_equal(got["pose"], exp["pose"])
    # (b) one beam, a flat distribution and offsets so fine that the exponent does not move:
    gpu2, ref2 = _matchers()
    p2 = dict(PARAMS, search_linear_size=1e-13, search_linear_resolution=1e-15)      # 201 x 201 candidates
    for m in (gpu2, ref2):
        m.initialize(**({"name": "plateau"} if m is gpu2 else {}), **p2)
        m.addScans([((0.0, 0.0, 0.0), CELL)])
    beam = np.array([[2.0, 2.0]])
    exp2 = ref2.matchScan(SCAN_POSE, beam, want_scores=True)
    n_best = int(np.sum(exp2["scores"] == exp2["scores"].min()))
    assert n_best > 256
    got2 = gpu2.matchScan(SCAN_POSE, beam)
    assert got2["best_index"] == exp2["best_index"] == int(np.argmin(exp2["scores"]))
    assert gpu2.adjudication_stats()[2] == 1                 # the list was truncated -- to its head


def test_near_best_over_the_whole_cfg4_lattice():
    """The slow path at the size of the 8-GPU workload: 315.5 M candidates searched in theta slabs
    with every score kept (2.5 GB), one pass over them -- the winner is alone within the tolerance."""
    from ndt_2d_amd import ScanMatcherNDT, synth
    gpu = ScanMatcherNDT(0)
    gpu.initialize("global_scan_matcher", **synth.matcher_params(4))
    gpu.addScans(synth.map_scans(4))
    guess, pts, _ = synth.query_scan(4)
    n_th, n_lin, _ = gpu.prepare_search(guess, pts)
    assert n_th * n_lin * n_lin == 315508257
    near, n = gpu.match_near_best(0, n_th)
    assert n == 1 and near == [80443810]
    # a tolerance wide enough to catch the runners-up of the same basin: ascending, the winner among them
    near, n = gpu.match_near_best(0, n_th, rel=0.05, capacity=64)
    assert n >= 2 and near == sorted(near) and 80443810 in near and len(near) == min(n, 64)
