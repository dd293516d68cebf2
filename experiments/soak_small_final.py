"""Round 6: the small-lattice search's final stage reads the blocks' records with sc1 loads and no acquire
fence.  A stale read would hand the reducer the PREVIOUS launch's record for a block -- so the soak alternates
K different scans (every launch's records differ from the launch before it) on the plugin's default lattice and
holds every result (score, pose, covariance, best index) to the bits of that scan's first result, while a
second context keeps the chip's other CUs busy with particle batches (its own thread).
The second leg does the same to the few-pose kernel, whose last block reads the other blocks' scores the same way:
pf_measure of 500 particles (the node's filter) with K alternating particle sets, weights / mean / covariance to
the bits of each set's first result.
    python experiments/soak_small_final.py <seconds per leg> [scans]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndt_2d_amd import ScanMatcherNDT, synth
from ndt_2d_amd.scan_matcher import pf_measure

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = ScanMatcherNDT(0)
m.initialize("local", range_max=4.75)          # the plugin's defaults: 35,280 candidates x 100 beams
m.addScans(synth.map_scans(1))
world = synth.world_of(1)
rng = np.random.default_rng(6)
scans = []
for k in range(K):
    true = np.array([0.3 * rng.uniform(-1, 1), 0.3 * rng.uniform(-1, 1), 0.05 * rng.uniform(-1, 1)])
    guess = true + np.array([0.02 * rng.uniform(-1, 1), 0.02 * rng.uniform(-1, 1), 0.03 * rng.uniform(-1, 1)])
    scans.append((guess, synth.scan(world, true, 600 + k)))

stop = False
def noise():
    n = ScanMatcherNDT(0)
    n.initialize("other", **synth.matcher_params(3))
    n.addScans(synth.map_scans(3))
    _, pts, _ = synth.query_scan(3)
    parts = synth.particles(3, 20000)
    while not stop:
        n.scorePoses(pts, parts)
t = threading.Thread(target=noise)
t.start()

def key(r):
    cov = r["covariance"]
    return (np.float64(r["score"]).tobytes(), np.asarray(r["pose"], dtype=np.float64).tobytes(),
            b"" if cov is None else np.asarray(cov, dtype=np.float64).tobytes(), int(r["best_index"]))
first = [key(m.matchScan(g, p)) for g, p in scans]
assert len(set(first)) == K, "the scans' results must differ for the soak to mean anything"
variant = m.last_variant()
n = 0
bad = 0
t0 = time.time()
while time.time() - t0 < seconds:
    for k, (g, p) in enumerate(scans):
        if key(m.matchScan(g, p)) != first[k]:
            bad += 1
        n += 1
print("%s: %d searches of %d alternating scans in %.0f s beside a second context's particle batches: %d results differ from their scan's first"
      % (variant, n, K, time.time() - t0, bad))

sets = [np.column_stack([0.5 * rng.uniform(-1, 1, 500), 0.5 * rng.uniform(-1, 1, 500), 0.2 * rng.uniform(-1, 1, 500)]) for _ in range(K)]
pts = scans[0][1]
def pkey(r):
    return r[0].tobytes() + r[1].tobytes() + r[2].tobytes()
pfirst = [pkey(pf_measure(m, s_, pts)) for s_ in sets]
assert len(set(pfirst)) == K
pvariant = m.last_variant()
pn = 0
pbad = 0
t0 = time.time()
while time.time() - t0 < seconds:
    for k, s_ in enumerate(sets):
        if pkey(pf_measure(m, s_, pts)) != pfirst[k]:
            pbad += 1
        pn += 1
stop = True
t.join()
print("%s: %d measures of %d alternating 500-particle sets in %.0f s: %d results differ from their set's first"
      % (pvariant, pn, K, time.time() - t0, pbad))
sys.exit(1 if (bad or pbad) else 0)
