"""Round-4 fuzz campaign on the final code: the randomised parity tests of tests/test_gpu_fuzz.py
over many more seeds than the suite runs -- the general case (every kernel variant against the
oracle, now with the WINNER asserted on every case: near-ties are settled on the host), the round-4
paths (multi-device matcher, host single-pose scoring, near-best list) and the large lattices.
    python experiments/fuzz_r04.py <which: general|multi|large> <first seed> <count>"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_fuzz as F  # noqa: E402

which, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fn = {"general": F.test_random_case, "multi": F.test_random_case_multi_device_and_host_paths,
      "large": F.test_random_large_lattice}[which]
bad = []
t0 = time.time()
done = 0
for seed in range(first, first + count):
    try:
        fn(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:300]))
    done += 1
print("%s seeds %d..%d: %d run, %d failures, %.0f s" % (which, first, first + done - 1, done, len(bad), time.time() - t0))
for b in bad[:10]:
    print(b)
