"""One-off: the large-lattice randomised parity test over many seeds."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_fuzz as F  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, first + count):
    try:
        F.test_random_large_lattice(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print("large-lattice seeds %d..%d: %d failures" % (first, first + count - 1, len(bad)))
for b in bad[:10]:
    print(b)
