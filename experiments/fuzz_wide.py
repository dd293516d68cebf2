"""One-off: the wide-window randomised parity test over many seeds (seed % 10 picks the
resolution class of tests/test_gpu_fuzz.py::test_random_wide_window, incl. the 8 x 8 block maps)."""
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
        F.test_random_wide_window(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print("wide-window seeds %d..%d: %d failures" % (first, first + count - 1, len(bad)))
for b in bad[:10]:
    print(b)
