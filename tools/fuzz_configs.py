"""Extended run of tests/test_gpu_random_configs.py: the same seeded draws over a much longer case range (one-off hunting; the suite keeps 112 cases).
   python3 tools/fuzz_configs.py <first case> <count>      -- prints one line per failing case with its exception"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import test_gpu_random_configs as R
from oracle import oracle
oracle.build()
first, count = int(sys.argv[1]), int(sys.argv[2])
if "--wide" in sys.argv:          # broader menus than the suite's: very wide and very narrow Gaussians, larger batches
    R.R_MENU = [1.5, 3.0, 30.0, 100.0, 400.0]
    R.S_FACTOR_MENU = [1.02, 1.5, 10.0]
    R.BATCH_MENU = [1, 129, 300, 512, 777, 1024]
    R.GPV_S_MENU = [3.0, 8.0, 240.0, 5000.0, 60000.0]
bad = 0; skipped = 0
t0 = time.time()
for fn in (R.test_perturbation_random_configuration, R.test_gpv_random_configuration, R.test_ring_random_configuration):
    for case in range(first, first + count):
        try:
            fn(oracle, case)
        except pytest.skip.Exception:
            skipped += 1
        except BaseException as ex:      # noqa
            bad += 1
            print(f"FAIL {fn.__name__}[{case}]: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
print(f"done: 3 x {count} cases from {first}, {bad} failures, {skipped} skipped, {time.time() - t0:.0f} s", flush=True)
