import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import test_gpu_fuzz as F

class _NoPatch:
    def setenv(self, k, v): os.environ[k] = v

bad = 0
for seed in range(12, 132):
    try:
        F.test_fuzz_against_oracle(seed, _NoPatch())
    except AssertionError as e:
        bad += 1
        print("SEED", seed, str(e)[:300])
print("done, failures:", bad)
