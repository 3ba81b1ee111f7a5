"""More seeds of tests/test_gpu_fuzz.py::test_fuzz_16_bit_codes_against_oracle than the suite runs (on the GPU box):
    python tools/fuzz_q16.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["DCMT_Q16_MIN_WAVES"] = "0"
import test_gpu_fuzz as T

class _NoPatch:
    def setenv(self, k, v): os.environ[k] = v

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
for s in range(first, first + count):
    T.test_fuzz_16_bit_codes_against_oracle(s, _NoPatch())
print(f"{count} seeds from {first}: {6 * count} cases bit-identical to the oracle")
