"""More seeds of tests/test_gpu_fuzz.py::test_fuzz_rows_around_the_path_against_oracle (N2 projection, N3 SLIC, N4 stereo refinement
on random sizes, parameters and contents).  Run on the GPU box: python tools/fuzz_rows.py [first_seed] [n]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import test_gpu_fuzz as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + n):
    try:
        F.test_fuzz_rows_around_the_path_against_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("SEED", seed, str(e)[:300])
print(f"done, {n} cases (seeds {first}..{first + n - 1}), failures: {bad}")
