"""GPU: which parameter seeds let the bin/run_train.sh-size trajectory test run n steps against the fp64 oracle
(a ReLU pre-activation within fp32 rounding of zero takes the other side than in fp64 at some seeds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_timed_path as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for seed in range(11, 31):
    try:
        T.trajectory_case("gmvae", 784, 128, 10, (512,), 64, n, seed)
        print(f"seed {seed}: ok at n = {n}", flush=True)
    except AssertionError as e:
        print(f"seed {seed}: FAILED {str(e)[:100]}", flush=True)
