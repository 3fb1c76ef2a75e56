import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util, torch
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_obstacle_map_gpu.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    try:
        t.test_random_hole_patterns_against_the_oracle(torch.device("cuda:0"), seed)
    except AssertionError as e:
        bad += 1
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print("seed", seed, "FAILED at line", tb.lineno, "|", str(e)[:200].replace("\n", " "))
    except Exception as e:
        bad += 1
        print("seed", seed, "RAISED", type(e).__name__, str(e)[:200])
print(f"hole seeds {a}..{b - 1}: {bad} failed")
