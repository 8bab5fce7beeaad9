"""one-off: many more seeds of the frozen-mode fuzz test than the suite carries"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from slimfastq_amd import capi
import test_frozen_tables as T
ctx = capi.Context(0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        T.test_frozen_fuzz_structurally_hostile_inputs(ctx, seed)
        print("seed", seed, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:300].replace("\n", " | "), flush=True)
print("failures:", bad)
