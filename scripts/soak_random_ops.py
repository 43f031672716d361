import sys, subprocess
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
import importlib, types
import numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import _pkg
from oracle import oracle as O
import test_random_ops_gpu as T
vsa = _pkg.vsa
bad = 0
for seed in range(100, 130):
    for metric in ("L2", "IP"):
        try:
            T.test_flat_random_operations(vsa, O, metric, seed)
            T.test_hnsw_random_operations(vsa, O, metric, seed + 1000)
        except AssertionError as e:
            bad += 1
            print("FAIL", metric, seed, str(e)[:300])
print("soak done, failures:", bad)
