#!/usr/bin/env python3
"""BASELINE config 2 alone (f = 81 icosphere, surface + volume Lagrange row, gradient descent): steps/s with and without
the resident step kernel.  usage: python3 tools/bench_config2.py [STEPS=400]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
out = bench.secondary_config("config2", 81, ["surface"], ["volume"], "gradient_descent", volume_row=True,
                             step_size=1e-3, steps=steps, warmup=30, device=0)
print(json.dumps({k: out[k] for k in ("value", "ms_per_step", "steps_accepted", "line_search_trials", "resident_step_kernel",
                                      "kernels_avg_us") if k in out}))
