#!/usr/bin/env python3
"""BASELINE config 3 WITH the volume constraint row (f = 320 icosphere, surface + Helfrich bending, CG): the
`config3_volume_row` object of bench.py's line on its own.  usage: python3 tools/bench_config3v.py [STEPS=200]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = bench.secondary_config("config3_volume_row", 320, ["surface", "bending"], ["volume"], "conjugate_gradient",
                             volume_row=True, step_size=1e-6, steps=steps, warmup=30, device=0)
print(json.dumps(out))
