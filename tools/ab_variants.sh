#!/bin/bash
# A/B a set of alternative library builds (build_variants/lib_*.so, e.g. different occupancy
# attributes) on the headline bench: prints steps/s and per-kernel HIP-event averages.
for so in "" build_variants/lib_*.so; do
  name=${so:-default}
  MEMBRANE_HIP_LIB=${so:+$PWD/$so} python bench.py --cpu-steps 0 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$name FAILED"; tail -n 3 gpurun_out/ab.err; continue; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(sys.argv[1], round(d["value"]), round(d.get("deterministic_mode", {}).get("value", 0)),
      {n: round(v["avg_us"], 1) for n, v in d["kernels"].items()})
PY
done
