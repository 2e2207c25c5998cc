#!/bin/bash
# kprobe over the default library and every build_variants/lib_*.so
python tools/kprobe.py "$@" --tag default
for so in build_variants/lib_*.so; do
  MEMBRANE_HIP_LIB=$PWD/$so python tools/kprobe.py "$@" || echo "$so FAILED"
done
