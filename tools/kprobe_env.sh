#!/bin/bash
# kprobe under a list of environment settings: tools/kprobe_env.sh "MS_PERSIST=0" "MS_PERSIST=1" ...
for e in "$@"; do
  env $e python tools/kprobe.py --tag "$e" 2>&1 | grep -v amdgpu.ids
done
