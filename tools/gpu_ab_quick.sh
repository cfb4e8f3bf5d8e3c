#!/bin/bash
# this tree's library against reina_model_amd/csrc/variants/libreina_head.so (whatever the build host put there), same box: per-day
# kernel times of the default year's first wave
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-abq}; mkdir -p $OUT; cd $R
for which in head new head new; do
  if [ $which = head ]; then export REINA_HIP_LIB=$R/reina_model_amd/csrc/variants/libreina_head.so; else unset REINA_HIP_LIB; fi
  timeout 900 python tools/day_modes.py ${2:-100000000} ${3:-130} auto 2>&1 | grep -E "^# (mean|max)" | sed "s/^/$which ${2:-100000000} /"
done | tee $OUT/${TAG}_ab.txt
