#!/bin/bash
# libraries side by side on one box, alternating: `new` = this tree's, any other name = reina_model_amd/csrc/variants/libreina_<name>.so
# (compiled on the build host: the last commit's sources, or a variant of this tree's) -- per-day kernel times of the default year's
# first wave.  usage: gpu_ab_quick.sh <tag> <agents> <days> <name> [<name> ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-abq}; N=${2:-100000000}; D=${3:-130}; shift 3; mkdir -p $OUT; cd $R
for which in ${@:-head new head new}; do
  if [ $which = new ]; then unset REINA_HIP_LIB; else export REINA_HIP_LIB=$R/reina_model_amd/csrc/variants/libreina_$which.so; fi
  timeout 900 python tools/day_modes.py $N $D auto 2>&1 | grep -E "^# (mean|max)" | sed "s/^/$which $N /"
done | tee $OUT/${TAG}_ab.txt
