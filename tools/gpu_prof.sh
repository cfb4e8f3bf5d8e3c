#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for k in 1 2 4 5 6 8 9; do
  REINA_PROF_WHAT=$k REINA_HIP_LIB=tools/libreina_prof$k.so python tools/day_prof.py ${1:-100000000} 93 200 2>&1 | grep "^part"
done
