#!/bin/bash
# k_hosp_install: what the round-5 changes are worth, one by one -- diagnostic builds of this tree (built here, on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-iv}; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_NO_BYBIT -o /tmp/lib_nobybit.so reina_model_amd/csrc/reina_hip.hip 2>/dev/null &

wait
for n in 1685983; do
for v in final nobybit; do
  if [ $v = final ]; then L=""; else L=/tmp/lib_$v.so; fi
  REINA_HIP_LIB=$L python tools/day_modes.py $n 365 auto 2>/dev/null | grep "^# mean" | sed "s/^/$n $v /"
  python - <<PY
import json, numpy as np
d=json.load(open('$R/gpurun_out/day_modes_$n.json'))['auto']
v=np.array([r.get('k_hosp_install',0) for r in d]); o=np.array([r.get('k_open',0) for r in d]); print('   k_hosp_install median %.1f p90 %.1f max %.1f | days 85-100 %.1f | k_open median %.1f' % (np.median(v), np.percentile(v,90), v.max(), v[85:100].mean(), np.median(o)))
PY
done; done | tee $OUT/${TAG}_variants.txt
cd /tmp && rm -rf old && tar xf $R/old_tree_r04.tar && cd /tmp/old && python -c "from reina_model_amd import build; build.build(verbose=False)" > /dev/null 2>&1
for k in 1 2; do (cd /tmp/old && python tools/day_modes.py 1685983 365 auto 2>/dev/null | grep '^# mean' | sed 's/^/old /'); (cd $R && python tools/day_modes.py 1685983 365 auto 2>/dev/null | grep '^# mean' | sed 's/^/new /'); done | tee -a $OUT/${TAG}_variants.txt
