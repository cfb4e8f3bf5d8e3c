#!/bin/bash
# install launch of the ordered days at 10^8 agents against the number of walking workgroups (REINA_WALK_DIV buckets each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for d in ${1:-16 32 64 128}; do echo "== REINA_WALK_DIV=$d"; REINA_WALK_DIV=$d python tools/day_modes.py 100000000 130 auto 2>/dev/null | awk 'NR>2 && (NR%8==5 || /^#/)' | tail -12; done
