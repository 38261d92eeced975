#!/bin/bash
# same-box A/B of the autopatched stage-A step under environment toggles:
#   gpurun -- 'bash tools/ab_autopatch_host.sh "GSR_AUTOPATCH_EXTRAS=0 GSR_AUTOPATCH_LOSS_REPORT=0" "" [rounds] [args of autopatch_host_profile.py]'
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq 1 $R); do for o in "$A" "$B"; do
  echo -n "[$o] "; env $o python tools/autopatch_host_profile.py --top 1 --steps 600 "$@" 2>/dev/null | grep "wall per step"
done; done
