#!/bin/bash
# Same-box A/B of library builds under gpurun_libs/ on the WHOLE bench line (other workloads, stage-A legs, drop-in legs):
#   gpurun -- 'bash tools/ab_full.sh "old F"'   -> gpurun_out/ab_full_<name>.json + a one-line digest each
D=3dgs_hierarchical_training_amd/csrc
cp $D/libgsr_hip.so /tmp/cur.so
mkdir -p gpurun_out
for w in $1; do
  cp gpurun_libs/lib_$w.so $D/libgsr_hip.so
  timeout 900 python bench.py --no-cpu-baseline > gpurun_out/ab_full_$w.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_full_$w.json").read().strip().splitlines()[-1])
ow = d.get("other_workloads", {})
print("$w", round(d["value"], 1), {k: (round(v.get("ms_per_step", v.get("image_iteration_ms", 0)), 4) if isinstance(v, dict) else v) for k, v in ow.items()},
      "dropin_autopatch", round(d.get("dropin_autopatch", {}).get("ms_per_step", 0), 4))
PY
done
cp /tmp/cur.so $D/libgsr_hip.so
