#!/bin/bash
# Build and run the traffic-counter calibration under rocprofv3 (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as the guide
# prescribes); summary -> gpurun_out/pmc_calib/ (tools/pmc_calib_summary.py turns it into profiles/r04_pmc_calib.json)
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$HERE/gather_calib.hip" -o /tmp/gather_calib || exit 1
/tmp/gather_calib > $OUT/expected.json || exit 1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- /tmp/gather_calib > /dev/null 2> $OUT/$C.err || echo "$C failed" >> $OUT/failed.txt
done
find $OUT -name "*counter_collection.csv" | head
