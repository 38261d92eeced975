#!/bin/bash
# PMC passes (one counter set per run, as MI355X_MICROARCH.md prescribes: TCC FETCH_SIZE / WRITE_SIZE separately,
# no tracing domains combined with --pmc).  Usage: tools/pmc_profile.sh <outdir> <kernel-regex> [bench args...]
OUT=$1; REGEX=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-include-regex "$REGEX" --output-format csv -d $OUT/pass$i -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed: $SET" >> $OUT/failed.txt
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
find $OUT -name "*.csv" | head -20
