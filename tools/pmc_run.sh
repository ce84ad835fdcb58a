#!/bin/bash
# tools/pmc_run.sh TAG — rocprofv3 PMC passes over a short bench run (each counter set in its own pass,
# with --kernel-trace only, as the MI355X guide prescribes).  Outputs under gpurun_out/pmc_TAG/.
set -u
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
CMD="python bench.py --no-cpu-baseline --steps 4 --warmup 1 --builds 1"
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM" \
  "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
find $OUT -name '*.csv' | head -40
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
