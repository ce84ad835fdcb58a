#!/bin/bash
# tools/pmc_stall.sh TAG — where do the traversal kernel's cycles go?  Separate --pmc passes (kernel-trace only).
set -u
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stall_$TAG
mkdir -p $OUT
i=0
for SET in \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum SQ_BUSY_CU_CYCLES" ; do
  i=$((i+1))
  if [ -n "${NRT_PMC_PASSES:-}" ] && [[ " $NRT_PMC_PASSES " != *" $i "* ]]; then continue; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o p -- python tools/pmc_traffic.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_stall_summary.py $OUT | tee $OUT/summary.txt
