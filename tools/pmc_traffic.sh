#!/bin/bash
# HBM traffic of the traversal kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits),
# --kernel-trace only, as MI355X_MICROARCH.md §HBM prescribes.
set -u
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/traffic_$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  if [ -n "${NRT_PMC_ONLY:-}" ] && [[ " $NRT_PMC_ONLY " != *" $(echo $C | cut -d" " -f1) "* ]]; then continue; fi
  N=$(echo $C | tr ' ' '_' | cut -c1-20)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -o p -- python tools/pmc_traffic.py > $OUT/$N.log 2>&1 || echo "pass $C failed"
done
python tools/pmc_traffic_summary.py $OUT | tee $OUT/summary.json
