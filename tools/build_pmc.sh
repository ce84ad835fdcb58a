#!/bin/bash
# tools/build_pmc.sh TAG [GRID] — hardware counters of the BUILD kernels (k_bin, k_partition, k_split, k_level_setup,
# k_subtree_rows, ...): rocprofv3 --pmc passes (each with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over four
# builds of Plane(GRID) (default 1000x500 = 1M triangles).  Raw CSVs under gpurun_out/build_pmc_TAG/, the per-kernel summary
# in gpurun_out/build_pmc_TAG/summary.txt (copy it to profiles/).
set -u
TAG=${1:-x}
export NRT_TRACE_GRID=${2:-1000x500}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/build_pmc_$TAG
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1 || true
CMD="python tools/build_trace.py run"
i=0
for SET in \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_GDS" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log | tr '\n' ' ')"
done
python tools/build_pmc_summary.py $OUT | tee $OUT/summary.txt
