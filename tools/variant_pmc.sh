#!/bin/bash
# Hardware counters of traversal variants selected by tunables, one rocprofv3 pass per counter set (kernel trace only, as
# MI355X_MICROARCH.md prescribes).   tools/variant_pmc.sh OUT.txt CFG "name|ENV=1|dict(...)" ...     (GPU box, repo root)
out=$1; cfg=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > "$out"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; combo=${rest#*|}
  d=gpurun_out/vpmc_${cfg}_$name
  rm -rf $d; mkdir -p $d
  i=0
  for SET in "FETCH_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    env $envs timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $d/p$i -o p$i -- python tools/pmc_child.py $cfg "$combo" > $d/p$i.log 2>&1 || echo "pass $i of $name failed" >> "$out"
  done
  echo "=== $cfg $name ($envs $combo)" >> "$out"
  grep -h "^kernel" $d/p1.log >> "$out"
  python tools/pmc_summary.py $d | grep -A12 "k_traverse" >> "$out"
done
cat "$out"
