#!/bin/bash
# Threshold sweep of the scene kernels on the fixture and the 10 000-instance scene (tools/scene_probe.py), one line per setting.
cd "$(dirname "$0")/.." || exit 1
for cfg in "8 56 1 64" "4 56 1 64" "16 56 1 64" "24 56 1 64" "8 32 1 64" "8 48 1 64" "8 64 1 64" "8 56 8 32" "8 56 16 24" "16 48 8 32" "12 60 1 64" "16 60 4 48"; do
  set -- $cfg
  echo "== trav_min $1 refill_min $2 cand_min $3 cand_busy_max $4"
  NRT_SCENE_TRAV=$1 NRT_SCENE_REFILL=$2 NRT_SCENE_CAND=$3 NRT_SCENE_CAND_BUSY=$4 timeout 100 python tools/scene_probe.py 10000 2>&1 | grep -E "fixture|instances" | sed -E 's/\(.*1920x1080\)//; s/ \| its plane.*//; s/of a 2208-triangle mesh: commit [0-9.]+ ms, //'
done
