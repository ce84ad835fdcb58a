#!/bin/bash
# (trav_min, refill_min) of the single-pass scene walk on 10 000 / 100 000 instances (tools/scene_probe.py)
cd "$(dirname "$0")/.." || exit 1
for cfg in ${CFGS:-"24 16" "24 24" "24 32" "32 16" "32 24" "32 32" "40 24" "48 24" "32 8" "20 20"}; do
  set -- $cfg
  echo "== trav_min $1 refill_min $2"
  NRT_SCENE_WALK_TRAV=$1 NRT_SCENE_WALK_REFILL=$2 NRT_SCENE_CAND=${CAND:-1} NRT_SCENE_CAND_BUSY=${BUSY:-64} timeout 100 python tools/scene_probe.py 10000 100000 --no-fixture 2>&1 | grep -E "instances" | sed -E 's/of a 2208-triangle mesh: commit [0-9.]+ ms, //' | cut -c1-90
done
