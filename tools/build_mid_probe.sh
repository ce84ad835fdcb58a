#!/bin/bash
# tools/build_mid_probe.sh OUT [mids...] — every case x hand-over size in its own process (tools/build_mid_cases.py)
out=${1:-gpurun_out/build_mid_cases.txt}; shift
mids=${@:-0 512 1024 2048}
: > $out
for c in soup20k_f32 soup200k_f32 soup200k_f64 clustered_f32 plane2400_f32 sphere_f32 plane1m_f32 plane1m_f64 plane10m_f32; do
  for m in $mids; do
    timeout 120 python tools/build_mid_cases.py $c $m >> $out 2>&1 || echo "$c mid $m FAILED rc=$?" >> $out
  done
done
grep -v amdgpu.ids $out
