"""benchlib — the pieces of bench.py: workload set-up, the timed region, hardware counters, the CPU baseline, the untimed extras,
the multi-GPU launcher helpers and the builder of the ONE JSON line the driver parses (kept small: benchlib/line.py)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Vector lane peak: 256 CUs x 4 SIMD-32 x 2.4 GHz max clock.  A wave64 fp32 VALU instruction occupies its SIMD for 2 cycles
# (MI355X_MICROARCH.md "Wave scheduling"; measured here with tools/ubench/valu_rate.hip -> profiles/r02a_valu_rate.txt:
# 2.8 cycles per v_fma_f32 / v_mul_f32 at 8 waves per SIMD, and twice that for the packed v_pk_* forms and v_max3/v_min3,
# i.e. packing saves issue slots, not lane-cycles).
VALU_LANES_PER_SIMD = 32
VALU_CYCLES_PER_WAVE_INST = 64 // VALU_LANES_PER_SIMD
N_XCD = 8  # GRBM_GUI_ACTIVE arrives summed over the XCDs
CLOCK_GHZ = 2.4
# What the vector L1 (TCP) sustains in tag look-ups per second when every lane of every wave fetches scattered 16-byte
# pieces, measured with tools/ubench/node_fetch.hip under the same counter (3145 M look-ups in 3.59 ms, table resident in
# L2; 896 G/s when resident in L1): profiles/r02g_node_fetch_ubench.txt, r02l_tcp_counter_calibration.txt.
L1_PEAK_GACC_S = 876.0
METRIC = "Mrays/s (primary + 1-bounce) at 1920x1080, 1M-tri mesh; BVH build ms"
