/* include/nanort_hip_prof.h — profiling entry points of libnanort_hip_prof.so.
 *
 * libnanort_hip_prof.so is libnanort_hip.so built from the same sources with -DNRT_PROF (nanort_amd/csrc/Makefile): the whole
 * C ABI of nanort_hip.h plus the two calls below and the kernel instantiations behind them (loop-occupancy counters, per-wave
 * time stamps).  The product library carries neither; its tunable "debug" ignores bits 32, 64 and 8192.  Used by the scripts under tools/
 * (NRT_USE_PROF_LIB=1 makes nanort_amd.capi load this library).  No reference counterpart.
 */
#ifndef NANORT_HIP_PROF_H_
#define NANORT_HIP_PROF_H_

#include "nanort_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Profiling aid: loop-occupancy counters of the last traversal launched with the tunable "debug" bit 32 set (a separately
 * instantiated, slower kernel).  out16[0..7] = inner-node-phase wave iterations, sum of active lanes, idle lanes at leaf-phase
 * entry, leaf-phase trips, sum of lanes testing a (first) record, refill events, lanes refilled, leaf-phase entries;
 * [8..10] = shader-clock ticks the waves spent refilling / in the inner-node phase / in the leaf phase; [11] = lanes
 * with a second record in a leaf trip; [12..15] = inner iterations, their active lanes, leaf trips and records tested counted only
 * while rays were still being handed out (the steady part of a launch: the rest is its drain).  Returns 0 on success. */
NRT_API int nrtDebugCounters(nrt_ctx *ctx, unsigned long long *out, int capacity /* >= 16 */);
/* Profiling aid: with NRT_DEBUG bit 8192 every wave of a traversal launch records when it started, ran out of rays and
 * finished (100 MHz realtime ticks, 3 x u64 per wave).  Copies up to `cap` records of the most recent launch; returns
 * the number of waves of that launch, or -1 (tools/drain_probe.py). */
NRT_API long nrtDebugWaveClocks(nrt_ctx *ctx, unsigned long long *out, long cap);
/* Profiling aid: loop counters of the last scene query made after nrtSceneSetTunable(scene, "count_loops", 1) (a separately
 * instantiated, slower k_scene_walk).  out16[0] = outer trips of all waves, [1..2] level-change blocks run / lanes served,
 * [3..4] inner-phase trips / lane steps, [5] of those in the top-level tree, [6..7] leaf-phase trips / lanes with a first
 * record, [8] instances opened, [9] top-level leaves reached, [10..12] shader-clock ticks in level changes / the inner phase /
 * the leaf phase.  Returns 0 on success (tools/scene_loop_stats.py). */
NRT_API int nrtSceneDebugCounters(nrt_scene *scene, unsigned long long *out, int capacity /* >= 16 */);

#ifdef __cplusplus
}
#endif

#endif /* NANORT_HIP_PROF_H_ */
