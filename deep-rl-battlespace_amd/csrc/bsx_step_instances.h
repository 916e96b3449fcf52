// bsx_step_instances.h -- the 76 instantiations of bsx_step_kernel<N, CONT, MULTI, ACTOR, LG, OFF32>, in three groups, one translation unit
// each (they compile side by side): per-call kernels (bsx_step_per_call.hip), multi-tick kernels (bsx_step_multi_tick.hip), fused rollouts
// (bsx_step_rollout.hip); and the four multi-tick instances of the two-wave 1v1 kernel (bsx_step_split.h) with the multi-tick group.  bsx_kernels.hip -- launchers and C ABI -- includes this file with BSX_INST_KW = extern: explicit instantiation
// DECLARATIONS, so that nothing is instantiated there.  A diagnostic variant build (-DBSX_VARIANT) is ONE translation unit: bsx_kernels.hip
// then defines all three groups itself (its stamp buffer is a device global of that unit).
#pragma once

#define BSX_STEP_INST(N, CONT, MULTI, ACTOR, LG, OFF32)                                                                                  \
    BSX_INST_KW template __global__ void bsxk::bsx_step_kernel<N, CONT, MULTI, ACTOR, LG, OFF32>(                                        \
        const int64_t, const uint2* const, const uint2* const, const uint2* const, const void* const, const uint2* const, const uint32_t* const, \
        const int, const bsxk::StepArgs);
// one team size of the per-call / multi-tick families: discrete (int32 or score-vector actions) and continuous, narrow and wide offsets
#define BSX_STEP_FAMILY(N, MULTI)                                                                                                        \
    BSX_STEP_INST(N, false, MULTI, false, false, false) BSX_STEP_INST(N, false, MULTI, false, false, true)                                \
    BSX_STEP_INST(N, false, MULTI, false, true, false) BSX_STEP_INST(N, false, MULTI, false, true, true)                                  \
    BSX_STEP_INST(N, true, MULTI, false, false, false) BSX_STEP_INST(N, true, MULTI, false, false, true)
#define BSX_ROLLOUT_FAMILY(N)                                                                                                            \
    BSX_STEP_INST(N, false, true, true, false, false) BSX_STEP_INST(N, false, true, true, false, true)                                    \
    BSX_STEP_INST(N, true, true, true, false, false) BSX_STEP_INST(N, true, true, true, false, true)

#ifdef BSX_INST_PER_CALL
BSX_STEP_FAMILY(0, false) BSX_STEP_FAMILY(1, false) BSX_STEP_FAMILY(2, false) BSX_STEP_FAMILY(3, false) BSX_STEP_FAMILY(4, false)
#endif
#ifdef BSX_INST_MULTI_TICK
BSX_STEP_FAMILY(0, true) BSX_STEP_FAMILY(1, true) BSX_STEP_FAMILY(2, true) BSX_STEP_FAMILY(3, true) BSX_STEP_FAMILY(4, true)
#endif
#ifdef BSX_INST_ROLLOUT
BSX_ROLLOUT_FAMILY(1) BSX_ROLLOUT_FAMILY(2) BSX_ROLLOUT_FAMILY(3) BSX_ROLLOUT_FAMILY(4)
#endif
#if defined(BSX_INST_SPLIT) || defined(BSX_INST_SPLIT_MANY)
#define BSX_SPLIT_INST(LG, OFF32, MANY, CONT, DRAW)                                                                                      \
    BSX_INST_KW template __global__ void bsxk::bsx_step_split_kernel<LG, OFF32, MANY, CONT, DRAW>(                                       \
        const int64_t, const uint2* const, const uint2* const, const uint2* const, const void* const, const uint2* const, const uint32_t* const, \
        const int, const bsxk::StepArgs);
#endif
#ifdef BSX_INST_SPLIT
// (per call, discrete: with the geometry wave's draw -- launches of up to 98 304 games -- and without it, up to 114 688; continuous: with it)
BSX_SPLIT_INST(false, false, 0, false, true) BSX_SPLIT_INST(false, true, 0, false, true) BSX_SPLIT_INST(true, false, 0, false, true) BSX_SPLIT_INST(true, true, 0, false, true)
BSX_SPLIT_INST(false, false, 0, false, false) BSX_SPLIT_INST(false, true, 0, false, false) BSX_SPLIT_INST(true, false, 0, false, false) BSX_SPLIT_INST(true, true, 0, false, false)
#ifdef BSX_INST_SPLIT_CONT                               // (continuous actions: the per-call form only; see bsx_step_split.h)
BSX_SPLIT_INST(false, false, 0, true, true) BSX_SPLIT_INST(false, true, 0, true, true)
#endif
#endif
#ifdef BSX_INST_SPLIT_MANY
BSX_SPLIT_INST(false, false, 1, false, false) BSX_SPLIT_INST(false, true, 1, false, false) BSX_SPLIT_INST(true, false, 1, false, false) BSX_SPLIT_INST(true, true, 1, false, false)
BSX_SPLIT_INST(false, false, 2, false, false) BSX_SPLIT_INST(false, true, 2, false, false) BSX_SPLIT_INST(true, false, 2, false, false) BSX_SPLIT_INST(true, true, 2, false, false)
#endif
