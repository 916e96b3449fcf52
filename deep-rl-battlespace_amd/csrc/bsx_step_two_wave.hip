// bsx_step_two_wave.hip -- instantiates the two-wave 1v1 step kernels of bsx_step_split.h, the per-call form (bsx_step_discrete / *_range up to
// 114 688 games, bsx_step_continuous / *_range up to 81 920) and the multi-tick form (bsx_step_many_discrete up to 65 536 games): see bsx_step_instances.h.
// The per-call unit's flags (build.py PER_CALL_FLAGS: -amdgpu-sched-strategy=max-ilp is worth 4.5 % on the per-call form, 1 ... 3 % on the
// multi-tick form); -ffp-contract=off is load-bearing.
#ifndef BSX_VARIANT            // (a diagnostic variant build is one translation unit: bsx_kernels.hip carries every instance)
#include "bsx_config.h"
#include "bsx_state.h"
#include "bsx_rng.h"
#include "bsx_geometry.h"
#include "bsx_instinct.h"
#include "bsx_step_kernel.h"
#include "bsx_step_split.h"
#define BSX_INST_KW
#define BSX_INST_SPLIT
#define BSX_INST_SPLIT_CONT
#define BSX_INST_SPLIT_MANY
#include "bsx_step_instances.h"
#endif
