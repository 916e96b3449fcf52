// bsx_step_multi_tick.hip -- instantiates the multi-tick step kernels (bsx_step_many_*: T calls in one launch): see bsx_step_instances.h.
// Same flags as bsx_kernels.hip (build.py): -ffp-contract=off is load-bearing.
#ifndef BSX_VARIANT            // (a diagnostic variant build is one translation unit: bsx_kernels.hip carries every instance)
#include "bsx_config.h"
#include "bsx_state.h"
#include "bsx_rng.h"
#include "bsx_geometry.h"
#include "bsx_instinct.h"
#include "bsx_step_kernel.h"
#define BSX_INST_KW
#define BSX_INST_MULTI_TICK
#include "bsx_step_instances.h"
#endif
