// bsx_step_per_call.hip -- instantiates the per-call step kernels (one launch per step(): bsx_step_discrete / _continuous / *_range): see bsx_step_instances.h.
// Same flags as bsx_kernels.hip (build.py): -ffp-contract=off is load-bearing.
#ifndef BSX_VARIANT            // (a diagnostic variant build is one translation unit: bsx_kernels.hip carries every instance)
#include "bsx_config.h"
#include "bsx_state.h"
#include "bsx_rng.h"
#include "bsx_geometry.h"
#include "bsx_instinct.h"
#include "bsx_step_kernel.h"
#define BSX_INST_KW
#define BSX_INST_PER_CALL
#include "bsx_step_instances.h"
#endif
