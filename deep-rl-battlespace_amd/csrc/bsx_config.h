// bsx_config.h -- what every translation unit of the step() path starts from: the HIP / C headers, the C ABI, the actor's shared code, and
// the build's switches (the product constants, or csrc/bsx_diag.h in a diagnostic VARIANT build).
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "battlespace_hip.h"
#include "bsx_actor_core.h"

// Diagnostic builds (tools/build_variant.py compiles this file with -DBSX_VARIANT) are the measuring instrument: timing-only ablations
// (DIAG bits; results are WRONG with any bit set) and in-kernel phase stamps, both in bsx_diag.h.  The product build sees the constants
// below: no ablation, stamps compile to nothing, bsx_build_flags() == 0.
#ifdef BSX_VARIANT
#include "bsx_diag.h"
#else
constexpr unsigned DIAG = 0;
constexpr int BUILD_FLAGS = 0;
#define STAMP(i) do { } while (0)
#define STAMP_HWID() do { } while (0)
#define FSTAMP(i) do { } while (0)
#define PSTAMP(i) do { } while (0)
#endif
// Product constants that earlier rounds kept as experiment switches (what was measured against them: profiles/r04_experiments.json,
// profiles/r05_experiments.json, profiles/HISTORY_r05.md; the rejected forms live in the history, not in this tree):
// multi-tick kernels that recompute lane-derived LDS / row addresses per tick instead of carrying them across the tick loop: bit n = the n-v-n
// kernels with int32 actions, bit 8 + n = those with score rows or continuous actions (3v3, 4v4; 2v2 ... 4v4 -- bsx_step_kernel.h)
constexpr int OPAQUE_MULTI_MASK = 0x1C18;
// atan2's coefficients come from constant memory for up to this many lockstep evaluations, from literals beyond (bsx_geometry.h)
constexpr int ATAN_TABLE_MAX_K = 2;
