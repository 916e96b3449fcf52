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

// Diagnostic builds (tools/build_variant.py compiles this file with -DBSX_VARIANT): timing-only ablations (DIAG bits; results are
// WRONG with any bit set) and in-kernel phase stamps live in bsx_diag.h.  The product build sees the constants below: no
// ablation, stamps compile to nothing, bsx_build_flags() == 0.
#ifdef BSX_VARIANT
#include "bsx_diag.h"
#else
constexpr unsigned DIAG = 0;
constexpr int BUILD_FLAGS = 0;
constexpr int OBS_FORM = 0;
constexpr bool X_CORNERS_ALL = false;
constexpr int X_DEPHASE = 0;
constexpr int X_MIN_WAVES = 1;
constexpr int X_OBS_PLAIN_FROM = 99;
constexpr int X_ATAN_TABLE_MAX_K = 2;
constexpr int X_OPAQUE_MULTI_MASK = 0x1C18;      // multi-tick kernels that recompute lane-derived addresses per tick (bsx_step_kernel.h)
constexpr int X_PAD_SALU = 0, X_PAD_VALU = 0, X_DEPHASE_SLOT = 0;
constexpr int X_PRIO_BY_SLOT = 0, X_PRIO_LATE = 0, X_PRIO_LATE_LEVEL = 1;
// The two-wave 1v1 kernels (bsx_step_split.h).  Multi-tick launches of up to 65 536 games: a GAME wave + an OUTPUTS wave per 64 agents (two forms, by size).
// Per-call launches of up to 114 688 games: form 4, a wave for everything but the observation geometry + a GEOMETRY wave fed with the
// post-move poses.  In both the first wave -- whose chain sets the pace -- runs at s_setprio 1: without that the per-call forms lose to
// the one-wave kernel.
constexpr bool X_SPLIT_MANY = true;
constexpr int X_SPLIT_MANY_FORM2_FROM = 32768;      // multi-tick launches of MORE games than this (two workgroups on some SIMD) take form 2 of the two-wave kernel
constexpr int X_SPLIT_FORM = 4, X_SPLIT_PRIO = 1;
constexpr bool X_SPLIT_OWN_LOADS = false;
constexpr int X_SPLIT_GEOM_PRIO = 0;
#define STAMP(i) do { } while (0)
#define STAMP_HWID() do { } while (0)
#define FSTAMP(i) do { } while (0)
#define PSTAMP(i) do { } while (0)
#endif
