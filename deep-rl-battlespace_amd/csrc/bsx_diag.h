// bsx_diag.h -- what a DIAGNOSTIC build of bsx_kernels.hip sees instead of the product constants.  Included only when the file
// is compiled with -DBSX_VARIANT, which only tools/build_variant.py does (csrc/variants/lib_<name>.so, selected through
// BSX_LIB_PATH; a variant whose results are not the reference's reports that through bsx_build_flags() and the binding refuses
// it without BSX_ALLOW_DIAG=1).  Never part of libbattlespace_hip.so.
//
//   -DBSX_DIAG=<bits>   timing-only ablations, results are WRONG with any bit set: 1 = skip observation math, 2 = skip the bullet
//                       loop, 4 = skip the ordered resolve, 8 = no Philox draw for the shot's jitter, 16 = the per-game counters are not loaded, 32 = no heading-table gather (the move's step is
//                       made up from the heading: what the one dependent load of the common path costs)
//   -DBSX_STAMPS        lane 0 of every wave stores s_memtime at 10 points into a debug buffer (bsx_debug_set_stamps) that nothing
//                       else reads; a stamped build is for reading SHARES, not run time
//   -DBSX_STAMPS -DBSX_STAMPS_FINE   stamps 3..6 move INSIDE the shot phase (after the shot ballot / the Philox draw / sincos / the
//                       first slot fetch); FSTAMP stores from every active lane (it sits in divergent code), the phase stamps 3..6 are off
//   -DBSX_X_OBS=<0|1|2> the store form of the observation rows (below; same results)
#pragma once

#ifndef BSX_DIAG
#define BSX_DIAG 0
#endif
constexpr unsigned DIAG = BSX_DIAG;

// how the observation rows of the per-step kernels leave (round 3's measurement of the 4v4 write side; same results):
//   -DBSX_X_OBS=0  straight from registers, 16-byte non-temporal stores per lane + tail (the product)
//   -DBSX_X_OBS=1  staged in LDS, transposed into fully coalesced 16-byte non-temporal stores
//   -DBSX_X_OBS=2  staged in LDS; 4v4: lane j of a game writes the j-th 64-byte segment of the game's 448 bytes whole
#ifndef BSX_X_OBS
#define BSX_X_OBS 0
#endif
constexpr int OBS_FORM = BSX_X_OBS;

// -DBSX_X_CORNERS_ALL: the corner-form rectangle tests for every team size (the product: 1v1 ... 3v3; same results)
#ifdef BSX_X_CORNERS_ALL
constexpr bool X_CORNERS_ALL = true;
#else
constexpr bool X_CORNERS_ALL = false;
#endif

// -DBSX_X_DEPHASE=<k | 256 | 512 | 1024>: the fused rollout's (| 512: every step kernel's) workgroups of the upper half of the grid
// (| 256: the odd ones) sleep k x 8 128 (| 1024: k x 1 984) cycles before their first tick (same results)
#ifndef BSX_X_DEPHASE
#define BSX_X_DEPHASE 0
#endif
constexpr int X_DEPHASE = BSX_X_DEPHASE;

// -DBSX_X_MIN_WAVES=<k>: the per-step kernels of teams >= 2 are compiled for at least k waves per SIMD (same results)
#ifndef BSX_X_MIN_WAVES
#define BSX_X_MIN_WAVES 1
#endif
constexpr int X_MIN_WAVES = BSX_X_MIN_WAVES;
// -DBSX_X_OBS_PLAIN_FROM=<n>: the observation rows of teams >= n leave as ordinary stores instead of non-temporal ones (same results)
#ifndef BSX_X_OBS_PLAIN_FROM
#define BSX_X_OBS_PLAIN_FROM 99
#endif
constexpr int X_OBS_PLAIN_FROM = BSX_X_OBS_PLAIN_FROM;

// -DBSX_X_ATAN_TABLE_MAX_K=<k>: atan2's coefficients from constant memory for up to k lockstep evaluations (product: 2; same results)
#ifndef BSX_X_ATAN_TABLE_MAX_K
#define BSX_X_ATAN_TABLE_MAX_K 2
#endif
constexpr int X_ATAN_TABLE_MAX_K = BSX_X_ATAN_TABLE_MAX_K;

// -DBSX_X_OPAQUE_MULTI_MASK=<bits>: bit n set = the multi-tick kernels of n-v-n with int32 actions recompute their lane-derived LDS / row
// addresses per tick instead of carrying them across the tick loop, bit 8 + n = those with score rows or continuous actions (product:
// 0x1C18 = 3v3, 4v4; 2v2 ... 4v4; same results)
#ifndef BSX_X_OPAQUE_MULTI_MASK
#define BSX_X_OPAQUE_MULTI_MASK 0x1C18
#endif
constexpr int X_OPAQUE_MULTI_MASK = BSX_X_OPAQUE_MULTI_MASK;

// -DBSX_X_PAD_SALU=<k> / -DBSX_X_PAD_VALU=<k>: k extra s_add_u32 / v_and_or_b32 per wave and call at the top of the geometry phase (same
// results): the marginal cost of ONE instruction of a class in each regime (round 5: is the scalar stream free beside other waves' VALU?)
#ifndef BSX_X_PAD_SALU
#define BSX_X_PAD_SALU 0
#endif
#ifndef BSX_X_PAD_VALU
#define BSX_X_PAD_VALU 0
#endif
constexpr int X_PAD_SALU = BSX_X_PAD_SALU, X_PAD_VALU = BSX_X_PAD_VALU;
// -DBSX_X_DEPHASE_SLOT=<k>: the wave in slot w of its SIMD (HW_ID) sleeps w x k x 64 cycles at kernel entry (same results)
#ifndef BSX_X_DEPHASE_SLOT
#define BSX_X_DEPHASE_SLOT 0
#endif
constexpr int X_DEPHASE_SLOT = BSX_X_DEPHASE_SLOT;

// -DBSX_X_SPLIT=<0|1|2|4>: the form of the two-wave kernel of bsx_step_split.h that 1v1 per-call launches take (same results): 0 = none (the
// one-wave kernel), 1 = a planes wave + a bullets wave per 64 agents, 2 = a wave for everything but the observation geometry + a geometry
// wave that repeats classify and move, 4 (the product) = the same with the geometry wave fed the post-move poses through LDS
#ifndef BSX_X_SPLIT
#define BSX_X_SPLIT 4
#endif
constexpr int X_SPLIT_FORM = BSX_X_SPLIT;
// -DBSX_X_SPLIT_PRIO=<-3..3>: s_setprio of the first wave of the two-wave kernels (negative: of the second wave of the per-call forms
// instead); the product has 1 (same results)
#ifndef BSX_X_SPLIT_PRIO
#define BSX_X_SPLIT_PRIO 1
#endif
constexpr int X_SPLIT_PRIO = BSX_X_SPLIT_PRIO;
// -DBSX_X_SPLIT_OWN_LOADS: per-call two-wave forms, each wave loads its own copy of the records instead of the LDS hand-over (same results)
#ifdef BSX_X_SPLIT_OWN_LOADS
constexpr bool X_SPLIT_OWN_LOADS = true;
#else
constexpr bool X_SPLIT_OWN_LOADS = false;
#endif
// -DBSX_X_SPLIT_GEOM_PRIO=<0..3>: per-call form 4, the geometry wave's priority once the poses are there (same results)
#ifndef BSX_X_SPLIT_GEOM_PRIO
#define BSX_X_SPLIT_GEOM_PRIO 0
#endif
constexpr int X_SPLIT_GEOM_PRIO = BSX_X_SPLIT_GEOM_PRIO;
// -DBSX_X_PRIO_LATE=<k> [-DBSX_X_PRIO_LATE_LEVEL=<1..3>]: one-wave kernels, the workgroups of the last (8 - k) eighths of the grid at a raised
// priority (a launch with more waves than a SIMD holds: its late starters; same results)
#ifndef BSX_X_PRIO_LATE
#define BSX_X_PRIO_LATE 0
#endif
#ifndef BSX_X_PRIO_LATE_LEVEL
#define BSX_X_PRIO_LATE_LEVEL 1
#endif
constexpr int X_PRIO_LATE = BSX_X_PRIO_LATE, X_PRIO_LATE_LEVEL = BSX_X_PRIO_LATE_LEVEL;
// -DBSX_X_PRIO_BY_SLOT=<1|2>: one-wave kernels, s_setprio by the wave's slot on its SIMD (1: slot & 1, 2: slot & 3; same results)
#ifndef BSX_X_PRIO_BY_SLOT
#define BSX_X_PRIO_BY_SLOT 0
#endif
constexpr int X_PRIO_BY_SLOT = BSX_X_PRIO_BY_SLOT;
// -DBSX_X_SPLIT_MANY_FORM2_FROM=<games>: multi-tick launches of MORE games than this take form 2 of the two-wave kernel (the outputs wave only
// takes what the game wave publishes per tick) instead of form 1 (it carries the state too); the product has 32 768; same results
#ifndef BSX_X_SPLIT_MANY_FORM2_FROM
#define BSX_X_SPLIT_MANY_FORM2_FROM 32768
#endif
constexpr int X_SPLIT_MANY_FORM2_FROM = BSX_X_SPLIT_MANY_FORM2_FROM;
// -DBSX_X_NO_SPLIT_MANY: multi-tick 1v1 launches keep the one-wave kernel whatever their size (the product takes the two-wave form of
// bsx_step_split.h up to 65 536 games; same results)
#ifdef BSX_X_NO_SPLIT_MANY
constexpr bool X_SPLIT_MANY = false;
#else
constexpr bool X_SPLIT_MANY = true;
#endif

#ifdef BSX_STAMPS
constexpr int BUILD_FLAGS = int(DIAG & 0xFFu) | 0x100;
__device__ unsigned long long* g_stamps = nullptr;
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (g_stamps && (threadIdx.x & 63) == 0) g_stamps[size_t(stamp_row) * 10 + (i)] = t_;  \
    } while (0)
// slot 9 of a two-wave kernel's row: the wave's HW_ID register (which SIMD of which CU it runs on) instead of a time
#define STAMP_HWID()                                                                               \
    do {                                                                                           \
        uint32_t hw_, xcc_;                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                          \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                        \
        if (g_stamps && (threadIdx.x & 63) == 0) g_stamps[size_t(stamp_row) * 10 + 9] = (unsigned long long)(hw_) | ((unsigned long long)(xcc_) << 32);       \
    } while (0)
// where the stamps go (device buffer of 10 * waves uint64)
extern "C" int bsx_debug_set_stamps(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return int(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)));
}
#else
constexpr int BUILD_FLAGS = int(DIAG & 0xFFu);
#define STAMP(i) do { } while (0)
#define STAMP_HWID() do { } while (0)
#endif

#if defined(BSX_STAMPS) && defined(BSX_STAMPS_FINE)
#define FSTAMP(i)                                                                                  \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (g_stamps) g_stamps[size_t(stamp_row) * 10 + (i)] = t_;                                 \
    } while (0)
#define PSTAMP(i) do { if ((i) < 3 || (i) > 6) STAMP(i); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#define PSTAMP(i) STAMP(i)
#endif
