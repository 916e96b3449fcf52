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
#pragma once

#ifndef BSX_DIAG
#define BSX_DIAG 0
#endif
constexpr unsigned DIAG = BSX_DIAG;

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
