// GPU box: cycles per instruction on gfx950 for the instruction classes the step kernel is made of, for one wave alone on a SIMD
// and for two waves sharing one (the headline batch runs two).  hipcc --offload-arch=gfx950 -O2 issue_rates.hip -o issue_rates
// Every test is a .rept block of 256 copies between two s_memtime reads (s_memtime counts at 100 MHz; the shader clock is
// read from rocm-smi style constants: we report memtime ticks * (clock / 100 MHz) using the measured s_sleep calibration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#define REPT 256
#define TEST(name, body)                                                                                   \
    __global__ void name(unsigned long long* out, double seedd) {                                          \
        double a = seedd + threadIdx.x, b = seedd * 1.0000001, c = 1.0 - seedd, d = a + 2.0;               \
        unsigned x = threadIdx.x * 2654435761u, y = x ^ 0x9e3779b9u, z = 12345u + threadIdx.x;             \
        unsigned long long m = x;                                                                          \
        unsigned long long t0, t1;                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");   \
        asm volatile(".rept 256\n\t" body "\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(x), "+v"(y), "+v"(z), "+v"(m)::"memory", "vcc", "s20", "s21", "s22", "s23"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                          \
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; } \
        if (a + b + c + d == 123.456 && x + y + z + m == 77) out[8000] = 1;                                \
    }

TEST(k_empty, "")
TEST(k_fma64_dep, "v_fma_f64 %0, %0, %1, %2")
TEST(k_fma64_ind2, "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %3, %3, %1, %2")
TEST(k_add64_dep, "v_add_f64 %0, %0, %1")
TEST(k_mul64_dep, "v_mul_f64 %0, %0, %1")
TEST(k_mov32, "v_mov_b32 %4, 0x12345678")
TEST(k_xor_dep, "v_xor_b32 %4, %4, %5")
TEST(k_xor_ind2, "v_xor_b32 %4, %4, %5\n\tv_xor_b32 %6, %6, %5")
TEST(k_mad64_dep, "v_mad_u64_u32 %7, vcc, %4, %5, %7")
TEST(k_mulhi_dep, "v_mul_hi_u32 %4, %4, %5")
TEST(k_mullo_dep, "v_mul_lo_u32 %4, %4, %5")
TEST(k_mul24_dep, "v_mul_u32_u24 %4, %4, %5")
TEST(k_rcp64_dep, "v_rcp_f64 %0, %0")
TEST(k_rsq64_dep, "v_rsq_f64 %0, %0")
TEST(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %4")
TEST(k_cndmask, "v_cndmask_b32 %4, %4, %5, vcc")
TEST(k_smov, "s_mov_b32 s20, 0x12345678")
TEST(k_smov2_fma, "s_mov_b32 s20, 0x12345678\n\ts_mov_b32 s21, 0x3ff00000\n\tv_fma_f64 %0, %0, %1, s[20:21]")
TEST(k_vmov2_fmac, "v_mov_b32 %4, 0x12345678\n\tv_mov_b32 %5, 0x3ff00000\n\tv_fma_f64 %0, %0, %1, %2")
TEST(k_cmp_sand, "v_cmp_lt_u32 vcc, %4, %5\n\ts_and_b64 s[20:21], vcc, exec\n\tv_cndmask_b32 %6, 0, 1, s[20:21]")
TEST(k_bperm, "ds_bpermute_b32 %4, %5, %4\n\ts_waitcnt lgkmcnt(0)")
TEST(k_readlane, "v_readfirstlane_b32 s20, %4\n\tv_xor_b32 %4, s20, %4")
TEST(k_snop, "s_nop 0")


TEST(k_cndmask_ind, "v_cndmask_b32 %6, %4, %5, vcc")
TEST(k_cndmask_e64, "v_cndmask_b32 %4, %4, %5, s[20:21]")
TEST(k_cmp_only, "v_cmp_lt_u32 vcc, %4, %5")
TEST(k_cmp_cnd, "v_cmp_lt_u32 vcc, %4, %5\n\tv_cndmask_b32 %4, %4, %5, vcc")
TEST(k_cmp64_cnd, "v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32 %4, %4, %5, vcc")
TEST(k_add_u32, "v_add_u32 %4, %4, %5")
TEST(k_or3, "v_or3_b32 %4, %4, %5, %6")
TEST(k_lshl_add_u64, "v_lshl_add_u64 %7, %7, 3, %7")
TEST(k_bfe, "v_bfe_u32 %4, %4, 3, 7")
TEST(k_cvt_i32_f64, "v_cvt_i32_f64 %4, %0")
TEST(k_cvt_f32_f64, "v_cvt_f32_f64 %4, %0")
TEST(k_ldexp64, "v_ldexp_f64 %0, %0, 1")
TEST(k_max64, "v_max_f64 %0, %0, %1")
TEST(k_divscale, "v_div_scale_f64 %0, vcc, %0, %1, %0")
TEST(k_divfixup, "v_div_fixup_f64 %0, %0, %1, %2")
TEST(k_ifblock, "s_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 1f\n\tv_xor_b32 %4, %4, %5\n1:\n\ts_or_b64 exec, exec, s[20:21]")
TEST(k_ballot_bcnt, "v_cmp_lt_u32 vcc, %4, %5\n\ts_bcnt1_i32_b64 s20, vcc\n\tv_add_u32 %4, s20, %4")
TEST(k_mbcnt, "v_mbcnt_lo_u32_b32 %4, vcc_lo, 0\n\tv_mbcnt_hi_u32_b32 %4, vcc_hi, %4")
TEST(k_dswrite_read, "ds_write_b32 %5, %4\n\tds_read_b32 %4, %5\n\ts_waitcnt lgkmcnt(0)")
TEST(k_dpp_xor, "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
TEST(k_fmac64_lit, "v_fmac_f64 %0, %1, %2")
TEST(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
TEST(k_xor_lit, "v_xor_b32 %4, 0x12345678, %4")


TEST(k_ifnobr, "s_and_saveexec_b64 s[20:21], vcc\n\tv_xor_b32 %4, %4, %5\n\ts_or_b64 exec, exec, s[20:21]")
TEST(k_cmp_ifnobr, "v_cmp_lt_u32 vcc, %5, %6\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_xor_b32 %4, %4, %5\n\ts_or_b64 exec, exec, s[20:21]")
TEST(k_cmp_if, "v_cmp_lt_u32 vcc, %5, %6\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 1f\n\tv_xor_b32 %4, %4, %5\n1:\n\ts_or_b64 exec, exec, s[20:21]")
TEST(k_cmp_sel, "v_cmp_lt_u32 vcc, %5, %4\n\tv_xor_b32 %6, %4, %5\n\tv_cndmask_b32 %4, %4, %6, vcc")
TEST(k_brvcc, "s_cbranch_vccz 1f\n\tv_xor_b32 %4, %4, %5\n1:")
TEST(k_brscc, "s_cmp_eq_u32 s20, 77\n\ts_cbranch_scc1 1f\n\tv_xor_b32 %4, %4, %5\n1:")
TEST(k_cmp_brvcc, "v_cmp_lt_u32 vcc, %5, %6\n\ts_cbranch_vccz 1f\n\tv_xor_b32 %4, %4, %5\n1:")
TEST(k_waitcnt, "s_waitcnt vmcnt(0) lgkmcnt(0)")
TEST(k_xor_sgpr, "v_xor_b32 %4, s20, %4")
TEST(k_sadd, "s_add_u32 s20, s20, 1")
TEST(k_sadd_vuse, "s_add_u32 s20, s20, 1\n\tv_xor_b32 %4, s20, %4")

typedef void (*kern_t)(unsigned long long*, double);
struct T { const char* name; kern_t k; int per; };

int main() {
    unsigned long long* out;
    hipMalloc(&out, 8192 * 8);
    std::vector<unsigned long long> h(8192);
    T tests[] = {{"empty", k_empty, 1}, {"v_fma_f64 dependent", k_fma64_dep, 1}, {"v_fma_f64 x2 independent", k_fma64_ind2, 2}, {"v_add_f64 dependent", k_add64_dep, 1},
                 {"v_mul_f64 dependent", k_mul64_dep, 1}, {"v_mov_b32 literal", k_mov32, 1}, {"v_xor_b32 dependent", k_xor_dep, 1}, {"v_xor_b32 x2 independent", k_xor_ind2, 2},
                 {"v_mad_u64_u32 dependent", k_mad64_dep, 1}, {"v_mul_hi_u32 dependent", k_mulhi_dep, 1}, {"v_mul_lo_u32 dependent", k_mullo_dep, 1},
                 {"v_mul_u32_u24 dependent", k_mul24_dep, 1}, {"v_rcp_f64 dependent", k_rcp64_dep, 1}, {"v_rsq_f64 dependent", k_rsq64_dep, 1},
                 {"v_cvt_f64_i32", k_cvt_f64_i32, 1}, {"v_cndmask_b32 dependent", k_cndmask, 1}, {"s_mov_b32 literal", k_smov, 1},
                 {"2 s_mov + v_fma_f64(sgpr) [per group]", k_smov2_fma, 1}, {"2 v_mov + v_fma_f64 [per group]", k_vmov2_fmac, 1},
                 {"v_cmp -> s_and -> v_cndmask [per group]", k_cmp_sand, 1}, {"ds_bpermute + wait [per group]", k_bperm, 1},
                 {"v_readfirstlane -> v_xor(sgpr) [per group]", k_readlane, 1}, {"s_nop 0", k_snop, 1},
                 {"v_cndmask_b32 (vcc) independent", k_cndmask_ind, 1}, {"v_cndmask_b32 e64 (sgpr pair mask) dependent", k_cndmask_e64, 1}, {"v_cmp_lt_u32 -> vcc", k_cmp_only, 1},
                 {"v_cmp_u32 + v_cndmask [per group]", k_cmp_cnd, 1}, {"v_cmp_f64 + v_cndmask [per group]", k_cmp64_cnd, 1}, {"v_add_u32 dependent", k_add_u32, 1},
                 {"v_or3_b32 dependent", k_or3, 1}, {"v_lshl_add_u64 dependent", k_lshl_add_u64, 1}, {"v_bfe_u32 dependent", k_bfe, 1}, {"v_cvt_i32_f64", k_cvt_i32_f64, 1},
                 {"v_cvt_f32_f64", k_cvt_f32_f64, 1}, {"v_ldexp_f64 dependent", k_ldexp64, 1}, {"v_max_f64 dependent", k_max64, 1}, {"v_div_scale_f64 dependent", k_divscale, 1},
                 {"v_div_fixup_f64 dependent", k_divfixup, 1}, {"if-block: saveexec + cbranch_execz + 1 valu + s_or exec [per group]", k_ifblock, 1},
                 {"v_cmp -> s_bcnt1 -> v_add(sgpr) [per group]", k_ballot_bcnt, 1}, {"v_mbcnt lo+hi [per group]", k_mbcnt, 1},
                 {"ds_write + ds_read + wait [per group]", k_dswrite_read, 1}, {"v_mov_b32_dpp quad_perm dependent", k_dpp_xor, 1},
                 {"v_fmac_f64 (accumulating) dependent", k_fmac64_lit, 1}, {"v_pk_fma_f32 dependent", k_pk_fma_f32, 1}, {"v_xor_b32 with a 32-bit literal", k_xor_lit, 1},
                 {"if-block without the skip branch: saveexec + valu + s_or [per group]", k_ifnobr, 1}, {"v_cmp + saveexec + valu + s_or [per group]", k_cmp_ifnobr, 1},
                 {"v_cmp + saveexec + cbranch_execz + valu + s_or [per group]", k_cmp_if, 1}, {"the same as a select: v_cmp + valu + v_cndmask [per group]", k_cmp_sel, 1},
                 {"s_cbranch_vccz (not taken) + valu [per group]", k_brvcc, 1}, {"s_cmp + s_cbranch_scc1 (not taken) + valu [per group]", k_brscc, 1},
                 {"v_cmp + s_cbranch_vccz (not taken) + valu [per group]", k_cmp_brvcc, 1}, {"s_waitcnt (nothing outstanding)", k_waitcnt, 1},
                 {"v_xor_b32 with an SGPR operand", k_xor_sgpr, 1}, {"s_add_u32 dependent", k_sadd, 1}, {"s_add_u32 -> v_xor(sgpr) [per group]", k_sadd_vuse, 1}};
    // s_memtime ticks -> shader cycles: calibrate with a chain of dependent v_xor (4 cycles each on a 16-lane SIMD)
    printf("%-52s %10s %10s %10s   (shader cycles per instruction [or group] of ONE wave's stream, from first start to last end of the workgroup's waves; 1 / 2 / 4 waves per SIMD = 4 / 8 / 16 waves per workgroup on one CU)\n", "test", "1 wave", "2 waves", "4 waves");
    for (auto& t : tests) {
        double r[3];
        int wpb[3] = {256, 512, 1024};
        for (int i = 0; i < 3; ++i) {
            hipLaunchKernelGGL(t.k, dim3(1), dim3(wpb[i]), 0, 0, out, 0.5);      // warm
            hipLaunchKernelGGL(t.k, dim3(64), dim3(wpb[i]), 0, 0, out, 0.5);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, 64 * 16 * 2 * 8, hipMemcpyDeviceToHost);
            const int nw = wpb[i] / 64;
            double s = 0;
            for (int b = 0; b < 64; ++b) {                               // block time = last end - first start over its waves
                unsigned long long lo = ~0ull, hi = 0;
                for (int w = 0; w < nw; ++w) { lo = std::min(lo, h[(b * 16 + w) * 2]); hi = std::max(hi, h[(b * 16 + w) * 2 + 1]); }
                s += double(hi - lo);
            }
            r[i] = s / 64 / REPT / t.per;
        }
        printf("%-52s %10.2f %10.2f %10.2f\n", t.name, r[0], r[1], r[2]);
    }
    return 0;
}
