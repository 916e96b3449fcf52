// GPU box: do instructions of DIFFERENT classes from two waves resident on one SIMD issue beside each other on gfx950?
//   hipcc --offload-arch=gfx950 -O2 coissue.hip -o coissue && ./coissue
// A workgroup of 8 waves sits on one CU, two waves per SIMD (the residency of the headline batch).  Waves 0..3
// run stream A, waves 4..7 stream B; every wave reads HW_ID so that the pairing (one A and one B per SIMD) is CHECKED, not assumed.
// Each stream is REPT copies of its body between two s_memtime reads.  Three launches per pair: A alone (4 waves, one per SIMD),
// B alone, A and B together.  Reported: shader cycles per body copy of an A wave and of a B wave, alone and together, and the
// pair's span (first start to last end of a SIMD's two waves).  Reading: if the classes co-issue, "together" equals "alone" for
// both; if the SIMD issues one instruction of any class per slot, the span is the sum of the two alone figures.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REPT 256
#define STR2(x) #x
#define STR(x) STR2(x)

struct Rec { unsigned long long t0, t1; unsigned hw, role; };

#define PROLOGUE                                                                                             \
    double a = seedd + threadIdx.x, b = seedd * 1.0000001, c = 1.0 - seedd;                                 \
    unsigned x = threadIdx.x * 2654435761u, y = x ^ 0x9e3779b9u, z = 12345u + threadIdx.x;                   \
    unsigned lds_addr = (threadIdx.x & 63) * 4;                                                              \
    const unsigned* gp = gbuf + (threadIdx.x & 63);                                                          \
    unsigned long long t0 = 0, t1 = 0;                                                                       \
    unsigned hw;                                                                                             \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));

#define STREAM(body)                                                                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); \
    asm volatile(".rept " STR(REPT) "\n\t" body "\n\t.endr\n\ts_waitcnt vmcnt(0) lgkmcnt(0)"                 \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(x), "+v"(y), "+v"(z) : "v"(lds_addr), "v"(gp)              \
                 : "memory", "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");

// mode: 0 = every wave runs A (4, 8 or 16 waves per workgroup), 1 = every wave runs B (4 waves), 2 = waves 0..3 A, 4..7 B (8 waves)
#define PAIR(name, bodyA, bodyB)                                                                             \
    __global__ void name(Rec* out, const unsigned* gbuf, double seedd, int mode) {                           \
        __shared__ unsigned lds[1024];                                                                        \
        lds[threadIdx.x & 1023] = threadIdx.x;                                                                \
        __syncthreads();                                                                                     \
        PROLOGUE                                                                                             \
        const unsigned wave = threadIdx.x >> 6;                                                              \
        const unsigned role = mode == 2 ? (wave >> 2) : (unsigned)mode;                                      \
        if (__builtin_amdgcn_readfirstlane(role) == 0) { STREAM(bodyA) } else { STREAM(bodyB) }              \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = Rec{t0, t1, hw, role};                     \
        if (a + b + c == 123.456 && x + y + z == 77) out[4000].hw = lds[x & 255];                            \
    }

// operands: %0 a  %1 b  %2 c (f64)   %3 x  %4 y  %5 z (u32)   %6 LDS byte address   %7 global pointer (64-bit VGPR pair)
#define V_XOR   "v_xor_b32 %3, %3, %4"
#define V_FMA64 "v_fma_f64 %0, %0, %1, %2"
#define S_ADD   "s_add_u32 s20, s20, 1"
#define S_MOVL  "s_mov_b32 s20, 0x12345678"
#define S_NOP   "s_nop 0"
#define S_WAIT  "s_waitcnt vmcnt(0) lgkmcnt(0)"
#define DS_RD   "ds_read_b32 %5, %6\n\ts_waitcnt lgkmcnt(0)"
#define DS_RD4  "ds_read_b32 %5, %6\n\tds_read_b32 %4, %6 offset:256\n\tds_read_b32 %5, %6 offset:512\n\tds_read_b32 %4, %6 offset:768\n\ts_waitcnt lgkmcnt(0)"
#define VM_LD   "global_load_dword %5, %7, off\n\ts_waitcnt vmcnt(0)"
#define VM_LD4  "global_load_dword %5, %7, off\n\tglobal_load_dword %4, %7, off offset:256\n\tglobal_load_dword %5, %7, off offset:512\n\tglobal_load_dword %4, %7, off offset:768\n\ts_waitcnt vmcnt(0)"
#define BALLOT  "v_cmp_lt_u32 vcc, %3, %4\n\ts_and_b64 s[22:23], vcc, exec\n\tv_cndmask_b32 %5, 0, 1, s[22:23]"
#define MASKED  "v_cmp_lt_u32 vcc, %4, %5\n\ts_and_saveexec_b64 s[22:23], vcc\n\tv_xor_b32 %3, %3, %4\n\ts_or_b64 exec, exec, s[22:23]"
// the step kernel's own class mix per wave at 65 536 x 1v1 (PMC r04_g: 600 VALU : 220 SALU : 13 LDS): 8 VALU + 3 SALU, no hand-offs
#define MIX_8V3S "v_xor_b32 %3, %3, %4\n\ts_add_u32 s20, s20, 1\n\tv_add_u32 %5, %5, %4\n\tv_fma_f64 %0, %0, %1, %2\n\tv_xor_b32 %3, %3, %4\n\ts_mov_b32 s21, 0x3ff00000\n\t" \
                 "v_add_u32 %5, %5, %4\n\tv_fma_f64 %0, %0, %1, %2\n\tv_xor_b32 %3, %3, %4\n\ts_mov_b32 s24, 0x12345678\n\tv_add_u32 %5, %5, %4"
// the same mix with the scalar results consumed by the next vector instruction (a literal pair feeding a binary64 operand)
#define MIX_DEP  "s_mov_b32 s20, 0x12345678\n\ts_mov_b32 s21, 0x3ff00000\n\tv_fma_f64 %0, %0, %1, s[20:21]\n\tv_xor_b32 %3, %3, %4\n\tv_add_u32 %5, %5, %4\n\t" \
                 "v_xor_b32 %3, %3, %4\n\ts_add_u32 s24, s24, 1\n\tv_add_u32 %5, s24, %5\n\tv_fma_f64 %0, %0, %1, %2\n\tv_xor_b32 %3, %3, %4\n\tv_add_u32 %5, %5, %4"
// three independent vector chains per wave (what "more independent work per wave" would give a lone wave)
#define V_IND3  "v_xor_b32 %3, %3, %4\n\tv_add_u32 %5, %5, %4\n\tv_fma_f64 %0, %0, %1, %2"

PAIR(p_valu_valu, V_XOR, V_XOR)
PAIR(p_valu_sadd, V_XOR, S_ADD)
PAIR(p_valu_smov, V_XOR, S_MOVL)
PAIR(p_fma_sadd, V_FMA64, S_ADD)
PAIR(p_valu_snop, V_XOR, S_NOP)
PAIR(p_valu_swait, V_XOR, S_WAIT)
PAIR(p_valu_lds, V_XOR, DS_RD)
PAIR(p_valu_lds4, V_XOR, DS_RD4)
PAIR(p_valu_vmem, V_XOR, VM_LD)
PAIR(p_valu_vmem4, V_XOR, VM_LD4)
PAIR(p_sadd_sadd, S_ADD, S_ADD)
PAIR(p_sadd_lds, S_ADD, DS_RD)
PAIR(p_valu_ballot, V_XOR, BALLOT)
PAIR(p_ballot_ballot, BALLOT, BALLOT)
PAIR(p_valu_masked, V_XOR, MASKED)
PAIR(p_masked_masked, MASKED, MASKED)
PAIR(p_mix_mix, MIX_8V3S, MIX_8V3S)
PAIR(p_mixdep_mixdep, MIX_DEP, MIX_DEP)
PAIR(p_ind3_ind3, V_IND3, V_IND3)
PAIR(p_ind3_sadd, V_IND3, S_ADD)
// part 2 only: one instruction class per kernel (B unused)
#define ONE(name, body) PAIR(name, body, S_NOP)
ONE(o_add_u32, "v_add_u32 %3, %3, %4")
ONE(o_and_or, "v_and_or_b32 %3, %3, %4, %5")
ONE(o_bfe, "v_bfe_u32 %3, %3, 3, 27")
ONE(o_cndmask, "v_cndmask_b32 %3, %3, %4, vcc")
ONE(o_cmp_u32, "v_cmp_lt_u32 vcc, %3, %4")
ONE(o_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
ONE(o_mul_lo, "v_mul_lo_u32 %3, %3, %4")
ONE(o_mul_hi, "v_mul_hi_u32 %3, %3, %4")
ONE(o_mul_u24, "v_mul_u32_u24 %3, %3, %4")
ONE(o_mad_u64, "v_mad_u64_u32 %0, vcc, %3, %4, %0")
ONE(o_add_f64, "v_add_f64 %0, %0, %1")
ONE(o_mul_f64, "v_mul_f64 %0, %0, %1")
ONE(o_fma_f64_ind, "v_fma_f64 %0, %1, %2, %0\n\tv_fma_f64 %1, %1, %2, %1")
ONE(o_rcp_f64, "v_rcp_f64 %0, %0")
ONE(o_rsq_f64, "v_rsq_f64 %0, %0")
ONE(o_cvt_f64_i32, "v_cvt_f64_i32 %0, %3")
ONE(o_cvt_i32_f64, "v_cvt_i32_f64 %3, %0")
ONE(o_cvt_f32_f64, "v_cvt_f32_f64 %3, %0")
ONE(o_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
ONE(o_fma_f32, "v_fma_f32 %3, %3, %4, %5")
ONE(o_pk_add_i16, "v_pk_add_i16 %3, %3, %4")
ONE(o_bitop3, "v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96")
ONE(o_dpp, "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
ONE(o_readlane, "v_readfirstlane_b32 s20, %3")
ONE(o_mbcnt, "v_mbcnt_lo_u32_b32 %3, vcc_lo, %3")
ONE(o_smov, "s_mov_b32 s20, 0x12345678")
ONE(o_sand64, "s_and_b64 s[22:23], s[22:23], exec")
ONE(o_scmp_br, "s_cmp_eq_u32 s20, 77\n\ts_cbranch_scc1 1f\n1:")
ONE(o_saveexec, "s_and_saveexec_b64 s[22:23], vcc\n\ts_or_b64 exec, exec, s[22:23]")
ONE(o_bperm, "ds_bpermute_b32 %5, %6, %3\n\ts_waitcnt lgkmcnt(0)")
ONE(o_lds_wr_rd, "ds_write_b32 %6, %3\n\tds_read_b32 %5, %6\n\ts_waitcnt lgkmcnt(0)")

typedef void (*kern_t)(Rec*, const unsigned*, double, int);
struct P { const char* a; const char* b; kern_t k; int na, nb; };

int main() {
    Rec* out; unsigned* gbuf;
    hipMalloc(&out, 8192 * sizeof(Rec));
    hipMalloc(&gbuf, 1 << 16);
    hipMemset(gbuf, 0, 1 << 16);
    std::vector<Rec> h(64 * 16);
    P pairs[] = {
        {"v_xor dep", "v_xor dep", p_valu_valu, 1, 1}, {"v_xor dep", "s_add_u32 dep", p_valu_sadd, 1, 1}, {"v_xor dep", "s_mov_b32 literal", p_valu_smov, 1, 1},
        {"v_fma_f64 dep", "s_add_u32 dep", p_fma_sadd, 1, 1}, {"v_xor dep", "s_nop 0", p_valu_snop, 1, 1}, {"v_xor dep", "s_waitcnt (idle)", p_valu_swait, 1, 1},
        {"v_xor dep", "ds_read + wait", p_valu_lds, 1, 1}, {"v_xor dep", "4 ds_read + wait [/grp]", p_valu_lds4, 1, 1},
        {"v_xor dep", "global_load + wait", p_valu_vmem, 1, 1}, {"v_xor dep", "4 global_load + wait [/grp]", p_valu_vmem4, 1, 1},
        {"s_add_u32 dep", "s_add_u32 dep", p_sadd_sadd, 1, 1}, {"s_add_u32 dep", "ds_read + wait", p_sadd_lds, 1, 1},
        {"v_xor dep", "v_cmp>s_and>v_cndmask [/grp]", p_valu_ballot, 1, 1}, {"ballot group", "ballot group", p_ballot_ballot, 1, 1},
        {"v_xor dep", "cmp+saveexec+valu+s_or [/grp]", p_valu_masked, 1, 1}, {"masked group", "masked group", p_masked_masked, 1, 1},
        {"8 VALU + 3 SALU indep [/grp of 11]", "same", p_mix_mix, 1, 1}, {"8 VALU + 3 SALU, scalar feeds vector [/grp of 11]", "same", p_mixdep_mixdep, 1, 1},
        {"3 indep VALU chains [/grp of 3]", "same", p_ind3_ind3, 1, 1}, {"3 indep VALU chains [/grp of 3]", "s_add_u32 dep", p_ind3_sadd, 1, 1}};
    printf("shader cycles per body copy (REPT %d), mean over 64 workgroups x 4 SIMDs; 'pairs ok' = SIMDs of the A+B launch holding exactly one A and one B wave\n", REPT);
    printf("%-48s | %-32s | %8s %8s | %8s %8s %8s | %8s | %s\n", "stream A", "stream B", "A alone", "B alone", "A with B", "B with A", "span", "sum/span", "pairs ok");
    for (auto& p : pairs) {
        double alone[2] = {0, 0}, tog[2] = {0, 0}, span = 0;
        int okpairs = 0, allpairs = 0;
        for (int mode = 0; mode < 3; ++mode) {
            const int threads = mode == 2 ? 512 : 256, nw = threads / 64;
            hipLaunchKernelGGL(p.k, dim3(1), dim3(threads), 0, 0, out, gbuf, 0.5, mode);      // warm (instruction cache)
            hipLaunchKernelGGL(p.k, dim3(64), dim3(threads), 0, 0, out, gbuf, 0.5, mode);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, 64 * 16 * sizeof(Rec), hipMemcpyDeviceToHost);
            double s[2] = {0, 0}; int cnt[2] = {0, 0};
            for (int b = 0; b < 64; ++b)
                for (int w = 0; w < nw; ++w) { const Rec& r = h[b * 16 + w]; s[r.role] += double(r.t1 - r.t0); cnt[r.role]++; }
            if (mode < 2) alone[mode] = s[mode] / cnt[mode] / REPT;
            else {
                tog[0] = s[0] / cnt[0] / REPT; tog[1] = s[1] / cnt[1] / REPT;
                for (int b = 0; b < 64; ++b)
                    for (int simd = 0; simd < 4; ++simd) {
                        unsigned long long lo = ~0ull, hi = 0; int na = 0, nb = 0;
                        for (int w = 0; w < 8; ++w) {
                            const Rec& r = h[b * 16 + w];
                            if (((r.hw >> 4) & 3) != (unsigned)simd) continue;
                            (r.role ? nb : na)++; lo = std::min(lo, r.t0); hi = std::max(hi, r.t1);
                        }
                        allpairs++;
                        if (na == 1 && nb == 1) { okpairs++; span += double(hi - lo); }
                    }
                span = okpairs ? span / okpairs / REPT : 0;
            }
        }
        printf("%-48s | %-32s | %8.2f %8.2f | %8.2f %8.2f %8.2f | %8.2f | %d/%d\n", p.a, p.b, alone[0], alone[1], tog[0], tog[1], span,
               span > 0 ? (alone[0] + alone[1]) / span : 0.0, okpairs, allpairs);
    }
    // ---- part 2: N waves per SIMD all running the SAME stream, released together by the workgroup barrier: what one SIMD issues per cycle
    struct S { const char* name; kern_t k; int instr; };
    S same[] = {{"v_xor dep", p_valu_valu, 1}, {"v_fma_f64 dep", p_fma_sadd, 1}, {"s_add_u32 dep", p_sadd_sadd, 1}, {"3 indep VALU chains (xor, add, fma64)", p_ind3_ind3, 3},
                {"8 VALU (2 of them fma64) + 3 SALU indep", p_mix_mix, 11}, {"8 VALU + 3 SALU, scalar feeds vector", p_mixdep_mixdep, 11},
                {"ballot group (v_cmp > s_and > v_cndmask)", p_ballot_ballot, 3}, {"masked group (cmp, saveexec, valu, s_or)", p_masked_masked, 4},
                {"v_add_u32 dep", o_add_u32, 1}, {"v_and_or_b32 dep", o_and_or, 1}, {"v_bfe_u32 dep", o_bfe, 1}, {"v_cndmask_b32 (vcc) dep", o_cndmask, 1},
                {"v_cmp_lt_u32 > vcc", o_cmp_u32, 1}, {"v_cmp_lt_f64 > vcc", o_cmp_f64, 1}, {"v_mul_lo_u32 dep", o_mul_lo, 1}, {"v_mul_hi_u32 dep", o_mul_hi, 1},
                {"v_mul_u32_u24 dep", o_mul_u24, 1}, {"v_mad_u64_u32 dep", o_mad_u64, 1}, {"v_add_f64 dep", o_add_f64, 1}, {"v_mul_f64 dep", o_mul_f64, 1},
                {"v_fma_f64 x2 independent", o_fma_f64_ind, 2}, {"v_rcp_f64 dep", o_rcp_f64, 1}, {"v_rsq_f64 dep", o_rsq_f64, 1}, {"v_cvt_f64_i32", o_cvt_f64_i32, 1},
                {"v_cvt_i32_f64", o_cvt_i32_f64, 1}, {"v_cvt_f32_f64", o_cvt_f32_f64, 1}, {"v_ldexp_f64 dep", o_ldexp_f64, 1}, {"v_fma_f32 dep", o_fma_f32, 1},
                {"v_pk_add_i16 dep", o_pk_add_i16, 1}, {"v_bitop3_b32 dep", o_bitop3, 1}, {"v_mov_b32_dpp quad_perm dep", o_dpp, 1},
                {"v_readfirstlane_b32", o_readlane, 1}, {"v_mbcnt_lo dep", o_mbcnt, 1}, {"s_mov_b32 literal", o_smov, 1}, {"s_and_b64 dep", o_sand64, 1},
                {"s_cmp + s_cbranch_scc1 (not taken) [/2]", o_scmp_br, 2}, {"s_and_saveexec + s_or exec [/2]", o_saveexec, 2},
                {"ds_bpermute + wait [/grp of 1]", o_bperm, 1}, {"ds_write + ds_read + wait [/grp of 2]", o_lds_wr_rd, 2}};
    printf("\nN waves per SIMD, same stream, barrier start: per-SIMD span in shader cycles per body copy, and per INSTRUCTION issued by the SIMD (span / (N x instructions per copy))\n");
    printf("%-48s | %10s %10s %10s | %10s %10s %10s\n", "stream", "N=1 /copy", "N=2", "N=4", "N=1 /instr", "N=2", "N=4");
    for (auto& t : same) {
        double per[3];
        for (int i = 0; i < 3; ++i) {
            const int N = 1 << i, threads = 256 * N, nw = threads / 64;
            hipLaunchKernelGGL(t.k, dim3(1), dim3(threads), 0, 0, out, gbuf, 0.5, 0);
            hipLaunchKernelGGL(t.k, dim3(64), dim3(threads), 0, 0, out, gbuf, 0.5, 0);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, 64 * 16 * sizeof(Rec), hipMemcpyDeviceToHost);
            double span = 0; int cnt = 0;
            for (int b = 0; b < 64; ++b)
                for (int simd = 0; simd < 4; ++simd) {
                    unsigned long long lo = ~0ull, hi = 0; int k = 0;
                    for (int w = 0; w < nw; ++w) {
                        const Rec& r = h[b * 16 + w];
                        if (((r.hw >> 4) & 3) != (unsigned)simd) continue;
                        ++k; lo = std::min(lo, r.t0); hi = std::max(hi, r.t1);
                    }
                    if (k == N) { span += double(hi - lo); ++cnt; }
                }
            per[i] = cnt ? span / cnt / REPT : 0;
        }
        printf("%-48s | %10.2f %10.2f %10.2f | %10.2f %10.2f %10.2f\n", t.name, per[0], per[1], per[2], per[0] / t.instr, per[1] / (2 * t.instr), per[2] / (4 * t.instr));
    }
    return 0;
}
