// GPU box: does a kernel's straight-line code run from a warm instruction cache when the same kernel is launched again?
// A 16 KB / 48 KB block of v_xor_b32 (one pass, no loop), timed inside the kernel with s_memtime, for launch 1, 2, 3 ... and again
// after a different large kernel has run in between.  hipcc --offload-arch=gfx950 -O2 icache_cold.hip -o icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define BODY(name, n)                                                                                      \
    __global__ void name(unsigned long long* out, unsigned seed) {                                         \
        unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x9e3779b9u;                                \
        unsigned long long t0, t1;                                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                          \
        asm volatile(".rept " #n "\n\tv_xor_b32 %0, %0, %1\n\tv_add_u32 %1, %1, %0\n\t.endr" : "+v"(x), "+v"(y)::"memory"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                    \
        if (x + y == 12345u) out[4000] = 1;                                                                \
    }
BODY(k16, 1024)    // 2 x 1024 x 8 B = 16 KB
BODY(k48, 3072)    // 48 KB
BODY(other, 4096)  // 64 KB of different code: evicts
int main() {
    unsigned long long* out; (void)hipMalloc(&out, 8192 * 8);
    std::vector<unsigned long long> h(4096);
    auto run = [&](void (*k)(unsigned long long*, unsigned), int instr, const char* tag) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, 1u);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, 256 * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0, mx = 0; for (int i = 0; i < 1024; ++i) { s += double(h[i]); mx = mx > double(h[i]) ? mx : double(h[i]); }
        printf("%-34s mean %.2f cycles per instruction, slowest wave %.2f\n", tag, s / 1024 / instr, mx / instr);
    };
    run(k16, 2048, "16 KB kernel, launch 1");
    run(k16, 2048, "16 KB kernel, launch 2");
    run(k16, 2048, "16 KB kernel, launch 3");
    run(other, 8192, "(64 KB of other code)");
    run(k16, 2048, "16 KB kernel, after other code");
    run(k16, 2048, "16 KB kernel, launch after that");
    run(k48, 6144, "48 KB kernel, launch 1");
    run(k48, 6144, "48 KB kernel, launch 2");
    run(other, 8192, "(64 KB of other code)");
    run(k48, 6144, "48 KB kernel, after other code");
    return 0;
}
