// Micro-benchmark (diagnostic, GPU box): latency of a wave's FIRST global load after kernel start, under the same
// launch shape as the step kernel (grid of 64-thread workgroups), for data the previous kernel wrote / did not write.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/first_load_latency.hip -o /tmp/fll && /tmp/fll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void touch(uint4* buf, size_t n, unsigned v) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = make_uint4(v, v, v, v);
}
__global__ void probe(const uint4* __restrict__ buf, size_t n, int loads, unsigned long long* out, uint4* sink) {
    unsigned long long t0, t1;
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < loads; ++k) {           // independent loads, issued back to back
        const uint4 v = buf[(i + size_t(k) * (n / 8)) & (n - 1)];   // n is a power of two
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0xFFFFFFFFu) sink[i] = acc;   // keep the loads alive
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    const int blocks_full = 2048;
    const size_t n = size_t(blocks_full) * 64 * 8;   // 16 MB of uint4
    uint4 *buf, *sink; unsigned long long* out;
    hipMalloc(&buf, n * sizeof(uint4)); hipMalloc(&sink, n * sizeof(uint4)); hipMalloc(&out, blocks_full * 8);
    std::vector<unsigned long long> h(blocks_full);
    for (int blocks : {1, 256, 2048}) for (int loads : {1, 4, 8}) for (int fresh : {0, 1}) {
        double med = 0; 
        for (int rep = 0; rep < 5; ++rep) {
            if (fresh) touch<<<dim3(unsigned((n + 255) / 256)), 256>>>(buf, n, rep + 1);
            else touch<<<dim3(unsigned((n + 255) / 256)), 256>>>(sink, n, rep + 1);
            probe<<<blocks, 64>>>(buf, n, loads, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> v(h.begin(), h.begin() + blocks); std::sort(v.begin(), v.end());
            med = double(v[v.size() / 2]);
            if (rep == 4) printf("blocks %4d loads/lane %d  data %-28s first-load round trip: median %6.0f cyc  p90 %6.0f  max %6.0f\n", blocks, loads,
                                 fresh ? "written by previous kernel" : "not touched by previous kernel", med, double(v[v.size() * 9 / 10]), double(v.back()));
        }
    }
    return 0;
}
