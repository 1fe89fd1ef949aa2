// tools/micro/copy_probe.hip — which float4 copy is this box's ceiling?  (grid size x loads in flight x non-temporal hints)   hipcc --offload-arch=gfx950 -O3 -o copy_probe copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k(f4 *__restrict__ dst, const f4 *__restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
template <int U, bool NTL, bool NTS> void run(const char *name, f4 *d, const f4 *s, size_t n16, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < 8; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<U, NTL, NTS>), dim3(256 * wgs), dim3(256), 0, 0, d, s, n16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1); if (r >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-28s wgs/CU %2d  %7.1f GB/s (read+write)\n", name, wgs, 2.0 * n16 * 16 / (ms[ms.size() / 2] * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = 5738496000ull / 16 * 16, n16 = bytes / 16;
    f4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 2, bytes);
    for (int wgs : {4, 8, 16, 32}) {
        run<1, false, false>("plain u1", d, s, n16, wgs);
        run<4, false, false>("plain u4", d, s, n16, wgs);
        run<4, true, true>("nt/nt u4", d, s, n16, wgs);
        run<4, false, true>("plain/nt u4", d, s, n16, wgs);
        run<8, false, false>("plain u8", d, s, n16, wgs);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) { hipEventRecord(e0); hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1); float m; hipEventElapsedTime(&m, e0, e1); printf("hipMemcpy D2D %7.1f GB/s\n", 2.0 * bytes / (m * 1e-3) / 1e9); }
    return 0;
}
