// which CUs does a hipExtStreamCreateWithCUMask stream really use?  (hipcc --offload-arch=gfx950 -O3 cumask_probe.hip -o cumask_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_census(unsigned *out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 15) | (v > 1e30f ? 16 : 0); }
}
int main(int argc, char **argv) {
    const int nblk = 4096;
    unsigned *d; CK(hipMalloc(&d, nblk * 8));
    std::vector<unsigned> h(2 * nblk);
    for (int bits : {8, 16, 32, 64, 256}) {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < bits; i++) m[i >> 5] |= 1u << (i & 31);
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, 8, m.data()));
        hipLaunchKernelGGL(k_census, dim3(nblk), dim3(64), 0, st, d, 20000);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
        std::set<unsigned> cus; std::set<unsigned> xccs;
        for (int b = 0; b < nblk; b++) { const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15; cus.insert((x << 8) | ((hw >> 8) & 0xFF)); xccs.insert(x); }  // cu_id bits 11:8, sh/se bits
        printf("mask bits 0..%d: %zu distinct (xcc, se, cu) ids on %zu XCCs\n", bits - 1, cus.size(), xccs.size());
        CK(hipStreamDestroy(st));
    }
    return 0;
}
