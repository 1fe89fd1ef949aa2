// Which access pattern does the memory system like when ONE wave serves MANY sequential streams?  (the FLAC kernels: a lane per subframe,
// every round a wave moves SEG bytes of each of its 64 subframes; build: hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o gather_probe)
//
// A wave owns 64 "subframes" of LEN int32 each, copies them src -> dst in rounds of SEG bytes per subframe (SEG = 128: 8 lanes x 16 B and 8
// subframes per instruction, 8 instructions per round; SEG = 256: 16 lanes per subframe, 16 instructions; ...), requests one round ahead.
// Parameters: spacing of the subframes (exactly LEN apart, or LEN + skew), which subframes a wave gets (neighbours, or strided far apart),
// waves per SIMD (grid), read-only / write-only / both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SEG, int MODE>   // MODE 0 copy, 1 read only, 2 write only
__global__ __launch_bounds__(64) void k_copy(const int *src, int *dst, const unsigned long long *soff, const unsigned long long *doff, int len, unsigned nsub, unsigned *sink) {
    constexpr int LPS = SEG / 16;          // lanes per subframe
    constexpr int SPI = 64 / LPS;          // subframes per instruction
    constexpr int NI = 64 / SPI;           // instructions per round
    const int lane = threadIdx.x, grp = lane / LPS, sub4 = 4 * (lane % LPS);
    const unsigned first = blockIdx.x * 64;
    if (first + 63 >= nsub) return;
    unsigned long long so[NI], dd[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) { so[i] = soff[first + SPI * i + grp] + sub4; dd[i] = doff[first + SPI * i + grp] + sub4; }
    uint4 pre[NI], out[NI];
    unsigned acc = 0;
    auto request = [&](int base) {
#pragma unroll
        for (int i = 0; i < NI; i++) pre[i] = MODE == 2 ? make_uint4(base, i, lane, 0) : *reinterpret_cast<const uint4 *>(src + so[i] + base);
    };
    request(0);
    constexpr int VPR = SEG / 4;           // values per round and subframe
    for (int base = 0; base < len; base += VPR) {
#pragma unroll
        for (int i = 0; i < NI; i++) out[i] = pre[i];
        if (base + VPR < len) request(base + VPR);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            if (MODE == 1) acc += out[i].x ^ out[i].w;
            else *reinterpret_cast<uint4 *>(dst + dd[i] + base) = out[i];
        }
    }
    if (MODE == 1 && acc == 0x12345678u) sink[0] = acc;
}

template <int SEG, int MODE> static float run(const int *src, int *dst, const unsigned long long *so, const unsigned long long *dof, int len, unsigned nsub, unsigned grid, unsigned *sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_copy<SEG, MODE>), dim3(grid), dim3(64), 0, 0, src, dst, so, dof, len, nsub, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_copy<SEG, MODE>), dim3(grid), dim3(64), 0, 0, src, dst, so, dof, len, nsub, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main() {
    const int len = 4096;                       // values per subframe (a FLAC block)
    const unsigned nsub = 2048 * 2 * 108 / 64 * 64;   // config 5: streams x channels x frames
    const size_t pitch_max = len + 64;
    int *src, *dst;
    unsigned *sink;
    unsigned long long *so, *dof;
    CK(hipMalloc(&src, (size_t)nsub * pitch_max * 4 + 4096));
    CK(hipMalloc(&dst, (size_t)nsub * pitch_max * 4 + 4096));
    CK(hipMalloc(&so, (size_t)nsub * 8)); CK(hipMalloc(&dof, (size_t)nsub * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, (size_t)nsub * pitch_max * 4));
    std::vector<unsigned long long> hs(nsub), hd(nsub);
    const double gb = (double)nsub * len * 4 / 1e9;
    printf("%u subframes of %d values: %.2f GB each way\n", nsub, len, gb);
    for (int skew = 0; skew <= 32; skew += 32)
        for (int strided = 0; strided <= 1; strided++) {
            // subframe j of the launch sits at slot p(j): neighbours, or (j mod 64) * (nsub / 64) + j / 64 (a wave's 64 lie far apart)
            for (unsigned j = 0; j < nsub; j++) {
                const unsigned long long slot = strided ? (unsigned long long)(j & 63) * (nsub / 64) + j / 64 : j;
                hs[j] = slot * (unsigned long long)(len + skew);
                hd[j] = slot * (unsigned long long)(len + skew);
            }
            CK(hipMemcpy(so, hs.data(), (size_t)nsub * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dof, hd.data(), (size_t)nsub * 8, hipMemcpyHostToDevice));
            const unsigned grid = nsub / 64;
            printf("skew %2d values, %s subframes per wave:\n", skew, strided ? "far-apart " : "neighbouring");
#define ROW(SEG)                                                                                                                             \
    {                                                                                                                                        \
        const float c = run<SEG, 0>(src, dst, so, dof, len, nsub, grid, sink), r = run<SEG, 1>(src, dst, so, dof, len, nsub, grid, sink),    \
                    w = run<SEG, 2>(src, dst, so, dof, len, nsub, grid, sink);                                                               \
        printf("  %4d B per subframe and round: copy %6.3f ms (%5.2f TB/s)  read %6.3f ms (%5.2f TB/s)  write %6.3f ms (%5.2f TB/s)\n", SEG, c, \
               2 * gb / c, r, gb / r, w, gb / w);                                                                                            \
    }
            ROW(128) ROW(256) ROW(512) ROW(1024)
        }
    return 0;
}
