// write-bandwidth micro-benchmark: which store pattern reaches torch.fill's rate?  (hipcc --offload-arch=gfx950 -O3 wr_probe.hip -o wr_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: persistent waves, each tile = 1024 floats written as 16 rows of 64 dwords (the resampler's pattern)
__global__ __launch_bounds__(256) void k_tile_dword(float *out, unsigned ntiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        float *o = out + (size_t)t * 1024;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r * 64 + lane] = (float)(t + r);
    }
}
// B: same tiles, 4 rows of 64 float4
__global__ __launch_bounds__(256) void k_tile_x4(float *out, unsigned ntiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        float4 *o = reinterpret_cast<float4 *>(out + (size_t)t * 1024);
#pragma unroll
        for (int r = 0; r < 4; r++) o[r * 64 + lane] = make_float4((float)t, (float)r, 0.f, 1.f);
    }
}
// C: grid-stride float4 fill (what a fill kernel does)
__global__ __launch_bounds__(256) void k_fill_x4(float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// D: one tile per wave, non-persistent (grid = ntiles / 4)
__global__ __launch_bounds__(256) void k_tile_dword_np(float *out, unsigned ntiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned t = blockIdx.x * 4 + wave;
    if (t >= ntiles) return;
    float *o = out + (size_t)t * 1024;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r * 64 + lane] = (float)(t + r);
}
// E: read 2 B + write 4 B per output (the headline's byte mix), no math: persistent tiles
__global__ __launch_bounds__(256) void k_tile_rw(const short *in, float *out, unsigned ntiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        const uint4 *ip = reinterpret_cast<const uint4 *>(in + (size_t)t * 1024);
        const uint4 a = ip[lane], b = ip[64 + lane];  // 2 KiB of int16 per tile
        float *o = out + (size_t)t * 1024;
        const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int r = 0; r < 16; r++) o[r * 64 + lane] = (float)(short)(w[r >> 1] >> (16 * (r & 1)));
    }
}

// F: E + the window goes through LDS (write 2 x 8 floats per lane, read 4 neighbours per output) — no arithmetic worth the name
__global__ __launch_bounds__(256) void k_tile_rw_lds(const short *in, float *out, unsigned ntiles) {
    __shared__ float sm[4][1040];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
    float *w = sm[wave];
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        const uint4 *ip = reinterpret_cast<const uint4 *>(in + (size_t)t * 1024);
        const uint4 a = ip[lane], b = ip[64 + lane];
        const unsigned ww[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 8; e++) { w[16 * lane + 2 * e] = (float)(short)ww[e]; w[16 * lane + 2 * e + 1] = (float)(short)(ww[e] >> 16); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *o = out + (size_t)t * 1024;
#pragma unroll
        for (int r = 0; r < 16; r++) { const unsigned q = (r * 59 + lane * 15 / 16) & 1023; o[r * 64 + lane] = w[q] + w[q + 1] + w[q + 2] + w[q + 3]; }
        __builtin_amdgcn_wave_barrier();
    }
}
// G: E + ~20 dependent-free FMAs per output, no LDS
__global__ __launch_bounds__(256) void k_tile_rw_valu(const short *in, float *out, unsigned ntiles, float k) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
    for (unsigned t = blockIdx.x * 4 + wave; t < ntiles; t += nw) {
        const uint4 *ip = reinterpret_cast<const uint4 *>(in + (size_t)t * 1024);
        const uint4 a = ip[lane], b = ip[64 + lane];
        float *o = out + (size_t)t * 1024;
        const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = (float)(short)(w[r >> 1] >> (16 * (r & 1))), u = v * k;
#pragma unroll
            for (int i = 0; i < 10; i++) { v = fmaf(v, k, u); u = fmaf(u, k, v); }
            o[r * 64 + lane] = v + u;
        }
    }
}

// H: E with the round-2 schedule: the NEXT tile's loads are issued first, the 16 results of the current tile wait in registers,
// the wave waits for those loads (vmcnt(0): the stores of the tile before have long completed), then issues its 16 stores —
// instead of waiting for loads right behind 16 fresh stores (on gfx9 that wait is vmcnt(0) and includes them)
__global__ __launch_bounds__(256) void k_tile_rw_late(const short *in, float *out, unsigned ntiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
    unsigned t = blockIdx.x * 4 + wave;
    if (t >= ntiles) return;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *ip = reinterpret_cast<const u32x4 *>(in + (size_t)t * 1024);
    u32x4 a = ip[lane], b = ip[64 + lane];
    asm volatile("" : "+v"(a), "+v"(b));  // waited for before the loop: otherwise hipcc's vmcnt(0) for this edge stands at the top of the loop, behind the stores
    for (;;) {
        const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; r++) res[r] = (float)(short)(w[r >> 1] >> (16 * (r & 1)));
        const unsigned tn = t + nw;
        const bool more = tn < ntiles;
        const u32x4 *np = reinterpret_cast<const u32x4 *>(in + (size_t)(more ? tn : t) * 1024);
        a = np[lane]; b = np[64 + lane];
        asm volatile("" : "+v"(a), "+v"(b), "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]));
        asm volatile("" : "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]), "+v"(res[13]), "+v"(res[14]), "+v"(res[15]));
        float *o = out + (size_t)t * 1024;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r * 64 + lane] = res[r];
        if (!more) break;
        t = tn;
    }
}

int main(int argc, char **argv) {
    const size_t nfl = (size_t)4096 * 480000;  // the headline's output
    float *out; short *in;
    CK(hipMalloc(&out, nfl * 4)); CK(hipMalloc(&in, nfl * 2)); CK(hipMemset(in, 1, nfl * 2));
    const unsigned ntiles = (unsigned)(nfl / 1024);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, double bytes) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; i++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s %8.3f ms  %7.0f GB/s\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e9);
    };
    for (int percu : {8, 16, 32}) {
        const unsigned grid = 256 * percu;
        char nm[64];
        snprintf(nm, sizeof nm, "A tile dword persistent, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_tile_dword, dim3(grid), dim3(256), 0, 0, out, ntiles); }, nfl * 4.0);
        snprintf(nm, sizeof nm, "B tile float4 persistent, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_tile_x4, dim3(grid), dim3(256), 0, 0, out, ntiles); }, nfl * 4.0);
        snprintf(nm, sizeof nm, "C grid-stride float4 fill, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_fill_x4, dim3(grid), dim3(256), 0, 0, reinterpret_cast<float4 *>(out), nfl / 4); }, nfl * 4.0);
        snprintf(nm, sizeof nm, "E tile read s16 + write f32, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_tile_rw, dim3(grid), dim3(256), 0, 0, in, out, ntiles); }, nfl * 6.0);
        snprintf(nm, sizeof nm, "H  = E, stores after the next loads' wait, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_tile_rw_late, dim3(grid), dim3(256), 0, 0, in, out, ntiles); }, nfl * 6.0);
        snprintf(nm, sizeof nm, "F  = E + window through LDS, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_tile_rw_lds, dim3(grid), dim3(256), 0, 0, in, out, ntiles); }, nfl * 6.0);
        snprintf(nm, sizeof nm, "G  = E + 20 FMAs per output, %d blocks/CU", percu);
        run(nm, [&] { hipLaunchKernelGGL(k_tile_rw_valu, dim3(grid), dim3(256), 0, 0, in, out, ntiles, 0.999f); }, nfl * 6.0);
    }
    run("D tile dword, one tile per wave", [&] { hipLaunchKernelGGL(k_tile_dword_np, dim3((ntiles + 3) / 4), dim3(256), 0, 0, out, ntiles); }, nfl * 4.0);
    return 0;
}
