// tools/calib_fetch.hip — what FETCH_SIZE / WRITE_SIZE report for the access patterns of this library, on known byte counts
// (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths
// and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_fetch tools/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- /tmp/calib_fetch      (and again with WRITE_SIZE)
// Kernels (each moves BYTES = 4 GiB, beyond the 256 MiB Infinity Cache):
//   k_stream_read      coalesced 16 B / lane streaming read                        (the wave kernels)
//   k_stream_write     coalesced 16 B / lane streaming write
//   k_lane_read<R>     lane-per-region: every lane walks ITS OWN contiguous region, R x 16 B per visit, 131 072 lanes resident   (k_flac_decode's
//                      bit-stream refills: R = 1; a 64-byte refill: R = 4; a whole 128-byte line: R = 8)
//   k_lane_write<R>    the same for stores (k_flac_decode's flush: R = 8 — 32 int32 values per lane and channel)
//   k_lane_rw          both at once: 16-B reads from one region, 128-B bursts of stores into two others (the decoder's working set per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_stream_read(const uint4 *in, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void k_stream_write(uint4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = make_uint4((unsigned)i, 1, 2, 3);
}

// region = n16 / lanes vectors per lane; a visit reads R vectors; `work` dependent ALU steps between visits stand in for the decoding of those bytes
template <int R>
__global__ __launch_bounds__(64) void k_lane_read(const uint4 *in, size_t per_lane16, int work, unsigned *sink) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    const uint4 *p = in + lane * per_lane16;
    unsigned acc = 0;
    for (size_t i = 0; i + R <= per_lane16; i += R) {
        uint4 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = p[i + r];
#pragma unroll
        for (int r = 0; r < R; r++) acc ^= v[r].x ^ v[r].y ^ v[r].z ^ v[r].w;
        for (int w = 0; w < work * R; w++) acc = acc * 1664525u + 1013904223u;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int R>
__global__ __launch_bounds__(64) void k_lane_write(uint4 *out, size_t per_lane16, int work) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint4 *p = out + lane * per_lane16;
    unsigned acc = (unsigned)lane;
    for (size_t i = 0; i + R <= per_lane16; i += R) {
        for (int w = 0; w < work * R; w++) acc = acc * 1664525u + 1013904223u;
#pragma unroll
        for (int r = 0; r < R; r++) p[i + r] = make_uint4(acc, r, 2, 3);
    }
}

// per lane: 1 vector read per visit from region A; after every 8 visits... the decoder's ratio for 16-bit stereo at ~0.6 compression: 16 B of bit stream
// become ~13 samples = 52 B of int32 output; here 16 B in : 64 B out (R = 4 vectors into alternating halves of the output region, 128-B bursts per half)
__global__ __launch_bounds__(64) void k_lane_rw(const uint4 *in, uint4 *out, size_t per_lane16_in, int work, unsigned *sink) {
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    const uint4 *p = in + lane * per_lane16_in;
    uint4 *q0 = out + lane * per_lane16_in * 4, *q1 = q0 + per_lane16_in * 2;
    unsigned acc = 0;
    for (size_t i = 0; i + 4 <= per_lane16_in; i += 4) {
        for (int k = 0; k < 4; k++) {
            const uint4 v = p[i + k];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            for (int w = 0; w < work; w++) acc = acc * 1664525u + 1013904223u;
        }
        // 16 vectors out: 8 (128 B) to each channel's run
#pragma unroll
        for (int r = 0; r < 8; r++) q0[2 * i + r] = make_uint4(acc, r, 0, 0);
#pragma unroll
        for (int r = 0; r < 8; r++) q1[2 * i + r] = make_uint4(acc, r, 1, 0);
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv) {
    const size_t BYTES = 4ull << 30, n16 = BYTES / 16;
    const int work = argc > 1 ? atoi(argv[1]) : 40;
    const size_t lanes = 131072;   // 8 workgroups of one wave on each of 256 CUs: k_flac_decode's residency
    void *a, *b;
    unsigned *sink;
    CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&b, BYTES)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, BYTES)); CK(hipMemset(b, 0, BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double bytes, auto launch) {
        launch();   // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s %8.3f ms  %7.1f GB/s  (%.3f GB known)\n", name, ms, bytes / ms / 1e6, bytes / 1e9);
    };
    const size_t per = n16 / lanes;
    timed("k_stream_read", (double)BYTES, [&] { hipLaunchKernelGGL(k_stream_read, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)a, n16, sink); });
    timed("k_stream_write", (double)BYTES, [&] { hipLaunchKernelGGL(k_stream_write, dim3(256 * 16), dim3(256), 0, 0, (uint4 *)b, n16); });
    timed("k_lane_read<1>", (double)BYTES, [&] { hipLaunchKernelGGL(k_lane_read<1>, dim3(lanes / 64), dim3(64), 0, 0, (const uint4 *)a, per, work, sink); });
    timed("k_lane_read<4>", (double)BYTES, [&] { hipLaunchKernelGGL(k_lane_read<4>, dim3(lanes / 64), dim3(64), 0, 0, (const uint4 *)a, per, work, sink); });
    timed("k_lane_read<8>", (double)BYTES, [&] { hipLaunchKernelGGL(k_lane_read<8>, dim3(lanes / 64), dim3(64), 0, 0, (const uint4 *)a, per, work, sink); });
    timed("k_lane_write<1>", (double)BYTES, [&] { hipLaunchKernelGGL(k_lane_write<1>, dim3(lanes / 64), dim3(64), 0, 0, (uint4 *)b, per, work); });
    timed("k_lane_write<8>", (double)BYTES, [&] { hipLaunchKernelGGL(k_lane_write<8>, dim3(lanes / 64), dim3(64), 0, 0, (uint4 *)b, per, work); });
    // 1 GiB in, 4 GiB out
    timed("k_lane_rw", (double)(BYTES / 4) * 5, [&] { hipLaunchKernelGGL(k_lane_rw, dim3(lanes / 64), dim3(64), 0, 0, (const uint4 *)a, (uint4 *)b, per / 4, work, sink); });
    return 0;
}
