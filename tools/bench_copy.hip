// tools/bench_copy.hip — the float4 device copy bench.py times beside the headline kernel (roofline.copy_ceiling_GBs): what THIS box moves when
// nothing is computed.  Bench infrastructure, not part of libaukit_hip.so or its ABI: built into tools/libbench_copy.so by __graft_entry__.build().
// 16 bytes per lane and access, grid-stride over a persistent grid (8 workgroups of 256 per CU), non-temporal on both sides (nothing is reused).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_bench_copy(f4 *__restrict__ dst, const f4 *__restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {   // U loads in flight per lane
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// shape 0: one plain load and store per lane and turn; 1: four non-temporal ones.  wgs: workgroups per CU of the persistent grid.
// (tools/micro/copy_probe.hip, profiles/r06_copy_probe.txt: which shape wins moves from box to box and run to run by 5 - 10 %: the bench takes the best)
extern "C" int bench_copy(void *dst, const void *src, size_t bytes, void *stream, int cus, int shape, int wgs) {
    const size_t n16 = bytes / 16;
    if (!n16) return 0;
    const dim3 grid((unsigned)(cus > 0 ? cus : 256) * (unsigned)(wgs > 0 ? wgs : 8));
    if (shape == 0) hipLaunchKernelGGL((k_bench_copy<1, false>), grid, dim3(256), 0, (hipStream_t)stream, (f4 *)dst, (const f4 *)src, n16);
    else hipLaunchKernelGGL((k_bench_copy<4, true>), grid, dim3(256), 0, (hipStream_t)stream, (f4 *)dst, (const f4 *)src, n16);
    return (int)hipGetLastError();
}
