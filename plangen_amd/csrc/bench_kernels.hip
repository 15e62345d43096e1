// Microbenchmarks of individual kernels inside the library (HIP events on the launch stream,
// weights rotated through > 256 MiB so the Infinity Cache cannot hold them).  Tuning aid only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../include/plangen_hip.h"
#include "kernels.h"

__global__ void fill_bf16_kernel(bf16* p, long n, uint32_t seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) / 65536.f - 0.5f) * 0.05f;
        ET<bf16>::st(p + i, v);
    }
}

extern "C" int pg_bench_skinny(int M, int N, int K, int variant, int S, int iters, int with_consumer, float* us_out) {
    const long wbytes = (long)N * K * 2;
    int nbuf = (int)((600L << 20) / wbytes) + 1; if (nbuf > 64) nbuf = 64; if (nbuf < 2) nbuf = 2;
    std::vector<bf16*> Ws(nbuf);
    for (auto& p : Ws) { if (hipMalloc((void**)&p, wbytes) != hipSuccess) return -2; hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, p, (long)N * K, 7u); }
    bf16 *x, *xn; float *out, *res;
    hipMalloc((void**)&x, (long)M * K * 2); hipMalloc((void**)&xn, (long)M * N * 2);
    hipMalloc((void**)&out, (long)S * M * N * 4); hipMalloc((void**)&res, (long)M * N * 4);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, x, (long)M * K, 3u);
    hipMemset(res, 0, (long)M * N * 4);
    bf16* wn; hipMalloc((void**)&wn, N * 2); hipLaunchKernelGGL(fill_bf16_kernel, dim3(8), dim3(256), 0, 0, wn, (long)N, 9u);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s; hipStreamCreate(&s);
    int rc = 0;
    for (int it = -5; it < iters; ++it) {
        if (it == 0) hipEventRecord(e0, s);
        if (!launch_gemm_skinny_variant(s, variant, x, Ws[(it + 5) % nbuf], out, M, N, K, S)) { rc = -1; break; }
        if (with_consumer) launch_rmsnorm<bf16>(s, res, out, S, (long)M * N, wn, xn, M, N, 1e-6f);
    }
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    if (hipGetLastError() != hipSuccess) rc = -2;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (us_out) *us_out = ms * 1e3f / iters;
    for (auto p : Ws) hipFree(p);
    hipFree(x); hipFree(xn); hipFree(out); hipFree(res); hipFree(wn);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return rc;
}

// Pure streaming read (calibration for the HBM-bound kernels): every block reads a contiguous
// slice with 16-byte loads, UN loads in flight per lane, result folded into one word per block.
template <bool NT>
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4* __restrict__ p, long nvec_per_block, uint32_t* __restrict__ out) {
    const u32x4* b = p + (long)blockIdx.x * nvec_per_block;
    uint32_t acc = 0;
    for (long i = threadIdx.x; i < nvec_per_block; i += 256 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long j = i + (long)u * 256;
            const long jj = j < nvec_per_block ? j : nvec_per_block - 1;
            v[u] = NT ? __builtin_nontemporal_load(b + jj) : b[jj];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
extern "C" int pg_bench_stream(long bytes, int blocks, int iters, int nt, float* us_out) {
    const long total = 3 * bytes;                      // rotate through 3 regions (> Infinity Cache)
    char* buf; if (hipMalloc((void**)&buf, total) != hipSuccess) return -2;
    hipMemset(buf, 1, total);
    uint32_t* out; hipMalloc((void**)&out, blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s; hipStreamCreate(&s);
    const long nvpb = bytes / 16 / blocks;
    for (int it = -3; it < iters; ++it) {
        if (it == 0) hipEventRecord(e0, s);
        const u32x4* p = (const u32x4*)(buf + (long)((it + 3) % 3) * bytes);
        if (nt) hipLaunchKernelGGL(stream_read_kernel<true>, dim3(blocks), dim3(256), 0, s, p, nvpb, out);
        else hipLaunchKernelGGL(stream_read_kernel<false>, dim3(blocks), dim3(256), 0, s, p, nvpb, out);
    }
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (us_out) *us_out = ms * 1e3f / iters;
    hipFree(buf); hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return 0;
}
