// Microbenchmarks of individual kernels inside the library (HIP events on the launch stream,
// weights rotated through > 256 MiB so the Infinity Cache cannot hold them).  Tuning aid only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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

__global__ void maxdiff_kernel(const float* a, const float* b, long n, float* out);
// Correctness screen for a decode-GEMM variant: out(variant, tiled or row-major W as it expects) vs the v1 kernel
// on the same random operands, split-K slabs reduced; returns max |diff| and max |ref|.
__global__ void reduce_slabs_kernel(const float* p, int S, long slab, float* o) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < slab; i += (long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += p[(long)s * slab + i];
        o[i] = a;
    }
}
extern "C" int pg_bench_skinny_verify(int M, int N, int K, int variant, int S, int tiled, int reps, float* maxdiff, float* maxref) {
    bf16 *x, *W, *Wt; float *o0, *o1, *r0, *r1, *md;
    hipMalloc((void**)&x, (long)M * K * 2); hipMalloc((void**)&W, (long)N * K * 2); hipMalloc((void**)&Wt, (long)N * K * 2);
    hipMalloc((void**)&o0, (long)S * M * N * 4); hipMalloc((void**)&o1, (long)S * M * N * 4);
    hipMalloc((void**)&r0, (long)M * N * 4); hipMalloc((void**)&r1, (long)M * N * 4); hipMalloc((void**)&md, 8);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, x, (long)M * K, 3u);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, W, (long)N * K, 7u);
    launch_tile_weights(0, W, Wt, N, K);
    hipMemset(md, 0, 8);
    int rc = 0;
    launch_gemm_skinny_variant(0, 1, x, W, o0, M, N, K, S);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o0, S, (long)M * N, r0);
    hipMemset(r1, 0, (long)M * N * 4);
    hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, md + 1);       // max |ref|
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(o1, 0xff, (long)S * M * N * 4);
        if (!launch_gemm_skinny_variant(0, variant, x, tiled ? Wt : W, o1, M, N, K, S)) { rc = -1; break; }
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o1, S, (long)M * N, r1);
        hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, md);
    }
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) rc = -2;
    float h[2] = {0, 0}; hipMemcpy(h, md, 8, hipMemcpyDeviceToHost);
    if (maxdiff) *maxdiff = h[0]; if (maxref) *maxref = h[1];
    hipFree(x); hipFree(W); hipFree(Wt); hipFree(o0); hipFree(o1); hipFree(r0); hipFree(r1); hipFree(md);
    return rc;
}

// Per-wave s_memtime stamps of ONE launch of a v4 variant (after warm-up launches on rotating weights):
// stamps_host [nwaves][64] cycle counters; returns the number of waves, or < 0.
extern unsigned long long* g_sk4_prof;
extern "C" int pg_bench_sk4_profile(int M, int N, int K, int variant, int S, unsigned long long* stamps_host, int max_waves) {
    const long wbytes = (long)N * K * 2;
    const int nbuf = 8;
    std::vector<bf16*> Ws(nbuf);
    for (auto& p : Ws) { if (hipMalloc((void**)&p, wbytes) != hipSuccess) return -2; hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, p, (long)N * K, 7u); }
    bf16* x; float* out; unsigned long long* prof;
    hipMalloc((void**)&x, (long)M * K * 2); hipMalloc((void**)&out, (long)S * M * N * 4);
    hipMalloc((void**)&prof, (size_t)max_waves * 64 * 8); hipMemset(prof, 0, (size_t)max_waves * 64 * 8);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, x, (long)M * K, 3u);
    hipStream_t s; hipStreamCreate(&s);
    int rc = 0;
    for (int it = 0; it < nbuf; ++it) {
        g_sk4_prof = it == nbuf - 1 ? prof : nullptr;
        if (!launch_gemm_skinny_variant(s, variant, x, Ws[it], out, M, N, K, S)) { rc = -1; break; }
    }
    g_sk4_prof = nullptr;
    hipStreamSynchronize(s);
    if (hipGetLastError() != hipSuccess) rc = -2;
    hipMemcpy(stamps_host, prof, (size_t)max_waves * 64 * 8, hipMemcpyDeviceToHost);
    for (auto p : Ws) hipFree(p);
    hipFree(x); hipFree(out); hipFree(prof); hipStreamDestroy(s);
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
    const int nreg = (int)((768L << 20) / bytes) + 2;     // rotate through > 768 MiB (Infinity Cache is 256 MiB)
    const long total = nreg * bytes;
    char* buf; if (hipMalloc((void**)&buf, total) != hipSuccess) return -2;
    hipMemset(buf, 1, total);
    uint32_t* out; hipMalloc((void**)&out, blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s; hipStreamCreate(&s);
    const long nvpb = bytes / 16 / blocks;
    for (int it = -3; it < iters; ++it) {
        if (it == 0) hipEventRecord(e0, s);
        const u32x4* p = (const u32x4*)(buf + (long)((it + 3) % nreg) * bytes);
        if (nt) hipLaunchKernelGGL(stream_read_kernel<true>, dim3(blocks), dim3(256), 0, s, p, nvpb, out);
        else hipLaunchKernelGGL(stream_read_kernel<false>, dim3(blocks), dim3(256), 0, s, p, nvpb, out);
    }
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (us_out) *us_out = ms * 1e3f / iters;
    hipFree(buf); hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return 0;
}

// ------------------------------------------------------------------------------- big GEMM / conv
// Times launch_gemm<bf16> (plain A, or the implicit-im2col 3x3 conv loader when Hi > 0) with the
// 256x256 kernel on (mode 1/2) or off (mode 0), and returns the max |difference| between the two
// kernels' fp32 outputs on uniform random operands (verify != 0).
__global__ void maxdiff_kernel(const float* a, const float* b, long n, float* out) {
    float m = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = fmaxf(m, d != d ? 1e30f : d);
    }
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax((int*)out, __float_as_int(m));
}
extern "C" int pg_bench_gemm(int M, int N, int K, int Hi, int Wi, int Cin, int up, int mode, int iters, int verify,
                             float* us_out, float* maxdiff_out) {
    const bool conv = Hi > 0;
    int B = 1;
    long a_elems = (long)M * K;
    if (conv) { const int Ho = Hi << up, Wo = Wi << up; B = M / (Ho * Wo); a_elems = (long)B * Hi * Wi * Cin; K = 9 * Cin; }
    bf16 *A, *Wt, *zeros; float *o0, *o1, *md;
    if (hipMalloc((void**)&A, a_elems * 2) != hipSuccess) return -2;
    hipMalloc((void**)&Wt, (long)N * K * 2); hipMalloc((void**)&zeros, 4096); hipMemset(zeros, 0, 4096);
    hipMalloc((void**)&o0, (long)M * N * 4); hipMalloc((void**)&md, 4); hipMemset(md, 0, 4);
    o1 = nullptr;
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, A, a_elems, 11u);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, Wt, (long)N * K, 5u);
    hipStream_t s; hipStreamCreate(&s);
    GemmA ga; ga.ptr = A; ga.lda = K;
    if (conv) { ga.kind = 1; ga.Hi = Hi; ga.Wi = Wi; ga.Cin = Cin; ga.up = up; ga.zeros = zeros; }
    GemmEpi e; e.out = o0; e.out_f32 = getenv("PG_BENCH_OUT_BF16") ? 0 : 1; e.ldc = N;      // bf16 output: timing only (verify = 0)
    PgTune tune; const PgTune* const saved = pg_tune; pg_tune = &tune;
    if (getenv("PG_CONV_HALO")) tune.conv_halo = atoi(getenv("PG_CONV_HALO"));
    hipDeviceSynchronize();
    if (verify) {
        hipMalloc((void**)&o1, (long)M * N * 4);
        tune.gemm256 = 0; e.out = o1; launch_gemm<bf16>(s, ga, Wt, K, 0, e, M, N, K, 1);
        // race screen: `verify` independent launches, each compared element-wise with the 128x128 result
        for (int v = 0; v < verify; ++v) {
            tune.gemm256 = mode ? mode : 1; e.out = o0; hipMemsetAsync(o0, 0xff, (long)M * N * 4, s); launch_gemm<bf16>(s, ga, Wt, K, 0, e, M, N, K, 1);
            hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, s, o0, o1, (long)M * N, md);
        }
        hipStreamSynchronize(s);
        hipMemcpy(maxdiff_out, md, 4, hipMemcpyDeviceToHost);
    }
    tune.gemm256 = mode; e.out = o0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_gemm<bf16>(s, ga, Wt, K, 0, e, M, N, K, 1);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_gemm<bf16>(s, ga, Wt, K, 0, e, M, N, K, 1);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    *us_out = ms * 1000.f / iters;
    pg_tune = saved;
    const int rc = hipGetLastError() == hipSuccess ? 0 : -1;
    hipFree(A); hipFree(Wt); hipFree(zeros); hipFree(o0); if (o1) hipFree(o1); hipFree(md);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return rc;
}
