#include <cstdio>
#include <cstdlib>
// Microbenchmarks of individual kernels inside the library (HIP events on the launch stream,
// weights rotated through > 256 MiB so the Infinity Cache cannot hold them).  Tuning aid only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/plangen_hip.h"
#include "kernels.h"
#include "gemm_common.h"
#include "diag.h"
#include "gemm_skinny.h"      // g_sk4_prof, SK_BK

__global__ void fill_bf16_kernel(bf16* p, long n, uint32_t seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) / 65536.f - 0.5f) * 0.05f;
        ET<bf16>::st(p + i, v);
    }
}

// pattern operands for race forensics: x[m][k] = (1 + m/128) * 2^(k/128 - 8), W = 1/128: every (row, k-chunk) contributes a distinct exact
// term, so a wrong output identifies which chunk's x piece was replaced by which
__global__ void fill_pattern_x_kernel(bf16* p, int M, int K) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * K; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), c = (int)(i % K) / 128;
        ET<bf16>::st(p + i, (1.f + (float)(m & 127) / 128.f) * exp2f((float)(c & 15) - 8.f));
    }
}
__global__ void fill_const_kernel(bf16* p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ET<bf16>::st(p + i, v);
}
extern "C" int pg_bench_skinny(int M, int N, int K, int variant, int S, int iters, int with_consumer, float* us_out) {
    const long wbytes = (long)N * K * 2;
    int nbuf = (int)((600L << 20) / wbytes) + 1; if (nbuf > 64) nbuf = 64; if (nbuf < 2) nbuf = 2;
    if (const char* e = getenv("PG_BENCH_NBUF")) nbuf = atoi(e) > 0 ? atoi(e) : nbuf;   // 1 = weights stay in the Infinity Cache
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
__global__ void save_if_bad_kernel(const float* r1, float* save, long n, const float* md, float tol, int* flag, int rep);
extern "C" int pg_bench_skinny_verify(int M, int N, int K, int variant, int S, int tiled, int reps, float* maxdiff, float* maxref) {
    bf16 *x, *W, *Wt; float *o0, *o1, *r0, *r1, *md;
    hipMalloc((void**)&x, (long)M * K * 2); hipMalloc((void**)&W, (long)N * K * 2); hipMalloc((void**)&Wt, (long)N * K * 2);
    hipMalloc((void**)&o0, (long)S * M * N * 4); hipMalloc((void**)&o1, (long)S * M * N * 4);
    hipMalloc((void**)&r0, (long)M * N * 4); hipMalloc((void**)&r1, (long)M * N * 4); hipMalloc((void**)&md, 8);
    const bool pattern = getenv("PG_VERIFY_PATTERN") != nullptr;
    if (pattern) {
        hipLaunchKernelGGL(fill_pattern_x_kernel, dim3(256), dim3(256), 0, 0, x, M, K);
        hipLaunchKernelGGL(fill_const_kernel, dim3(2048), dim3(256), 0, 0, W, (long)N * K, 1.f / 128.f);
    } else {
        hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, x, (long)M * K, 3u);
        hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, W, (long)N * K, 7u);
    }
    launch_tile_weights(0, W, Wt, N, K);
    hipMemset(md, 0, 8);
    int rc = 0;
    float* vsave = nullptr; int* vflag = nullptr;
    if (getenv("PG_VERIFY_SAVE")) { hipMalloc((void**)&vsave, (long)M * N * 4); hipMalloc((void**)&vflag, 8); hipMemset(vflag, 0, 8); }
    launch_gemm_skinny_variant(0, 1, x, W, o0, M, N, K, S);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o0, S, (long)M * N, r0);
    hipMemset(r1, 0, (long)M * N * 4);
    hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, md + 1);       // max |ref|
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(o1, 0xff, (long)S * M * N * 4);
        if (!launch_gemm_skinny_variant(0, variant, x, tiled ? Wt : W, o1, M, N, K, S)) { rc = -1; break; }
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o1, S, (long)M * N, r1);
        hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, md);
        if (vsave) hipLaunchKernelGGL(save_if_bad_kernel, dim3(256), dim3(256), 0, 0, r1, vsave, (long)M * N, md, pattern ? 1e-3f : 2e-3f * 0.047f, vflag, rep);
    }
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) rc = -2;
    float h[2] = {0, 0}; hipMemcpy(h, md, 8, hipMemcpyDeviceToHost);
    if (vsave) {
        int hf[2]; hipMemcpy(hf, vflag, 8, hipMemcpyDeviceToHost);
        if (hf[0]) {
            std::vector<float> a((size_t)M * N), b((size_t)M * N);
            hipMemcpy(a.data(), r0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), vsave, b.size() * 4, hipMemcpyDeviceToHost);
            const float tol = 2e-3f * h[1];
            fprintf(stderr, "[verify] v%d N=%d S=%d first bad launch %d; wrong (row: n-tiles) =", variant, N, S, hf[1]);
            for (int r = 0; r < M; ++r) {
                int cnt = 0, lo = -1, hi = -1;
                for (int ct = 0; ct < N / 16; ++ct) {
                    bool wrong = false;
                    for (int j = 0; j < 16; ++j) wrong |= !(fabsf(a[(long)r * N + ct * 16 + j] - b[(long)r * N + ct * 16 + j]) <= tol);
                    if (wrong) { ++cnt; if (lo < 0) lo = ct; hi = ct; }
                }
                if (cnt) fprintf(stderr, " %d:%d[%d..%d] got-ref=%.6f ref=%.6f", r, cnt, lo, hi, b[(long)r * N + lo * 16] - a[(long)r * N + lo * 16], a[(long)r * N + lo * 16]);
            }
            fprintf(stderr, "\n");
        }
        hipFree(vsave); hipFree(vflag);
    }
    if (maxdiff) *maxdiff = h[0]; if (maxref) *maxref = h[1];
    hipFree(x); hipFree(W); hipFree(Wt); hipFree(o0); hipFree(o1); hipFree(r0); hipFree(r1); hipFree(md);
    return rc;
}

__global__ void save_if_bad_kernel(const float* r1, float* save, long n, const float* md, float tol, int* flag, int rep) {
    if (!(*md > tol) || (flag[0] != 0 && flag[1] != rep)) return;          // first bad launch only (stream order: one launch decides at a time)
    if (blockIdx.x == 0 && threadIdx.x == 0) { flag[1] = rep; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) save[i] = r1[i];
    __syncthreads();
    if (threadIdx.x == 0) atomicExch(&flag[0], 1);
}
// Diagnostic form of the screen: synchronous per-launch compare; reports the number of bad launches and, for the first one,
// up to ``cap`` (row, col) positions of wrong elements (tolerance 2e-3 of max |ref|).  x / W are re-randomised per batch by ``seed``.
extern "C" int pg_bench_skinny_diag(int M, int N, int K, int variant, int S, int reps, unsigned seed, int* bad_launches, int* pos, int cap, int* npos) {
    bf16 *x, *W, *Wt; float *o0, *o1, *r0, *r1, *md;
    hipMalloc((void**)&x, (long)M * K * 2); hipMalloc((void**)&W, (long)N * K * 2); hipMalloc((void**)&Wt, (long)N * K * 2);
    hipMalloc((void**)&o0, (long)S * M * N * 4); hipMalloc((void**)&o1, (long)S * M * N * 4);
    hipMalloc((void**)&r0, (long)M * N * 4); hipMalloc((void**)&r1, (long)M * N * 4); hipMalloc((void**)&md, 8);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, x, (long)M * K, 3u + seed);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, W, (long)N * K, 7u + seed);
    launch_tile_weights(0, W, Wt, N, K);
    launch_gemm_skinny_variant(0, 1, x, W, o0, M, N, K, S);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o0, S, (long)M * N, r0);
    hipMemset(r1, 0, (long)M * N * 4); hipMemset(md, 0, 8);
    hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, md + 1);
    const bool async = getenv("PG_DIAG_ASYNC") != nullptr;
    float h[2] = {0.f, 0.047f};
    if (!(async && getenv("PG_DIAG_NOSYNC"))) hipMemcpy(h, md, 8, hipMemcpyDeviceToHost);       // NOSYNC: the loop follows the reference launches with no host sync (fixed-seed max |ref| = 0.047)
    const float tol = 2e-3f * h[1];
    int rc = 0, bad = 0, np = 0;
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    float *mdr = nullptr, *save = nullptr; int* flag = nullptr;
    if (async) {
        hipMalloc((void**)&mdr, reps * 4); hipMemset(mdr, 0, reps * 4); hipMalloc((void**)&save, (long)M * N * 4); hipMalloc((void**)&flag, 8); hipMemset(flag, 0, 8);
        for (int rep = 0; rep < reps; ++rep) {
            if (getenv("PG_DIAG_SYNCSET")) hipMemset(o1, 0xff, (long)S * M * N * 4); else hipMemsetAsync(o1, 0xff, (long)S * M * N * 4, 0);
            if (!launch_gemm_skinny_variant(0, variant, x, Wt, o1, M, N, K, S)) { rc = -1; break; }
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o1, S, (long)M * N, r1);
            hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, mdr + rep);
            hipLaunchKernelGGL(save_if_bad_kernel, dim3(256), dim3(256), 0, 0, r1, save, (long)M * N, mdr + rep, tol, flag, rep);
        }
        hipDeviceSynchronize();
        std::vector<float> hm(reps); hipMemcpy(hm.data(), mdr, reps * 4, hipMemcpyDeviceToHost);
        int hf[2]; hipMemcpy(hf, flag, 8, hipMemcpyDeviceToHost);
        for (int rep = 0; rep < reps; ++rep) bad += !(hm[rep] <= tol);
        if (bad) {
            hipMemcpy(a.data(), r0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), save, b.size() * 4, hipMemcpyDeviceToHost);
            // summary: (-2, rep) for every bad launch, then one (row, 16-column tile) entry per tile of the first bad launch that holds a wrong element
            for (int rep = 0; rep < reps && np < cap; ++rep) if (!(hm[rep] <= tol)) { pos[np * 2] = -2; pos[np * 2 + 1] = rep; ++np; }
            for (int r = 0; r < M; ++r)
                for (int ct = 0; ct < N / 16 && np < cap; ++ct) {
                    bool wrong = false;
                    for (int j = 0; j < 16; ++j) wrong |= !(fabsf(a[(long)r * N + ct * 16 + j] - b[(long)r * N + ct * 16 + j]) <= tol);
                    if (wrong) { pos[np * 2] = r; pos[np * 2 + 1] = ct; ++np; }
                }
            if (np < cap) { pos[np * 2] = -1; pos[np * 2 + 1] = hf[1]; ++np; }
        }
        hipFree(mdr); hipFree(save); hipFree(flag);
    }
    for (int rep = 0; rep < reps && !async; ++rep) {
        hipMemsetAsync(o1, 0xff, (long)S * M * N * 4, 0); hipMemsetAsync(md, 0, 4, 0);
        if (!launch_gemm_skinny_variant(0, variant, x, Wt, o1, M, N, K, S)) { rc = -1; break; }
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(1024), dim3(256), 0, 0, o1, S, (long)M * N, r1);
        hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, r0, r1, (long)M * N, md);
        hipMemcpy(h, md, 4, hipMemcpyDeviceToHost);
        if (!(h[0] <= tol)) {
            if (bad++ == 0) {
                hipMemcpy(a.data(), r0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, b.size() * 4, hipMemcpyDeviceToHost);
                for (long i = 0; i < (long)M * N && np < cap; ++i)
                    if (!(fabsf(a[i] - b[i]) <= tol)) { pos[np * 2] = (int)(i / N); pos[np * 2 + 1] = (int)(i % N); ++np; }
                if (np < cap) { pos[np * 2] = -1; pos[np * 2 + 1] = rep; ++np; }
            }
        }
    }
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) rc = -2;
    if (bad_launches) *bad_launches = bad; if (npos) *npos = np;
    hipFree(x); hipFree(W); hipFree(Wt); hipFree(o0); hipFree(o1); hipFree(r0); hipFree(r1); hipFree(md);
    return rc;
}

// Per-wave s_memtime stamps of ONE launch of a v4 variant (after warm-up launches on rotating weights):
// stamps_host [nwaves][64] cycle counters; returns the number of waves, or < 0.
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
// GroupNorm statistics from a GEMM / convolution EPILOGUE (conv_halo.hip, round 6: gemm256.hip) against the stand-alone statistics kernel on the tensor
// the same launch stored: (mean, rstd) of every (image, group) must agree (both add fp32 partials in double).  plain = 1: a 1x1 convolution (kind 0,
// K = Cin) like AttnBlock.proj_out; else 3x3 (optionally with the nearest-2x upsample).  res = 1: fp32 residual in the epilogue.  Returns the number of
// splits the launch reported (0: the kernel that took the shape emits no partials), max |mean diff| and max relative rstd diff.
extern "C" int pg_bench_conv_gn_check(int B, int Hi, int Wi, int Cin, int Cout, int up, int plain, int res, int gemm256_mode, int* nsplit_out, float* mean_diff, float* rstd_rel) {
    const int Ho = plain ? Hi : (Hi << up), Wo = plain ? Wi : (Wi << up);
    const long HW = (long)Ho * Wo, M = (long)B * HW;
    const int K = plain ? Cin : 9 * Cin;
    const long a_elems = (long)B * Hi * Wi * Cin;
    bf16 *A, *Wt, *zeros; float *o0, *rsd = nullptr, *wsA, *wsB, *stA, *stB, *gam;
    if (hipMalloc((void**)&A, a_elems * 2) != hipSuccess) return -2;
    hipMalloc((void**)&Wt, (long)Cout * K * 2); hipMalloc((void**)&zeros, 4096); hipMemset(zeros, 0, 4096);
    hipMalloc((void**)&o0, M * Cout * 4);
    hipMalloc((void**)&wsA, (size_t)B * 1024 * 64 * 4); hipMalloc((void**)&wsB, (size_t)B * 1024 * 64 * 4);
    hipMalloc((void**)&stA, (size_t)B * 64 * 4); hipMalloc((void**)&stB, (size_t)B * 64 * 4);
    hipMalloc((void**)&gam, (size_t)Cout * 4); hipMemset(gam, 0, (size_t)Cout * 4);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, A, a_elems, 11u);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, Wt, (long)Cout * K, 5u);
    if (res) { hipMalloc((void**)&rsd, M * Cout * 4); bf16* tmp; hipMalloc((void**)&tmp, M * Cout * 2);
               hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, tmp, M * Cout, 23u);
               launch_convert<float>(0, tmp, 1, rsd, M * Cout); hipDeviceSynchronize(); hipFree(tmp); }
    hipStream_t s; hipStreamCreate(&s);
    GemmA ga; ga.ptr = A; ga.lda = K;
    if (!plain) { ga.kind = 1; ga.Hi = Hi; ga.Wi = Wi; ga.Cin = Cin; ga.up = up; ga.zeros = zeros; }
    int ns = 0;
    ga.gn_part = wsB; ga.gn_nsplit = &ns; ga.gn_hw = plain ? (int)HW : 0;
    GemmEpi e; e.out = o0; e.out_f32 = 1; e.ldc = Cout; e.bias_n = gam; e.residual = rsd; e.res_f32 = 1;
    PgTune tune; tune.diag = diag_hooks(); tune.gemm256 = gemm256_mode; tune.gn_epilogue256 = 1; const PgTune* const saved = pg_tune; pg_tune = &tune;
    hipMemsetAsync(wsB, 0xff, (size_t)B * 1024 * 64 * 4, s);                     // NaN pattern: a slot nobody writes shows up
    launch_gemm<bf16>(s, ga, Wt, K, 0, e, (int)M, Cout, K, 1);
    pg_tune = saved;
    *nsplit_out = ns; *mean_diff = 0.f; *rstd_rel = 0.f;
    int rc = 0;
    if (ns > 0) {
        launch_gn_finalize(s, wsB, stB, nullptr, gam, gam, B, ns, (int)HW, Cout, 1e-6f);
        launch_gn_stats(s, o0, 0, stA, wsA, B, (int)HW, Cout, 1e-6f, nullptr, gam, gam);
        std::vector<float> a((size_t)B * 64), b((size_t)B * 64);
        hipStreamSynchronize(s);
        hipMemcpy(a.data(), stA, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), stB, b.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < a.size(); i += 2) {
            const float dm = fabsf(a[i] - b[i]), dr = fabsf(a[i + 1] - b[i + 1]) / fabsf(a[i + 1]);
            if (!(dm <= *mean_diff)) *mean_diff = dm;                              // NaN-propagating max
            if (!(dr <= *rstd_rel)) *rstd_rel = dr;
        }
    }
    hipStreamSynchronize(s);
    if (hipGetLastError() != hipSuccess) rc = -1;
    hipFree(A); hipFree(Wt); hipFree(zeros); hipFree(o0); if (rsd) hipFree(rsd); hipFree(wsA); hipFree(wsB); hipFree(stA); hipFree(stB); hipFree(gam);
    hipStreamDestroy(s);
    return rc;
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
    // PG_BENCH_CONST=1: constant operands (no bit toggling in the MFMA datapath): separates "the schedule cannot feed the matrix cores" from "the chip clocks
    // down under the power of random-data MFMAs" -- the same kernel, the same instruction stream, a different clock (profiles/r05_c)
    if (getenv("PG_BENCH_CONST")) {
        hipLaunchKernelGGL(fill_const_kernel, dim3(2048), dim3(256), 0, 0, A, a_elems, 0.0078125f);
        hipLaunchKernelGGL(fill_const_kernel, dim3(2048), dim3(256), 0, 0, Wt, (long)N * K, 0.0078125f);
    }
    hipStream_t s; hipStreamCreate(&s);
    GemmA ga; ga.ptr = A; ga.lda = K;
    if (conv) { ga.kind = 1; ga.Hi = Hi; ga.Wi = Wi; ga.Cin = Cin; ga.up = up; ga.zeros = zeros; }
    GemmEpi e; e.out = o0; e.out_f32 = getenv("PG_BENCH_OUT_BF16") ? 0 : 1; e.ldc = N;      // bf16 output: timing only (verify = 0)
    // PG_BENCH_RES=1: fp32 residual (the prefill o / down projections, the ResBlock's second convolution); PG_BENCH_BIAS=1: per-column bias (every convolution,
    // the SigLIP linears): the epilogue's loads and what their waits cost (round 6)
    float *rsd = nullptr, *bia = nullptr;
    if (getenv("PG_BENCH_RES")) { hipMalloc((void**)&rsd, (long)M * N * 4); hipMemset(rsd, 0, (long)M * N * 4); e.residual = rsd; e.res_f32 = 1; }
    if (getenv("PG_BENCH_BIAS")) { hipMalloc((void**)&bia, (long)N * 4); hipMemset(bia, 0, (long)N * 4); e.bias_n = bia; }
    float* gnp = nullptr; int gn_ns = 0;
    if (conv && getenv("PG_BENCH_GN")) {          // GroupNorm partial sums from the halo convolution's epilogue (what the VQ pipeline runs), timing only
        hipMalloc((void**)&gnp, (size_t)B * 8192 * 64 * 4);
        ga.gn_part = gnp; ga.gn_nsplit = &gn_ns;
    }
    PgTune tune; tune.diag = diag_hooks(); const PgTune* const saved = pg_tune; pg_tune = &tune;
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
    hipFree(A); hipFree(Wt); hipFree(zeros); hipFree(o0); if (o1) hipFree(o1); hipFree(md); if (gnp) hipFree(gnp); if (rsd) hipFree(rsd); if (bia) hipFree(bia);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Ordering probe: does a YOUNGER register load retire (vmcnt) before an OLDER LDS-DMA load of the same wave has landed in LDS?
// Per iteration and wave: sentinel -> own LDS slot; LDS-DMA of a cold 1 KiB line set (never touched before); register load of a hot
// line (same address every iteration); s_waitcnt vmcnt(1); ds_read of the slot.  A sentinel read = the counted wait was satisfied by
// the younger load.  mode bit 0: hot load is nt; bit 1: order reversed (register load older, DMA younger, vmcnt(1) then check the
// REGISTER value instead -- the control).
__global__ __launch_bounds__(256) void dma_order_kernel(const unsigned* __restrict__ cold, const unsigned* __restrict__ hot, int iters, long stride_words,
                                                        int mode, unsigned* __restrict__ fails, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4][256];
    __shared__ __attribute__((aligned(16))) unsigned lds2[4][256];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + w;
    unsigned bad = 0, acc = 0;
    for (int it = 0; it < iters; ++it) {
        // mode bit 16: the DMA instruction spans FOUR cold 4 KiB pages (16 lanes each), like the x-tile pieces of the decode GEMM
        const unsigned* src = (mode & 16) ? cold + ((long)it * gridDim.x * 4 + wave) * stride_words * 4 + (long)(l >> 4) * stride_words + (l & 15) * 4
                                          : cold + ((long)it * gridDim.x * 4 + wave) * stride_words + l * 4;
        *(uint4*)&lds[w][l * 4] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4 hv;
        glds16(src, (char*)&lds[w][0]);
        if (mode & 4) { hv = (u32x4){0, 0, 0, 0}; glds16(hot + l * 4, (char*)&lds2[w][0]); }              // younger op is a second LDS-DMA (hot line) instead of a register load
        else if (mode & 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(hv) : "v"(hot + l * 4) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(hv) : "v"(hot + l * 4) : "memory");
        if (mode & 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");      // 8: positive control (no wait at all)
        u32x4 got;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got) : "v"((unsigned)(size_t)&lds[w][l * 4]) : "memory");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(hv) :: "memory");
        const unsigned idx = (mode & 16) ? (unsigned)(((long)it * gridDim.x * 4 + wave) * stride_words * 4 + (long)(l >> 4) * stride_words + (l & 15) * 4)
                                         : (unsigned)(((long)it * gridDim.x * 4 + wave) * stride_words + l * 4);
        bad += (got[0] != idx * 2654435761u) || (got[3] != (idx + 3) * 2654435761u);
        acc += hv[0];
    }
    bad = __builtin_amdgcn_readfirstlane(__popcll(__ballot(bad != 0)) ? 1 : 0) * 0 + bad;
    if (bad) atomicAdd(fails, bad);
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ void fill_hash_kernel(unsigned* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u;
}
extern "C" int pg_bench_dma_order(int blocks, int iters, int mode, unsigned* fails_out, unsigned* checks_out) {
    const long stride_words = 1024;                 // 4 KiB apart: every DMA touches fresh lines (and pages)
    const long n = (long)iters * blocks * 4 * stride_words * ((mode & 16) ? 4 : 1);
    unsigned *cold, *hot, *fails, *sink;
    if (hipMalloc((void**)&cold, n * 4) != hipSuccess) return -3;
    hipMalloc((void**)&hot, 4096); hipMalloc((void**)&fails, 4); hipMalloc((void**)&sink, 4);
    hipLaunchKernelGGL(fill_hash_kernel, dim3(2048), dim3(256), 0, 0, cold, n);
    hipMemset(hot, 1, 4096); hipMemset(fails, 0, 4);
    hipLaunchKernelGGL(dma_order_kernel, dim3(blocks), dim3(256), 0, 0, cold, hot, iters, stride_words, mode, fails, sink);
    hipDeviceSynchronize();
    int rc = hipGetLastError() == hipSuccess ? 0 : -2;
    unsigned h = 0; hipMemcpy(&h, fails, 4, hipMemcpyDeviceToHost);
    if (fails_out) *fails_out = h; if (checks_out) *checks_out = (unsigned)((long)blocks * 4 * 64 * iters);
    hipFree(cold); hipFree(hot); hipFree(fails); hipFree(sink);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Semantics probe for ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds lds[i] = i (16-bit); lane l reads at byte address
// addr[l]; out[l*4 + j] = element j of the returned 64 bits.
__global__ void tr16_probe_kernel(const int* __restrict__ addr, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(v.x & 0xffff); out[threadIdx.x * 4 + 1] = (unsigned short)(v.x >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v.y & 0xffff); out[threadIdx.x * 4 + 3] = (unsigned short)(v.y >> 16);
}
extern "C" int pg_bench_tr16_probe(const int* addr_host, unsigned short* out_host) {
    int* a; unsigned short* o;
    hipMalloc((void**)&a, 64 * 4); hipMalloc((void**)&o, 256 * 2);
    hipMemcpy(a, addr_host, 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, 0, a, o);
    hipDeviceSynchronize();
    const int rc = hipGetLastError() == hipSuccess ? 0 : -2;
    hipMemcpy(out_host, o, 256 * 2, hipMemcpyDeviceToHost);
    hipFree(a); hipFree(o);
    return rc;
}


// ------------------------------------------------------------------------------- weight-stream kernel (round 4 run-ahead prefetcher; now only the background-load stressor)
// One wave = one stream of 1 KiB wave-loads (16 B per lane, contiguous), DEPTH of them in flight, data discarded.  Trigger k of step st
// (= the (first + k * stride)-th ticket of that step, i.e. the norm launch behind o_proj of layer k) releases layer k's list: gate|up,
// down, the NEXT layer's qkv, in that order -- the order the main stream consumes them.  A piece index i of a matrix maps to
// region i % R, piece i / R, so every consumer block's region gets its first pieces first.  Every spin is bounded (wall clock): a
// prefetcher that sees no ticket for 30 ms exits; the main stream never waits for this kernel's data.
template <int DEPTH, int NT, int REGS = 0>
__global__ __launch_bounds__(128) void weight_prefetch_kernel(const PfLayer* __restrict__ plan, int n_layers, const uint32_t* prog, int steps,
                                                              int per_step, int first, int stride, uint32_t* stats) {
    __shared__ __attribute__((aligned(16))) char sink[2][1024];
    const int lane = threadIdx.x & 63;
    const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.x * (blockDim.x >> 6);
    unsigned done_layers = 0, skipped = 0;
    const long total = (long)steps * n_layers;
    long k = 0;                                                 // global trigger index = st * n_layers + layer
    unsigned long long t_last = wall_clock64();
    while (k < total) {
        const int st = (int)(k / n_layers), l = (int)(k % n_layers);
        const unsigned target = (unsigned)st * per_step + first + l * stride;
        unsigned p = 0;
        if (per_step > 0)                                       // per_step == 0: free-running stressor (tools/sk4_load_stress.py), no tickets
        for (;;) {                                              // relaxed poll by every lane of one load (a wave-uniform scalar would be cached)
            p = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p >= target) break;
            __builtin_amdgcn_s_sleep(16);
            // no ticket for 30 ms (100 MHz clock): the loop is over or was aborted.  Before the FIRST ticket the bound is 2 ms: a first ticket that late
            // means this kernel sits on the decode stream's own hardware queue and is what keeps the loop from starting -- leave at once
            if (wall_clock64() - t_last > (k == 0 ? 200000ull : 3000000ull)) {
                if (stats && threadIdx.x == 0 && blockIdx.x == 0) { stats[0] = done_layers; stats[1] = skipped; stats[2] = 1; }
                return;
            }
        }
        t_last = wall_clock64();
        // lagging: when the main stream is already past LATER triggers, jump to the newest one (its weights are what is needed next)
        if (per_step > 0) {
            const unsigned into = p - (unsigned)st * per_step;                      // tickets of this step seen so far (may exceed per_step)
            long knew = k;
            if (into >= (unsigned)per_step) knew = (long)(st + into / per_step) * n_layers;       // at least one whole step ahead: restart at that step's layer 0 trigger
            else if (into >= (unsigned)(first + stride)) { const int lmax = (int)((into - first) / stride); knew = (long)st * n_layers + (lmax < n_layers ? lmax : n_layers - 1); }
            if (knew > k) { skipped += (unsigned)(knew - k); k = knew; continue; }
        }
        const PfLayer& L = plan[l];
#pragma unroll 1
        for (int mi = 0; mi < 4; ++mi) {
            const char* base = (const char*)L.m[mi].base;
            const unsigned npieces = L.m[mi].kib, R = L.m[mi].regions ? L.m[mi].regions : 1u;
            if (!base) continue;
            const unsigned per_region = npieces / R;                                 // pieces beyond R * per_region (none for the decode shapes) are left cold
            unsigned i = wid;
            const unsigned lim = per_region * R;
            if constexpr (REGS) {       // stressor variant: plain register loads in groups of 8 (no LDS-DMA from this kernel)
                for (; i < lim; i += nw * 8) {
                    u32x4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned ii = i + u * nw < lim ? i + u * nw : i;
                        const char* ptr = base + ((size_t)(ii % R) * per_region + ii / R) * 1024 + lane * 16;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(t[u]) : "v"(ptr) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(t[u]));
                }
            } else
            for (; i < lim; i += nw) {
                const unsigned r = i % R, j = i / R;
                const char* ptr = base + ((size_t)r * per_region + j) * 1024 + lane * 16;
                // LDS-DMA into a 1 KiB per-wave sink nobody reads: no VGPR is written, so nothing the compiler re-uses can be overwritten by
                // a load that lands later (a register-destination asm load whose result is "dead" gets its register recycled while in flight)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)ptr, (lds_ptr_t)sink[threadIdx.x >> 6], 16, 0, NT ? 2 : 0);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ++done_layers;
        ++k;
    }
    if (stats && threadIdx.x == 0 && blockIdx.x == 0) { stats[0] = done_layers; stats[1] = skipped; stats[2] = 0; }
}
void launch_weight_prefetch(hipStream_t s, const PfLayer* plan_dev, int n_layers, const uint32_t* prog, int steps, int per_step,
                            int first, int stride, int blocks, int depth, int nt, uint32_t* stats) {
    if (steps <= 0 || n_layers <= 0) return;
    dim3 g(blocks), b(128);
#define PG_PF(D, N) hipLaunchKernelGGL((weight_prefetch_kernel<D, N>), g, b, 0, s, plan_dev, n_layers, prog, steps, per_step, first, stride, stats)
    if (depth < 0) { hipLaunchKernelGGL((weight_prefetch_kernel<8, 0, 1>), g, b, 0, s, plan_dev, n_layers, prog, steps, per_step, first, stride, stats); return; }
    if (nt) { if (depth >= 32) PG_PF(32, 1); else if (depth >= 16) PG_PF(16, 1); else PG_PF(8, 1); }
    else { if (depth >= 32) PG_PF(32, 0); else if (depth >= 16) PG_PF(16, 0); else PG_PF(8, 0); }
#undef PG_PF
}


// Background memory load for hazard screens (tools/sk4_load_stress.py): the run-ahead weight stream kernel free-running over a private
// ``mb`` MB buffer on its own stream while the caller launches the kernel under test.  mode 0: LDS-DMA sink, 1: LDS-DMA nt, 2: register loads.
static hipStream_t g_bg_stream = nullptr; static char* g_bg_buf = nullptr; static PfLayer* g_bg_plan = nullptr; static uint32_t* g_bg_words = nullptr;
extern "C" int pg_bench_background(int mb, int passes, int blocks, int depth, int mode) {
    if (!g_bg_stream) {
        if (hipStreamCreateWithFlags(&g_bg_stream, hipStreamNonBlocking) != hipSuccess) return -2;
        if (hipMalloc((void**)&g_bg_plan, sizeof(PfLayer)) != hipSuccess || hipMalloc((void**)&g_bg_words, 64) != hipSuccess) return -2;
        hipMemset(g_bg_words, 0, 64);
    }
    static int cur_mb = 0;
    if (mb != cur_mb) {
        if (g_bg_buf) { hipStreamSynchronize(g_bg_stream); hipFree(g_bg_buf); g_bg_buf = nullptr; }
        if (hipMalloc((void**)&g_bg_buf, (size_t)mb << 20) != hipSuccess) return -2;
        hipMemset(g_bg_buf, 1, (size_t)mb << 20);
        cur_mb = mb;
    }
    PfLayer pl{}; pl.m[0].base = g_bg_buf; pl.m[0].kib = (uint32_t)mb * 1024u; pl.m[0].regions = 64;
    hipMemcpy(g_bg_plan, &pl, sizeof(pl), hipMemcpyHostToDevice);
    launch_weight_prefetch(g_bg_stream, g_bg_plan, 1, g_bg_words, passes, 0, 0, 0, blocks, mode == 2 ? -1 : depth, mode == 1, g_bg_words + 4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int pg_bench_background_done() {        // 1 when the background kernel has finished (non-blocking)
    if (!g_bg_stream) return 1;
    return hipStreamQuery(g_bg_stream) == hipSuccess ? 1 : 0;
}
extern "C" int pg_bench_background_join() { return g_bg_stream && hipStreamSynchronize(g_bg_stream) != hipSuccess ? -2 : 0; }


// Exhaustive check of the hardware f32 -> bf16 conversion (common.h) against the software round-to-nearest-even of rounds 1-3: all 2^32 bit
// patterns; NaN inputs only have to stay NaN.  Returns the number of differing non-NaN patterns (expected 0) and of NaNs that did not stay NaN.
__global__ void bf16_cvt_check_kernel(unsigned long long* out) {
    unsigned long long bad = 0, badnan = 0;
    const unsigned long long n = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float f = __uint_as_float((uint32_t)i);
        const uint32_t hw = f32_to_bf16_bits(f), sw = f32_to_bf16_bits_sw(f);
        const uint32_t pk = pack_bf16x2(f, -f);
        if (((uint32_t)i & 0x7fffffffu) > 0x7f800000u) { if ((hw & 0x7fffu) <= 0x7f80u) ++badnan; }
        else { if (hw != sw || (pk & 0xffffu) != sw || (pk >> 16) != (sw ^ 0x8000u)) ++bad; }
    }
    if (bad) atomicAdd(out, bad);
    if (badnan) atomicAdd(out + 1, badnan);
}
extern "C" int pg_bench_bf16_cvt_check(unsigned long long* mismatches, unsigned long long* nan_lost) {
    unsigned long long* d;
    if (hipMalloc((void**)&d, 16) != hipSuccess) return -2;
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(bf16_cvt_check_kernel, dim3(4096), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    const int rc = hipGetLastError() == hipSuccess ? 0 : -2;
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    hipFree(d);
    if (mismatches) *mismatches = h[0];
    if (nan_lost) *nan_lost = h[1];
    return rc;
}


// ------------------------------------------------------------------------------- fused-norm producer (round 5 experiment)
// Pair A: production split-K GEMM (tiled weights) + rmsnorm512 (reduce + residual + norm).  B: the EPI 5 producer alone (reduce + residual +
// bf16(x w) + per-row sums of squares in its tail).  Times both over rotating weights and checks B against A's arithmetic: x_new bit-identical,
// xw == bf16(x_new * w), ssq within fixed-point resolution of sum x_new^2.
__global__ void fused_check_kernel(const float* xa, const float* xb, const bf16* xw, const bf16* w, const unsigned long long* ssq, int M, int N, unsigned* bad) {
    const int m = blockIdx.x;
    double q = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float a = xa[(long)m * N + n], b = xb[(long)m * N + n];
        if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(bad, 1u);
        const uint32_t want = f32_to_bf16_bits(ET<bf16>::ld(w + n) * b);
        if (want != (uint32_t)(*(const unsigned short*)(xw + (long)m * N + n))) atomicAdd(bad + 1, 1u);
        q += (double)b * b;
    }
    __shared__ double sq[256];
    sq[threadIdx.x] = q; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sq[threadIdx.x] += sq[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const double got = (double)ssq[m] / 268435456.0;
        if (fabs(got - sq[0]) > 1e-5 * sq[0] + 1e-6) atomicAdd(bad + 2, 1u);
    }
}
extern "C" int pg_bench_fused_norm(int M, int N, int K, int S, int iters, float* us_pair, float* us_gemm, float* us_fused, unsigned* bad_out) {
    const long wbytes = (long)N * K * 2;
    int nbuf = (int)((600L << 20) / wbytes) + 1; if (nbuf > 32) nbuf = 32; if (nbuf < 2) nbuf = 2;
    std::vector<bf16*> Ws(nbuf);
    bf16* Wrow; hipMalloc((void**)&Wrow, wbytes);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(2048), dim3(256), 0, 0, Wrow, (long)N * K, 7u);
    for (auto& p : Ws) { if (hipMalloc((void**)&p, wbytes) != hipSuccess) return -2; launch_tile_weights(0, Wrow, p, N, K); }
    bf16 *x, *wn, *xn, *xw; float *slabs, *xa, *xb, *x0; unsigned long long* ssq; unsigned *ticket, *bad; SkFuse* site;
    hipMalloc((void**)&x, (long)M * K * 2); hipMalloc((void**)&wn, N * 2); hipMalloc((void**)&xn, (long)M * N * 2); hipMalloc((void**)&xw, (long)M * N * 2);
    hipMalloc((void**)&slabs, (long)S * M * N * 4); hipMalloc((void**)&xa, (long)M * N * 4); hipMalloc((void**)&xb, (long)M * N * 4); hipMalloc((void**)&x0, (long)M * N * 4);
    hipMalloc((void**)&ssq, M * 8); hipMalloc((void**)&ticket, 4096 * 4); hipMalloc((void**)&bad, 16); hipMalloc((void**)&site, sizeof(SkFuse));
    hipMemset(ticket, 0, 4096 * 4); hipMemset(bad, 0, 16);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, x, (long)M * K, 3u);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(16), dim3(256), 0, 0, wn, (long)N, 9u);
    hipLaunchKernelGGL(fill_bf16_kernel, dim3(256), dim3(256), 0, 0, (bf16*)x0, (long)M * N * 2, 13u);      // fp32 residual: arbitrary finite bit patterns of two bf16 halves
    SkFuse hs{xb, wn, xw, ssq, ticket, nullptr};
    hipMemcpy(site, &hs, sizeof(hs), hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    PgTune tune; const PgTune* const saved = pg_tune; pg_tune = &tune;
    int rc = 0;
    // ---- correctness: 20 rounds, fresh residual each
    for (int r = 0; r < 20 && rc == 0; ++r) {
        hipMemcpyAsync(xa, x0, (long)M * N * 4, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(xb, x0, (long)M * N * 4, hipMemcpyDeviceToDevice, s);
        hipMemsetAsync(ssq, 0, M * 8, s);
        if (!launch_gemm_skinny_tiled_only(s, x, Ws[r % nbuf], slabs, M, N, K, S)) { rc = -1; break; }
        launch_rmsnorm<bf16>(s, xa, slabs, S, (long)M * N, wn, xn, M, N, 1e-6f);
        hipMemsetAsync(slabs, 0xff, (long)S * M * N * 4, s);
        if (!launch_gemm_skinny_fused_norm(s, x, Ws[r % nbuf], slabs, M, N, K, S, site)) { rc = -1; break; }
        hipLaunchKernelGGL(fused_check_kernel, dim3(M), dim3(256), 0, s, xa, xb, xw, wn, ssq, M, N, bad);
    }
    hipStreamSynchronize(s);
    hipMemcpy(bad_out, bad, 12, hipMemcpyDeviceToHost);
    // ---- timing
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    auto time_it = [&](int mode, float* out) {
        for (int it = -5; it < iters; ++it) {
            if (it == 0) hipEventRecord(e0, s);
            bf16* Wt = Ws[(it + 5) % nbuf];
            if (mode == 0) { launch_gemm_skinny_tiled_only(s, x, Wt, slabs, M, N, K, S); launch_rmsnorm<bf16>(s, xa, slabs, S, (long)M * N, wn, xn, M, N, 1e-6f); }
            else if (mode == 1) launch_gemm_skinny_tiled_only(s, x, Wt, slabs, M, N, K, S);
            else launch_gemm_skinny_fused_norm(s, x, Wt, slabs, M, N, K, S, site);
        }
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); *out = ms * 1000.f / iters;
    };
    if (rc == 0) { time_it(0, us_pair); time_it(1, us_gemm); time_it(2, us_fused); }
    if (hipGetLastError() != hipSuccess) rc = -2;
    pg_tune = saved;
    for (auto p : Ws) hipFree(p);
    hipFree(Wrow); hipFree(x); hipFree(wn); hipFree(xn); hipFree(xw); hipFree(slabs); hipFree(xa); hipFree(xb); hipFree(x0); hipFree(ssq); hipFree(ticket); hipFree(bad); hipFree(site);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return rc;
}


// ------------------------------------------------------------------------------- MFMA-only probe (round 5)
// What dense bf16 rate do the matrix cores SUSTAIN on this chip with no memory traffic at all?  Every wave keeps NACC independent accumulators and
// issues MFMAs back to back from registers; operands are random (bit toggling in the datapath) or constant.  mode 0: v_mfma_f32_16x16x32_bf16,
// mode 1: v_mfma_f32_32x32x16_bf16.  The prefill GEMMs run at 1.0-1.09 PF on random data and 1.3 PF on constant data (profiles/r05_c): this
// probe says how much of that gap is the chip's power / clock behaviour rather than the kernel's schedule.
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* __restrict__ out, int iters, int constant) {
    const int l = threadIdx.x & 63;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 ua, ub;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t h = (uint32_t)(l * 131 + i * 17 + j * 7 + blockIdx.x * 977 + threadIdx.x) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            uint32_t g = h * 2246822519u + 12345u; g ^= g >> 16;
            // two bf16 per dword in [-1, 1): sign + exponent 0x3f00..0x3f7f + random mantissa
            const uint32_t lo = constant ? 0x3c00u : ((h & 0x8000u) | 0x3f00u | (h & 0x7fu)), hi = constant ? 0x3c00u : ((g & 0x8000u) | 0x3f00u | (g & 0x7fu));
            ua[j] = lo | (hi << 16);
            const uint32_t lo2 = constant ? 0x3c00u : (((h >> 16) & 0x8000u) | 0x3f00u | ((h >> 8) & 0x7fu)), hi2 = constant ? 0x3c00u : (((g >> 16) & 0x8000u) | 0x3f00u | ((g >> 8) & 0x7fu));
            ub[j] = lo2 | (hi2 << 16);
        }
        a[i] = *(const bf16x8*)&ua; b[i] = *(const bf16x8*)&ub;
    }
    if constexpr (MODE == 0) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 123.456f) out[threadIdx.x] = s;
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[i & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        if (s == 123.456f) out[threadIdx.x] = s;
    }
}

// LDS-fed variant of the probe (mode 2): ONE wave per SIMD, per-wave tile 128 x 128 = 4 x 4 MFMA tiles of 32x32 (256 accumulator registers -> AGPRs),
// per k-step (16) 4 A + 4 B fragments read from LDS with ds_read_b128 (double-buffered in registers: the reads of step s+1 are issued before the
// 16 MFMAs of step s), LDS filled once with random bf16.  No global traffic, no barriers in the loop: the ceiling of a "big per-wave tile" GEMM main loop.
__global__ __launch_bounds__(256) void mfma_lds_fed_kernel(float* __restrict__ out, int iters, int constant) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KiB: A [256 rows][128 B] | B [256 rows][128 B]
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 256) {
        uint32_t h = (uint32_t)(i * 2654435761u + blockIdx.x * 977u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        ((uint32_t*)smem)[i] = constant ? 0x3c003c00u : (((h & 0x8000u) | 0x3f00u | (h & 0x7fu)) | ((((h >> 16) & 0x8000u) | 0x3f00u | ((h >> 8) & 0x7fu)) << 16));
    }
    __syncthreads();
    const int wr = w >> 1, wc = w & 1, lr = l & 31, kg = l >> 5;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 af[2][4], bfr[2][4];
    auto rd = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ra = wr * 128 + t * 32 + lr, rb = wc * 128 + t * 32 + lr;
            const int c = ks * 2 + kg;
            af[buf][t] = *(const bf16x8*)(smem + ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4));
            bfr[buf][t] = *(const bf16x8*)(smem + 32768 + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4));
        }
    };
    rd(0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            rd((ks + 1) & 1, (ks + 1) & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 123.456f) out[tid] = s;
}
// mode: 0 / 1 = MFMA shape; waves_per_simd 1..2 (block = 256 threads = 4 waves; grid = 256 CUs x waves_per_simd blocks); returns TFLOP/s over `ms_target` ms of work
extern "C" int pg_bench_mfma_peak(int mode, int waves_per_simd, int constant, int iters, float* tflops_out, float* ms_out) {
    float* out; if (hipMalloc((void**)&out, 4096) != hipSuccess) return -2;
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(256 * waves_per_simd), block(256);
    if (mode == 2) (void)hipFuncSetAttribute((const void*)mfma_lds_fed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(mfma_peak_kernel<0>, grid, block, 0, s, out, iters, constant);
        else if (mode == 1) hipLaunchKernelGGL(mfma_peak_kernel<1>, grid, block, 0, s, out, iters, constant);
        else hipLaunchKernelGGL(mfma_lds_fed_kernel, dim3(256), block, 65536, s, out, iters, constant);
    };
    launch(); launch();
    hipEventRecord(e0, s);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // FLOPs per wave per iteration: mode 0: 16 MFMAs x 16*16*32*2; mode 1: 8 MFMAs x 32*32*16*2 -- both 262 144
    // mode 2: 256 blocks x 4 waves x iters x 4 k-steps x 16 MFMAs x 32*32*16*2
    const double fl = mode == 2 ? (double)reps * 256 * 4.0 * iters * 4 * 16 * 32768.0 : (double)reps * grid.x * 4.0 * iters * 262144.0;
    *tflops_out = (float)(fl / (ms * 1e-3) / 1e12); *ms_out = ms / reps;
    const int rc = hipGetLastError() == hipSuccess ? 0 : -2;
    hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return rc;
}

