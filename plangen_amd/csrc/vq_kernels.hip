// VQ-16 tokenizer kernels other than the implicit-GEMM convolutions (those are gemm.hip's
// ConvLoader): code -> post_quant table gather, GroupNorm(32) statistics and apply(+swish),
// row softmax for the single-head AttnBlock, the Cout=3 output conv, the Cin=3 input conv and
// the nearest-code argmin of the encoder.  Activations are NHWC (channels contiguous) so a
// pixel's channels are one coalesced run.  Reference: three_party/Janus/janus/models/vq_model.py.
#include <type_traits>
#include "kernels.h"

// ------------------------------------------------------------------------------- gather
template <typename T>
__global__ void vq_gather_kernel(const T* __restrict__ table, const int32_t* __restrict__ codes,
                                 T* __restrict__ out, int C, int vocab) {
    const int p = blockIdx.x;
    int c = codes[p];
    c = c < 0 ? 0 : (c >= vocab ? vocab - 1 : c);
    for (int i = threadIdx.x; i < C; i += blockDim.x) out[(long)p * C + i] = table[(long)c * C + i];
}
template <typename T>
void launch_vq_gather(hipStream_t s, const T* table, const int32_t* codes, T* out, int n, int C, int vocab) {
    if (n <= 0) return;
    hipLaunchKernelGGL(vq_gather_kernel<T>, dim3(n), dim3(64), 0, s, table, codes, out, C, vocab);
}
template void launch_vq_gather<float>(hipStream_t, const float*, const int32_t*, float*, int, int, int);
template void launch_vq_gather<bf16>(hipStream_t, const bf16*, const int32_t*, bf16*, int, int, int);

// ------------------------------------------------------------------------------- GroupNorm stats
// grid (nsplit, B), 256 threads.  Thread -> (pixel lane, 16-byte channel vector); per-thread
// per-channel fp32 partial sums, deterministic LDS tree (no atomics), per-(split, group)
// partials to ws, then a finalize kernel combines the splits in double.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ ws,
                                                      int HW, int C, int nsplit) {
    constexpr int EPV = ET<T>::EPV;
    extern __shared__ float lds[];                    // [2][PL][C]
    const int b = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
    const int VPP = C / EPV;                          // vectors per pixel (divides 256 or C > 256*EPV handled by loop)
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    const int PL = 256 / VPP > 0 ? 256 / VPP : 1;     // pixel lanes
    const int vi = tid % VPP, pl = tid / VPP;
    float s[EPV], q[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (pl < PL && VPP <= 256) {
        const T* xb = x + (long)b * HW * C + vi * EPV;
        for (int p = p0 + pl; p < p1; p += PL) {
            const u32x4 v = *(const u32x4*)(xb + (long)p * C);
            float f[EPV]; ET<T>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
        }
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            lds[(0 * PL + pl) * C + vi * EPV + e] = s[e];
            lds[(1 * PL + pl) * C + vi * EPV + e] = q[e];
        }
    }
    __syncthreads();
    // per-channel reduce over pixel lanes, then per-group over its channels (thread per group)
    const int cpg = C / 32;
    if (tid < 32) {
        float gs = 0.f, gq = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c)
            for (int j = 0; j < PL; ++j) { gs += lds[(0 * PL + j) * C + c]; gq += lds[(1 * PL + j) * C + c]; }
        float* o = ws + (((long)b * nsplit + sp) * 32 + tid) * 2;
        o[0] = gs; o[1] = gq;
    }
}
// Combine the split partials (double) and emit per-(image, channel) affine coefficients:
// y = x * coef[b][c][0] + coef[b][c][1]  with  a = rstd*gamma, sh = beta - mean*rstd*gamma.
__global__ void gn_finalize_kernel(const float* __restrict__ ws, float* __restrict__ stats, float* __restrict__ coef,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, int nsplit,
                                   double cnt, float eps, int C) {
    __shared__ float s_mean[32], s_rstd[32];
    const int b = blockIdx.x, t = threadIdx.x;
    // 8 threads per group walk the partials (up to 1024 conv tiles per image) and meet in three shuffles:
    // fixed order, double accumulation
    const int grp = t >> 3, sub = t & 7;
    double s = 0.0, q = 0.0;
    if (grp < 32)
        for (int sp = sub; sp < nsplit; sp += 8) {
            const float* o = ws + (((long)b * nsplit + sp) * 32 + grp) * 2;
            s += (double)o[0]; q += (double)o[1];
        }
#pragma unroll
    for (int o_ = 1; o_ < 8; o_ <<= 1) { s += __shfl_xor(s, o_, 64); q += __shfl_xor(q, o_, 64); }
    if (grp < 32 && sub == 0) {
        const int t = grp;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
        stats[((long)b * 32 + t) * 2 + 0] = mf; stats[((long)b * 32 + t) * 2 + 1] = rf;
        s_mean[t] = mf; s_rstd[t] = rf;
    }
    __syncthreads();
    if (coef) {
        const int cpg = C / 32;
        for (int c = t; c < C; c += blockDim.x) {
            const int g = c / cpg;
            const float a = s_rstd[g] * gamma[c];
            coef[((long)b * C + c) * 2 + 0] = a;
            coef[((long)b * C + c) * 2 + 1] = beta[c] - s_mean[g] * a;
        }
    }
}
void launch_gn_finalize(hipStream_t s, const float* ws, float* stats, float* coef, const float* gamma, const float* beta, int B,
                        int nsplit, int HW, int C, float eps) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, ws, stats, coef, gamma, beta, nsplit, (double)HW * (C / 32), eps, C);
}
// Splits per image.  Round 6: one split per 64 pixels (was per 1 024): at 24^2 a split per 1 024 pixels meant ONE block per image -- 64 blocks walking 288 dependent
// iterations each, 77 us for 75 MB (1.0 TB/s); with 9 splits per image the same pass is 576 blocks of 32 iterations.  gn_finalize_kernel adds the splits in double.
static int gn_nsplit(int HW) { int n = (HW + 63) / 64; return n < 1 ? 1 : (n > 256 ? 256 : n); }
void launch_gn_stats(hipStream_t s, const void* x, int is_bf16, float* stats, float* ws, int B, int HW, int C, float eps,
                     float* coef, const float* gamma, const float* beta) {
    const int nsplit = gn_nsplit(HW);
    const int EPV = is_bf16 ? 8 : 4;
    const int VPP = C / EPV, PL = 256 / VPP > 0 ? 256 / VPP : 1;
    const size_t lds = (size_t)2 * PL * C * sizeof(float);
    if (is_bf16) hipLaunchKernelGGL(gn_stats_kernel<bf16>, dim3(nsplit, B), dim3(256), lds, s, (const bf16*)x, ws, HW, C, nsplit);
    else hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nsplit, B), dim3(256), lds, s, (const float*)x, ws, HW, C, nsplit);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, ws, stats, coef, gamma, beta, nsplit, (double)HW * (C / 32), eps, C);
}

// y = swish?(x * a[b][c] + sh[b][c]); grid (chunks, B); input TI (fp32 skip stream or T), output TO.
template <typename TI, typename TO, bool PRECISE>
__global__ __launch_bounds__(256) void gn_apply_kernel(const TI* __restrict__ x, const float* __restrict__ coef,
                                                      TO* __restrict__ y, int C, int swish, long vec_per_img) {
    constexpr int EPV = ET<TO>::EPV;
    const int b = blockIdx.y;
    const int vpc = C / EPV;                              // vectors per pixel
    const TI* xb = x + (long)b * vec_per_img * EPV;
    TO* yb = y + (long)b * vec_per_img * EPV;
    const float* cf = coef + (long)b * C * 2;
    for (long vi = (long)blockIdx.x * 256 + threadIdx.x; vi < vec_per_img; vi += (long)gridDim.x * 256) {
        const int c0 = (int)(vi % vpc) * EPV;
        const long e0 = vi * EPV;
        float f[EPV];
        if constexpr (sizeof(TI) == sizeof(TO)) {
            const u32x4 v = *(const u32x4*)(xb + e0);
            ET<TI>::unpack(v, f);
        } else {
            const u32x4 v0 = *(const u32x4*)(xb + e0), v1 = *(const u32x4*)(xb + e0 + 4);
            ET<float>::unpack(v0, f); ET<float>::unpack(v1, f + 4);
        }
#pragma unroll
        for (int e = 0; e < EPV; e += 2) {
            const f32x4 ab = *(const f32x4*)(cf + (c0 + e) * 2);          // a0 sh0 a1 sh1
            float t0 = fmaf(f[e], ab.x, ab.y), t1 = fmaf(f[e + 1], ab.z, ab.w);
            if (swish) {
                t0 = PRECISE ? t0 / (1.f + expf(-t0)) : t0 * __frcp_rn(1.f + __expf(-t0));
                t1 = PRECISE ? t1 / (1.f + expf(-t1)) : t1 * __frcp_rn(1.f + __expf(-t1));
            }
            f[e] = t0; f[e + 1] = t1;
        }
        *(u32x4*)(yb + e0) = ET<TO>::pack(f);
    }
}
// Round 6: the same pass when a pixel's vectors divide the block (C / EPV | 256: every GroupNorm of the VQ model): a thread's channel block is then the SAME in every
// grid-stride step, so its EPV (scale, shift) pairs are loaded once instead of four 16-byte coefficient loads + a modulo per vector, and four vectors are in flight
// per step (all loads, then the arithmetic, then the stores).  Same arithmetic per element as gn_apply_kernel -> identical values.
template <typename TI, typename TO, bool PRECISE>
__global__ __launch_bounds__(256) void gn_apply4_kernel(const TI* __restrict__ x, const float* __restrict__ coef, TO* __restrict__ y, int C, int swish, long vec_per_img) {
    constexpr int EPV = ET<TO>::EPV, U = 4;
    const int b = blockIdx.y;
    const int vpc = C / EPV;
    const TI* xb = x + (long)b * vec_per_img * EPV;
    TO* yb = y + (long)b * vec_per_img * EPV;
    const int c0 = (int)(threadIdx.x % vpc) * EPV;                  // blockIdx.x * 256 and the grid stride are multiples of vpc
    const float* cf = coef + (long)b * C * 2 + c0 * 2;
    float ca[EPV], cs[EPV];
#pragma unroll
    for (int e = 0; e < EPV; e += 2) { const f32x4 ab = *(const f32x4*)(cf + e * 2); ca[e] = ab.x; cs[e] = ab.y; ca[e + 1] = ab.z; cs[e + 1] = ab.w; }
    const long stride = (long)gridDim.x * 256;
    for (long v0 = (long)blockIdx.x * 256 + threadIdx.x; v0 < vec_per_img; v0 += U * stride) {
        u32x4 raw[U][sizeof(TI) == sizeof(TO) ? 1 : 2];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long vi = v0 + k * stride;
            if (vi < vec_per_img) {
                raw[k][0] = *(const u32x4*)(xb + vi * EPV);
                if constexpr (sizeof(TI) != sizeof(TO)) raw[k][1] = *(const u32x4*)(xb + vi * EPV + 4);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long vi = v0 + k * stride;
            if (vi < vec_per_img) {
                float f[EPV];
                if constexpr (sizeof(TI) == sizeof(TO)) ET<TI>::unpack(raw[k][0], f);
                else { ET<float>::unpack(raw[k][0], f); ET<float>::unpack(raw[k][1], f + 4); }
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float t = fmaf(f[e], ca[e], cs[e]);
                    if (swish) t = PRECISE ? t / (1.f + expf(-t)) : t * __frcp_rn(1.f + __expf(-t));
                    f[e] = t;
                }
                *(u32x4*)(yb + vi * EPV) = ET<TO>::pack(f);
            }
        }
    }
}
template <typename TI, typename TO>
void launch_gn_apply(hipStream_t s, const TI* x, const float* coef, TO* y, int B, int HW, int C, int swish) {
    const long vec_per_img = (long)HW * C / ET<TO>::EPV;
    int blocks = (int)((vec_per_img + 255) / 256);
    const int cap = 4096 / (B > 0 ? B : 1) + 1;
    if (blocks > cap) blocks = cap;
    constexpr bool PRECISE = std::is_same<TO, float>::value;
    const int vpc = C / ET<TO>::EPV;
    if (vpc > 0 && C % ET<TO>::EPV == 0 && 256 % vpc == 0)
        hipLaunchKernelGGL((gn_apply4_kernel<TI, TO, PRECISE>), dim3(blocks, B), dim3(256), 0, s, x, coef, y, C, swish, vec_per_img);
    else
        hipLaunchKernelGGL((gn_apply_kernel<TI, TO, PRECISE>), dim3(blocks, B), dim3(256), 0, s, x, coef, y, C, swish, vec_per_img);
}
template void launch_gn_apply<float, float>(hipStream_t, const float*, const float*, float*, int, int, int, int);
template void launch_gn_apply<float, bf16>(hipStream_t, const float*, const float*, bf16*, int, int, int, int);
template void launch_gn_apply<bf16, bf16>(hipStream_t, const bf16*, const float*, bf16*, int, int, int, int);

// ------------------------------------------------------------------------------- row softmax
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int rows, int n, float scale) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (r >= rows) return;
    const float* xr = x + (long)r * n;
    float mx = -INFINITY;
    for (int i = l; i < n; i += 64) mx = fmaxf(mx, xr[i] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = l; i < n; i += 64) sum += expf(xr[i] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = l; i < n; i += 64) ET<T>::st(y + (long)r * n + i, expf(xr[i] * scale - mx) * inv);
}
template <typename T>
void launch_softmax_rows(hipStream_t s, const float* x, T* y, int rows, int n, float scale) {
    hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, rows, n, scale);
}
template void launch_softmax_rows<float>(hipStream_t, const float*, float*, int, int, float);
template void launch_softmax_rows<bf16>(hipStream_t, const float*, bf16*, int, int, float);

// ------------------------------------------------------------------------------- conv_out (Cout tiny)
// Cout <= 4 (conv_out 128 -> 3).  A block owns a strip of 64 output pixels of one image row:
// the 3 x 66-pixel input halo strip is staged in LDS with coalesced 16-byte loads (each input
// pixel is fetched 3x instead of 9x, and never as 64 different cache lines per instruction),
// pixel stride padded by 16 B (conflict-free ds_read_b128); 4 threads per pixel split the
// channels, weights sit in LDS as fp32; NHWC in, NCHW out.
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ bias, void* __restrict__ out,
                                                           int out_bf16, int H, int W, int Cin, int Cout) {
    constexpr int EPV = ET<T>::EPV;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int PSTR = Cin * (int)sizeof(T) + 16;                  // padded pixel stride (bytes)
    float* wl = (float*)(sm + 3 * 66 * PSTR);                    // [Cout][9][Cin] fp32
    const int tid = threadIdx.x;
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 64;
    for (int i = tid; i < Cout * 9 * Cin; i += 256) wl[i] = ET<T>::ld(w + i);
    const int vpp = Cin / EPV;                                   // 16-byte vectors per pixel
    for (int v = tid; v < 3 * 66 * vpp; v += 256) {
        const int cv = v % vpp, p = (v / vpp) % 66, r = v / (vpp * 66);
        const int sy = y + r - 1, sx = x0 + p - 1;
        u32x4 val = (u32x4){0u, 0u, 0u, 0u};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) val = *(const u32x4*)(x + (((long)b * H + sy) * W + sx) * Cin + cv * EPV);
        *(u32x4*)(sm + (r * 66 + p) * PSTR + cv * 16) = val;
    }
    __syncthreads();
    const int pix = tid >> 2, q = tid & 3;                       // 64 pixels x 4 channel quarters
    const int cq = Cin / 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        const char* xp = sm + ((tap / 3) * 66 + pix + tap % 3) * PSTR + q * cq * (int)sizeof(T);
        for (int c = 0; c < cq; c += EPV) {
            const u32x4 v = *(const u32x4*)(xp + c * sizeof(T));
            float f[EPV]; ET<T>::unpack(v, f);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                if (co < Cout) {
                    const float* wp = wl + (co * 9 + tap) * Cin + q * cq + c;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) acc[co] = fmaf(f[e], wp[e], acc[co]);
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        acc[co] += __shfl_xor(acc[co], 1, 64);
        acc[co] += __shfl_xor(acc[co], 2, 64);
    }
    const int xx = x0 + pix;
    if (q == 0 && xx < W) {
        for (int co = 0; co < Cout; ++co) {
            const float v = acc[co] + bias[co];
            const long o = (((long)b * Cout + co) * H + y) * W + xx;
            if (out_bf16) ET<bf16>::st((bf16*)out + o, v); else ((float*)out)[o] = v;
        }
    }
}
template <typename T>
void launch_conv3x3_small(hipStream_t s, const T* x, const T* w, const float* bias, void* out, int out_bf16,
                          int B, int H, int W, int Cin, int Cout) {
    const size_t lds = (size_t)3 * 66 * (Cin * sizeof(T) + 16) + (size_t)Cout * 9 * Cin * sizeof(float);
    auto kfn = conv3x3_small_kernel<T>;
    // the LDS need depends on (Cin, Cout) of the call, so the attribute is set per launch; a refused size must not be followed by a launch that
    // would fail less legibly: the sticky error is left for the C ABI's hipGetLastError() check (-> PG_ERR_HIP), nothing is enqueued
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return;
    hipLaunchKernelGGL(kfn, dim3((W + 63) / 64, H, B), dim3(256), lds, s, x, w, bias, out, out_bf16, H, W, Cin, Cout);
}
template void launch_conv3x3_small<float>(hipStream_t, const float*, const float*, const float*, void*, int, int, int, int, int, int);
template void launch_conv3x3_small<bf16>(hipStream_t, const bf16*, const bf16*, const float*, void*, int, int, int, int, int, int);

// ------------------------------------------------------------------------------- conv_in (Cin tiny)
// Encoder conv_in 3 -> ch (vq_model.py:60): NCHW input, NHWC T output.  One thread per OUTPUT CHANNEL
// keeps its 27 weights in registers; a block walks a 64-pixel strip of one image row: the 3 x 3 x 66
// input patch sits in LDS (zero padded), every pixel costs 9 broadcast LDS reads (the new window column),
// 27 FMAs and one store that is contiguous across the channel lanes (512 B per pixel).
template <typename T>
__global__ __launch_bounds__(128) void conv3x3_in_kernel(const void* __restrict__ x, int x_bf16, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ out, int H, int W, int Cout) {
    __shared__ float patch[3][3][68];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int i = tid; i < 3 * 3 * 66; i += 128) {
        const int ci = i / (3 * 66), r = (i / 66) % 3, c = i % 66;
        const int sy = y + r - 1, sx = x0 + c - 1;
        float v = 0.f;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const long xi = (((long)b * 3 + ci) * H + sy) * W + sx;
            v = x_bf16 ? ET<bf16>::ld((const bf16*)x + xi) : ((const float*)x)[xi];
        }
        patch[ci][r][c] = v;
    }
    __syncthreads();
    for (int co = tid; co < Cout; co += 128) {
        float wr[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) wr[k] = w[(long)co * 27 + k];              // [co][ci][tap]
        const float bs = bias[co];
        float win[3][3][3];                                                    // [ci][row][col]
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int r = 0; r < 3; ++r) { win[ci][r][1] = patch[ci][r][0]; win[ci][r][2] = patch[ci][r][1]; }
        const int npx = min(64, W - x0);
        for (int p = 0; p < npx; ++p) {
            float acc = bs;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    win[ci][r][0] = win[ci][r][1]; win[ci][r][1] = win[ci][r][2]; win[ci][r][2] = patch[ci][r][p + 2];
                }
            // same accumulation order as the reference loop: ci outer, tap = row*3 + col inner
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc = fmaf(win[ci][r][c], wr[ci * 9 + r * 3 + c], acc);
            ET<T>::st(out + (((long)b * H + y) * W + x0 + p) * Cout + co, acc);
        }
    }
}
template <typename T>
void launch_conv3x3_in(hipStream_t s, const void* x, int x_bf16, const float* w, const float* bias, T* out,
                       int B, int H, int W, int Cin, int Cout) {
    (void)Cin;                                                                 // == 3 (checked by the caller)
    hipLaunchKernelGGL(conv3x3_in_kernel<T>, dim3((W + 63) / 64, H, B), dim3(128), 0, s, x, x_bf16, w, bias, out, H, W, Cout);
}
template void launch_conv3x3_in<float>(hipStream_t, const void*, int, const float*, const float*, float*, int, int, int, int, int);
template void launch_conv3x3_in<bf16>(hipStream_t, const void*, int, const float*, const float*, bf16*, int, int, int, int, int);

// ------------------------------------------------------------------------------- nearest code
// VectorQuantizer.forward (vq_model.py:236-258): z and codebook L2-normalised,
// d = |z|^2 + |e|^2 - 2 z.e, argmin (first minimum).  One block per latent vector.
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                       int64_t* __restrict__ idx, int D, int V) {
    __shared__ float sv[4]; __shared__ int si[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    float zn[8]; float ss = 0.f;
    for (int d = 0; d < D; ++d) { zn[d] = z[(long)r * D + d]; ss = fmaf(zn[d], zn[d], ss); }
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float zz = 0.f;
    for (int d = 0; d < D; ++d) { zn[d] /= nrm; zz = fmaf(zn[d], zn[d], zz); }
    float best = INFINITY; int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        float ee = 0.f, dot = 0.f;
        for (int d = 0; d < D; ++d) { const float e = cb[(long)v * D + d]; ee = fmaf(e, e, ee); dot = fmaf(zn[d], e, dot); }
        const float dist = fmaf(-2.f, dot, zz + ee);
        if (dist < best) { best = dist; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        float v = sv[0]; int i = si[0];
        for (int k = 1; k < 4; ++k) if (sv[k] < v || (sv[k] == v && si[k] < i)) { v = sv[k]; i = si[k]; }
        idx[r] = i;
    }
}
// Round 4: ZB latent vectors per block -- every codebook entry a thread loads (and its |e|^2) serves ZB distance computations instead of one
// (the one-vector kernel above re-reads the 512 KiB codebook from L2 once per latent vector: 19 GB for 64 images, 7.8 ms).  The arithmetic
// of a (vector, code) pair is the SAME expression sequence as above (normalisation, ee, dot, zz + ee - 2 dot, first minimum), spelled as
// explicit fmaf chains in BOTH kernels so the equality holds by construction rather than by hipcc making the same -ffp-contract choices in
// two differently shaped loops; asserted against the reference fixtures and against the one-vector kernel (tests/test_gpu_ops.py).
template <int ZB>
__global__ __launch_bounds__(256) void vq_argmin_multi_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                             int64_t* __restrict__ idx, int n, int V) {
    constexpr int D = 8;
    __shared__ float sv[ZB][4]; __shared__ int si[ZB][4];
    const int r0 = blockIdx.x * ZB, tid = threadIdx.x;
    float zn[ZB][D], zz[ZB], best[ZB]; int bi[ZB];
#pragma unroll
    for (int j = 0; j < ZB; ++j) {
        const int r = r0 + j < n ? r0 + j : n - 1;
        float ss = 0.f;
        for (int d = 0; d < D; ++d) { zn[j][d] = z[(long)r * D + d]; ss = fmaf(zn[j][d], zn[j][d], ss); }
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        zz[j] = 0.f;
        for (int d = 0; d < D; ++d) { zn[j][d] /= nrm; zz[j] = fmaf(zn[j][d], zn[j][d], zz[j]); }
        best[j] = INFINITY; bi[j] = 0x7fffffff;
    }
    for (int v = tid; v < V; v += 256) {
        float e[D];
        const f32x4 e0 = *(const f32x4*)(cb + (long)v * D), e1 = *(const f32x4*)(cb + (long)v * D + 4);
        e[0] = e0[0]; e[1] = e0[1]; e[2] = e0[2]; e[3] = e0[3]; e[4] = e1[0]; e[5] = e1[1]; e[6] = e1[2]; e[7] = e1[3];
        float ee = 0.f;
        for (int d = 0; d < D; ++d) ee = fmaf(e[d], e[d], ee);
#pragma unroll
        for (int j = 0; j < ZB; ++j) {
            float dot = 0.f;
            for (int d = 0; d < D; ++d) dot = fmaf(zn[j][d], e[d], dot);
            const float dist = fmaf(-2.f, dot, zz[j] + ee);
            if (dist < best[j]) { best[j] = dist; bi[j] = v; }
        }
    }
#pragma unroll
    for (int j = 0; j < ZB; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best[j], o, 64); const int oi = __shfl_xor(bi[j], o, 64);
            if (ov < best[j] || (ov == best[j] && oi < bi[j])) { best[j] = ov; bi[j] = oi; }
        }
        if ((tid & 63) == 0) { sv[j][tid >> 6] = best[j]; si[j][tid >> 6] = bi[j]; }
    }
    __syncthreads();
    if (tid < ZB && r0 + tid < n) {
        float v = sv[tid][0]; int i = si[tid][0];
        for (int k = 1; k < 4; ++k) if (sv[tid][k] < v || (sv[tid][k] == v && si[tid][k] < i)) { v = sv[tid][k]; i = si[tid][k]; }
        idx[r0 + tid] = i;
    }
}
void launch_vq_argmin(hipStream_t s, const float* z, const float* codebook, int64_t* idx, int n, int D, int V) {
    if (n <= 0) return;
    if (D == 8 && pg_tune->vq_argmin_multi && n >= 64) {
        constexpr int ZB = 8;
        hipLaunchKernelGGL(vq_argmin_multi_kernel<ZB>, dim3((n + ZB - 1) / ZB), dim3(256), 0, s, z, codebook, idx, n, V);
        return;
    }
    hipLaunchKernelGGL(vq_argmin_kernel, dim3(n), dim3(256), 0, s, z, codebook, idx, D, V);
}

// ------------------------------------------------------------------------------- SigLIP helpers
// LayerNorm(eps) over the last dim: fp32 rows in, T out (siglip_vit.py Block.norm1/norm2, final norm).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ y, int C, float eps) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (long)m * C;
    float s = 0.f;
    for (int i = tid; i < C; i += 256) s += xr[i];
    const float mean = block_sum<4>(s, red) / (float)C;
    float q = 0.f;
    for (int i = tid; i < C; i += 256) { const float d = xr[i] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(block_sum<4>(q, red) / (float)C + eps);
    for (int i = tid; i < C; i += 256) ET<T>::st(y + (long)m * C + i, (xr[i] - mean) * rstd * gamma[i] + beta[i]);
}
// C == NV * 256: one WAVE per row (4 rows per block), the row held in registers (NV f32x4 per lane), two-pass statistics with wave
// reductions only (no LDS, no block barrier), 16-byte loads and 8- / 16-byte stores.  The generic kernel above re-reads the row three
// times with 4-byte accesses behind four block barriers (82 us per call on the 36 864 x 1024 SigLIP activations).
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, int M, float eps) {
    constexpr int C = NV * 256;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (m >= M) return;
    const float* xr = x + (long)m * C;
    f32x4 v[NV], gv[NV], bv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { v[j] = *(const f32x4*)(xr + j * 256 + l * 4); gv[j] = *(const f32x4*)(gamma + j * 256 + l * 4); bv[j] = *(const f32x4*)(beta + j * 256 + l * 4); }
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s1 += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(s1) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gv[j][e] + bv[j][e];
        T* dst = y + (long)m * C + j * 256 + l * 4;
        if constexpr (sizeof(T) == 2) { u32x2 pk; pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]); *(u32x2*)dst = pk; }
        else *(f32x4*)dst = (f32x4){o[0], o[1], o[2], o[3]};
    }
}
template <typename T>
void launch_layernorm(hipStream_t s, const float* x, const float* gamma, const float* beta, T* y, int M, int C, float eps) {
    if (M <= 0) return;
    if (C == 1024 && pg_tune->ln_wave) { hipLaunchKernelGGL((layernorm_wave_kernel<T, 4>), dim3((M + 3) / 4), dim3(256), 0, s, x, gamma, beta, y, M, eps); return; }
    hipLaunchKernelGGL(layernorm_kernel<T>, dim3(M), dim3(256), 0, s, x, gamma, beta, y, C, eps);
}
template void launch_layernorm<float>(hipStream_t, const float*, const float*, const float*, float*, int, int, float);
template void launch_layernorm<bf16>(hipStream_t, const float*, const float*, const float*, bf16*, int, int, float);

// PatchEmbed's Conv2d(3, C, ps, stride ps) as a GEMM: patches [B*P, 3*ps*ps], k = (c, py, px).
template <typename T>
__global__ void patchify_kernel(const void* __restrict__ img, int img_bf16, T* __restrict__ out, int S, int ps) {
    const int g = S / ps, K = 3 * ps * ps;
    const int row = blockIdx.x;                       // b*P + p
    const int b = row / (g * g), p = row % (g * g), py0 = (p / g) * ps, px0 = (p % g) * ps;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int c = k / (ps * ps), r = k % (ps * ps), py = r / ps, px = r % ps;
        const long si = (((long)b * 3 + c) * S + py0 + py) * S + px0 + px;
        const float v = img_bf16 ? ET<bf16>::ld((const bf16*)img + si) : ((const float*)img)[si];
        ET<T>::st(out + (long)row * K + k, v);
    }
}
template <typename T>
void launch_patchify(hipStream_t s, const void* img, int img_bf16, T* out, int B, int S, int ps) {
    const int g = S / ps;
    hipLaunchKernelGGL(patchify_kernel<T>, dim3(B * g * g), dim3(256), 0, s, img, img_bf16, out, S, ps);
}
template void launch_patchify<float>(hipStream_t, const void*, int, float*, int, int, int);
template void launch_patchify<bf16>(hipStream_t, const void*, int, bf16*, int, int, int);

__global__ void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, int P, int C) {
    const int row = blockIdx.x, p = row % P;
    for (int i = threadIdx.x; i < C; i += blockDim.x) x[(long)row * C + i] += pos[(long)p * C + i];
}
void launch_add_pos(hipStream_t s, float* x, const float* pos, int B, int P, int C) {
    hipLaunchKernelGGL(add_pos_kernel, dim3(B * P), dim3(256), 0, s, x, pos, P, C);
}
