// A-operand loaders, the runtime epilogue and the LDS-DMA helper shared by the MFMA GEMM kernels
// (gemm.hip: 128x128 tile; gemm256.hip: 256x256 eight-phase tile).
#pragma once
#include "kernels.h"

// ------------------------------------------------------------------------------- loaders
template <typename T>
struct PlainLoader {
    const T* A; long lda; int M;
    const T* rowp[8];
    __device__ __forceinline__ void init(int i, int m) { rowp[i] = A + (long)(m < M ? m : M - 1) * lda; }
    __device__ __forceinline__ void set_ktile(int) {}
    __device__ __forceinline__ const T* ptr(int i, int k) const { return rowp[i] + k; }
    __device__ __forceinline__ float elem(int m, int k) const { return m < M ? ET<T>::ld(A + (long)m * lda + k) : 0.f; }
};

// 3x3 conv, pad 1 (stride 1, optional nearest-2x upsample of the input) or the encoder's
// stride-2 / pad (0,1,0,1) form.  Input NHWC [B,Hi,Wi,Cin]; output pixel m = (b, y, x)
// over [B,Ho,Wo]; K index = tap*Cin + ci.
template <typename T>
struct ConvLoader {
    const T* X; const T* zeros; int Hi, Wi, Cin, up, stride2, Ho, Wo, M;
    const T* img[4]; int yy[4], xx[4];
    const T* cur[4];               // source pixel of the CURRENT tap for each staging slot (nullptr = halo -> zeros)
    int ci0, cur_tap;
    __device__ __forceinline__ void decode(int m, int& b, int& y, int& x) const {
        x = m % Wo; int t = m / Wo; y = t % Ho; b = t / Ho;
    }
    __device__ __forceinline__ void init(int i, int m) {
        if (m >= M) m = M - 1;
        int b, y, x; decode(m, b, y, x);
        img[i] = X + (long)b * Hi * Wi * Cin; yy[i] = y; xx[i] = x;
        cur_tap = -1;
    }
    __device__ __forceinline__ bool src_yx(int y, int x, int ddy, int ddx, int& sy, int& sx) const {
        if (stride2) { sy = 2 * y + ddy; sx = 2 * x + ddx; return sy < Hi && sx < Wi; }
        const int oy = y + ddy - 1, ox = x + ddx - 1;
        sy = oy >> up; sx = ox >> up;
        return oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
    }
    // K tiles arrive in increasing k; the halo test and the 64-bit address are recomputed only when
    // the tap changes (every Cin/64 tiles).  Cin is a power of two (64..512): shift / mask.
    __device__ __forceinline__ void set_ktile(int k0) {
        const int lc = 31 - __builtin_clz(Cin);
        const int tap = k0 >> lc; ci0 = k0 & (Cin - 1);
        if (tap != cur_tap) {
            cur_tap = tap;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int sy, sx;
                cur[i] = src_yx(yy[i], xx[i], dy, dx, sy, sx) ? img[i] + ((long)sy * Wi + sx) * Cin : nullptr;
            }
        }
    }
    // k = k-tile base + chunk offset (chunk of 8 elements inside one tap because Cin % 64 == 0)
    __device__ __forceinline__ const T* ptr(int i, int k) const {
        const int koff = k & 63;
        return cur[i] ? cur[i] + ci0 + koff : zeros + koff;
    }
    __device__ __forceinline__ float elem(int m, int k) const {
        if (m >= M) return 0.f;
        int b, y, x; decode(m, b, y, x);
        const int tap = k / Cin, ci = k - tap * Cin, ddy = tap / 3, ddx = tap - ddy * 3;
        int sy, sx;
        if (!src_yx(y, x, ddy, ddx, sy, sx)) return 0.f;
        return ET<T>::ld(X + (((long)b * Hi + sy) * Wi + sx) * Cin + ci);
    }
};

// Same addressing as ConvLoader with NS staging slots in 3 registers each (32-bit element offsets: image
// base, packed output (y, x), source pixel of the current tap or -1 for the zero halo) -- the big-tile
// kernels spend their registers on accumulators.  bf16 only; input must hold < 2^31 elements.
template <int NS>
struct ConvLoaderS {
    const bf16* X; const bf16* zeros; int Hi, Wi, Cin, up, stride2, Ho, Wo, M;
    int ibase[NS], yx[NS], cur[NS];
    int ci0, cur_tap;
    void setup(const GemmA& a, int M_) {
        X = (const bf16*)a.ptr; zeros = (const bf16*)a.zeros; Hi = a.Hi; Wi = a.Wi; Cin = a.Cin; up = a.up; stride2 = (a.kind == 2);
        Ho = stride2 ? a.Hi / 2 : (a.Hi << a.up); Wo = stride2 ? a.Wi / 2 : (a.Wi << a.up); M = M_;
    }
    __device__ __forceinline__ void A_offset(long off) { X += off; }
    __device__ __forceinline__ void init(int i, int m) {
        if (m >= M) m = M - 1;
        const int x = m % Wo, t = m / Wo, y = t % Ho, b = t / Ho;
        ibase[i] = b * Hi * Wi * Cin; yx[i] = (y << 16) | x;
        cur_tap = -1;
    }
    __device__ __forceinline__ void set_ktile(int k0) {
        const int lc = 31 - __builtin_clz(Cin);
        const int tap = k0 >> lc; ci0 = k0 & (Cin - 1);
        if (tap != cur_tap) {
            cur_tap = tap;
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int y = yx[i] >> 16, x = yx[i] & 0xffff;
                int sy, sx; bool ok;
                if (stride2) { sy = 2 * y + dy; sx = 2 * x + dx; ok = sy < Hi && sx < Wi; }
                else { const int oy = y + dy - 1, ox = x + dx - 1; sy = oy >> up; sx = ox >> up; ok = oy >= 0 && oy < Ho && ox >= 0 && ox < Wo; }
                cur[i] = ok ? ibase[i] + (sy * Wi + sx) * Cin : -1;
            }
        }
    }
    __device__ __forceinline__ const bf16* ptr(int i, int k) const {
        const int koff = k & 63;
        return cur[i] >= 0 ? X + (cur[i] + ci0 + koff) : zeros + koff;
    }
};

// ------------------------------------------------------------------------------- epilogue
template <typename T>
struct Epi {
    GemmEpi e; int M, N;
    __device__ __forceinline__ void operator()(long coff, long roff, int row, int col, float v) const {
        if (row >= M || col >= N) return;
        v *= e.scale;
        if (e.bias_n) v += e.bias_n[col];
        if (e.bias_m) v += e.bias_m[row];
        if (e.residual) {
            const long ldr = e.ldr ? e.ldr : e.ldc;
            const long ro = roff + (long)row * ldr + col;
            v += e.res_f32 ? ((const float*)e.residual)[ro] : ET<T>::ld((const T*)e.residual + ro);
        }
        if (e.act == 1) v = gelu_erf(v);
        const long o = coff + (long)row * e.ldc + col;
        if (e.out_f32) ((float*)e.out)[o] = v; else ET<T>::st((T*)e.out + o, v);
    }
    // four consecutive columns of one row (col % 4 == 0); 16-byte (fp32) / 8-byte (bf16) accesses
    // when the layout allows it (vec_ok()), scalar otherwise.  Same arithmetic order as operator().
    __device__ __forceinline__ bool vec_ok(long coff, long roff) const {
        const long ldr = e.ldr ? e.ldr : e.ldc;
        return ((e.ldc | coff | ldr | roff) & 3) == 0 && (N & 3) == 0;
    }
    // vector path of store4 split in two for kernels that also want the stored values (GroupNorm partials):
    // in-range rows / columns and vec_ok() layouts only
    __device__ __forceinline__ f32x4 final4(long roff, int row, int col, f32x4 v) const {
        v *= e.scale;
        if (e.bias_n) v += *(const f32x4*)(e.bias_n + col);
        if (e.bias_m) v += e.bias_m[row];
        if (e.residual) {
            const long ldr = e.ldr ? e.ldr : e.ldc;
            const long ro = roff + (long)row * ldr + col;
            if (e.res_f32 || sizeof(T) == 4) v += *(const f32x4*)((const float*)e.residual + ro);
            else {
                const uint2 r2 = *(const uint2*)((const bf16*)e.residual + ro);
                v += (f32x4){bf16_lo(r2.x), bf16_hi(r2.x), bf16_lo(r2.y), bf16_hi(r2.y)};
            }
        }
        if (e.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        return v;
    }
    __device__ __forceinline__ void put4(long coff, int row, int col, f32x4 v) const {
        const long o = coff + (long)row * e.ldc + col;
        if (e.out_f32 || sizeof(T) == 4) *(f32x4*)((float*)e.out + o) = v;
        else { uint2 q; q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); *(uint2*)((bf16*)e.out + o) = q; }
    }
    __device__ __forceinline__ void store4(long coff, long roff, int row, int col, f32x4 v, bool vec) const {
        if (row >= M || col >= N) return;
        if (!vec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) (*this)(coff, roff, row, col + j, v[j]);
            return;
        }
        v *= e.scale;
        if (e.bias_n) v += *(const f32x4*)(e.bias_n + col);
        if (e.bias_m) v += e.bias_m[row];
        if (e.residual) {
            const long ldr = e.ldr ? e.ldr : e.ldc;
            const long ro = roff + (long)row * ldr + col;
            if (e.res_f32 || sizeof(T) == 4) v += *(const f32x4*)((const float*)e.residual + ro);
            else {
                const uint2 r2 = *(const uint2*)((const bf16*)e.residual + ro);
                v += (f32x4){bf16_lo(r2.x), bf16_hi(r2.x), bf16_lo(r2.y), bf16_hi(r2.y)};
            }
        }
        if (e.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        const long o = coff + (long)row * e.ldc + col;
        if (e.out_f32 || sizeof(T) == 4) *(f32x4*)((float*)e.out + o) = v;
        else { uint2 q; q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); *(uint2*)((bf16*)e.out + o) = q; }
    }
    // NF scalar elements at once, same arithmetic and order as operator(): loads batched like store4_batch (the 128x128 kernel's C layout
    // gives a lane single elements of 4 consecutive rows)
    template <int NF>
    __device__ __forceinline__ void store1_batch(long coff, long roff, const int (&row)[NF], const int (&col)[NF], const float (&acc)[NF]) const {
        int rc[NF], cc[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) { rc[i] = row[i] < M ? row[i] : M - 1; cc[i] = col[i] < N ? col[i] : N - 1; }
        float b[NF], bm[NF], r[NF];
        if (e.bias_n) {
#pragma unroll
            for (int i = 0; i < NF; ++i) b[i] = e.bias_n[cc[i]];
        }
        if (e.bias_m) {
#pragma unroll
            for (int i = 0; i < NF; ++i) bm[i] = e.bias_m[rc[i]];
        }
        if (e.residual) {
            const long ldr = e.ldr ? e.ldr : e.ldc;
            if (e.res_f32) {
#pragma unroll
                for (int i = 0; i < NF; ++i) r[i] = ((const float*)e.residual)[roff + (long)rc[i] * ldr + cc[i]];
            } else {
#pragma unroll
                for (int i = 0; i < NF; ++i) r[i] = ET<T>::ld((const T*)e.residual + roff + (long)rc[i] * ldr + cc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            float v = acc[i] * e.scale;
            if (e.bias_n) v += b[i];
            if (e.bias_m) v += bm[i];
            if (e.residual) v += r[i];
            if (e.act == 1) v = gelu_erf(v);
            if (row[i] < M && col[i] < N) {
                const long o = coff + (long)row[i] * e.ldc + col[i];
                if (e.out_f32) ((float*)e.out)[o] = v; else ET<T>::st((T*)e.out + o, v);
            }
        }
    }
    // NF fragments at once, same arithmetic and order as store4(): every bias / residual load of the batch is issued before the first
    // use (one wave-uniform branch per operand kind, indices clamped into the matrix, stores predicated).  Calling store4() per
    // fragment compiles to load -> s_waitcnt vmcnt(0) -> load -> s_waitcnt vmcnt(0) -> store for EVERY fragment (the `if (e.bias_n)` /
    // `if (e.residual)` guards become exec-masked blocks with a drain at each join): two dependent memory round trips per fragment,
    // 16-32 fragments per tile.  vals_out (optional): the stored values (GroupNorm partials of conv_halo).
    template <int NF>
    __device__ __forceinline__ void store4_batch(long coff, long roff, const int (&row)[NF], const int (&col)[NF], const f32x4 (&acc)[NF], bool vec,
                                                 f32x4* vals_out = nullptr) const {
        if (!vec) {
#pragma unroll
            for (int i = 0; i < NF; ++i) { store4(coff, roff, row[i], col[i], acc[i], false); if (vals_out) vals_out[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            return;
        }
        int rc[NF], cc[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) { rc[i] = row[i] < M ? row[i] : M - 1; cc[i] = col[i] + 4 <= N ? col[i] : N - 4; }
        f32x4 b[NF], r[NF];
        float bm[NF];
        if (e.bias_n) {
#pragma unroll
            for (int i = 0; i < NF; ++i) b[i] = *(const f32x4*)(e.bias_n + cc[i]);
        }
        if (e.bias_m) {
#pragma unroll
            for (int i = 0; i < NF; ++i) bm[i] = e.bias_m[rc[i]];
        }
        if (e.residual) {
            const long ldr = e.ldr ? e.ldr : e.ldc;
            if (e.res_f32 || sizeof(T) == 4) {
#pragma unroll
                for (int i = 0; i < NF; ++i) r[i] = *(const f32x4*)((const float*)e.residual + roff + (long)rc[i] * ldr + cc[i]);
            } else {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const uint2 r2 = *(const uint2*)((const bf16*)e.residual + roff + (long)rc[i] * ldr + cc[i]);
                    r[i] = (f32x4){bf16_lo(r2.x), bf16_hi(r2.x), bf16_lo(r2.y), bf16_hi(r2.y)};
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            f32x4 v = acc[i];
            v *= e.scale;
            if (e.bias_n) v += b[i];
            if (e.bias_m) v += bm[i];
            if (e.residual) v += r[i];
            if (e.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            if (vals_out) vals_out[i] = v;
            if (row[i] < M && col[i] < N) {
                const long o = coff + (long)row[i] * e.ldc + col[i];
                if (e.out_f32 || sizeof(T) == 4) *(f32x4*)((float*)e.out + o) = v;
                else { uint2 q; q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); *(uint2*)((bf16*)e.out + o) = q; }
            }
        }
    }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (hardware); source address is per lane.
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// loaders need a batch offset hook
template <typename T> struct PlainLoaderB : PlainLoader<T> {
    __device__ __forceinline__ void A_offset(long off) { this->A += off; }
};
template <typename T> struct ConvLoaderB : ConvLoader<T> {
    __device__ __forceinline__ void A_offset(long off) { this->X += off; }
};

