// Device-side helpers shared by all kernels.  gfx950 (MI355X / CDNA4) only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define WAVE 64

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b16) { return __uint_as_float(b16 << 16); }
__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
// Software round-to-nearest-even (rounds 1-3: ~6 VALU operations per element).  Kept as the reference of the exhaustive check below.
__device__ __forceinline__ uint32_t f32_to_bf16_bits_sw(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN -> quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// Round 4: gfx950 converts in hardware (v_cvt_pk_bf16_f32, one instruction per PAIR, round-to-nearest-even).  Identical to the software form
// for every non-NaN fp32 bit pattern (pg_bench_bf16_cvt_check walks all 2^32, tests/test_gpu_ops.py); NaNs stay NaNs (payload may differ).
typedef __attribute__((ext_vector_type(2))) __bf16 pg_bf16x2_t;
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    const __bf16 b = (__bf16)f;
    return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    pg_bf16x2_t r; r[0] = (__bf16)lo; r[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, r);
}

// Element-type traits: T = float (PG_F32 mode) or bf16 (PG_BF16 mode).
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int EPV = 4;                  // elements per 16-byte vector
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    // unpack a 16-byte vector into EPV floats
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v; v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]);
        v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]); return v;
    }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct ET<bf16> {
    static constexpr int EPV = 8;
    __device__ static __forceinline__ float ld(const bf16* p) {
        return bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(p));
    }
    __device__ static __forceinline__ void st(bf16* p, float v) {
        *reinterpret_cast<uint16_t*>(p) = (uint16_t)f32_to_bf16_bits(v);
    }
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
        f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
        f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v; v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
        v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]); return v;
    }
    __device__ static __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

// ---- wave / block reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over a block of NW waves; every thread gets the result.  ``red`` >= NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float swish_precise(float x) { return x / (1.f + expf(-x)); }

// XCD-aware block remap (MI355X: 8 XCDs, block b lands on XCD b%8): give each XCD a
// contiguous chunk of the logical tile space so neighbouring tiles share one L2.
// Bijective for any nb (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Counter-based RNG (splitmix64 finaliser) -> uniform strictly inside (0,1).
// 23 random bits: x + 0.5 with x < 2^23 is exact in fp32 (24-bit significand), so
// u in [2^-24, 1 - 2^-24].  (24 bits would round 16777215.5 up to 2^24 -> u == 1.0 ->
// Gumbel noise +inf.)
__device__ __forceinline__ uint64_t rng_bits(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xBF58476D1CE4E5B9ull * (b + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float uniform_from_bits(uint64_t z) {
    return ((float)(z >> 41) + 0.5f) * (1.0f / 8388608.0f);
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t a, uint64_t b) {
    return uniform_from_bits(rng_bits(seed, a, b));
}

// One row of (split-K reduce + residual add + RMSNorm) by a 256-thread block (rmsnorm_kernel and the
// norm blocks of the fused norm+GEMM launch share it).  ``red``: >= 4 floats of LDS.
template <typename T, int NV, bool SC1 = false, int SB = 4, int NT = 256>      // SB: slabs loaded in the up-front batch (4, or 8 for S > 4; 0 = the caller guarantees S == 0: no slab loads at all -- prefill, round 5); NT: threads per row
__device__ __forceinline__ void rmsnorm_row(int m, float* __restrict__ x, const float* __restrict__ partial,
                                            int S, long slab, const T* __restrict__ w,
                                            T* __restrict__ xn, int H, float eps, float* red) {
    const int tid = threadIdx.x;
    float* xr = x + (long)m * H;
    f32x4 v[NV];
    float ss = 0.f;
    // norm weights first: independent of the reduction, so the row costs ONE memory round trip
    u32x2 wv2[NV]; f32x4 wv4[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid * 4 + j * (NT * 4);
        if (xn && i < H) { if constexpr (sizeof(T) == 2) wv2[j] = *(const u32x2*)(w + i); else wv4[j] = *(const f32x4*)(w + i); }
    }
    // ALL of the row's loads go out before the first add: x and the first four slabs of every one of the thread's NV vectors in one
    // straight-line batch (a plain ``for s`` loop is not unrolled by hipcc for runtime S and degenerates into S dependent round trips;
    // and with the batch inside the per-vector loop the second vector's loads waited for the first vector's adds: two round trips
    // per row at NV = 2, 5.2 us per launch)
    f32x4 t[NV][SB > 0 ? SB : 1];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid * 4 + j * (NT * 4);
        const int ic = i < H ? i : 0;                                  // clamped: loads unconditional, results discarded
        v[j] = *(const f32x4*)(xr + ic);
        if constexpr (SB == 0) continue;                               // S == 0 by contract: the branch-free form below would re-read the row SB times for nothing
        // branch-free: S == 0 (no slabs, ``partial`` may be null) reads the residual row again and discards it; a conditional load
        // makes hipcc drain vmcnt at every join
        const float* pp = (S > 0 ? partial + (long)m * H : xr) + ic;
        const long sl = S > 0 ? slab : 0;
        const int smax = S > 0 ? S - 1 : 0;
#pragma unroll
        for (int u = 0; u < SB; ++u) t[j][u] = *(const f32x4*)(pp + (long)(u < smax ? u : smax) * sl);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid * 4 + j * (NT * 4);
        if (i < H) {
#pragma unroll
            for (int u = 0; u < SB; ++u) if (u < S) v[j] += t[j][u];
            const float* pp = partial + (long)m * H + i;
            for (int s0 = SB; s0 < S; s0 += 8) {                       // S > SB: further batches of 8 independent requests
                f32x4 t8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sidx = s0 + u < S ? s0 + u : S - 1;
                    t8[u] = *(const f32x4*)(pp + (long)sidx * slab);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (s0 + u < S) v[j] += t8[u];
            }
            ss += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        } else v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    ss = block_sum<NT / 64>(ss, red);
    // the updated residual row is stored AFTER the reduction: in front of it, the block barrier's release waited for the stores'
    // acknowledgements (`s_waitcnt vmcnt(0)` before `s_barrier`), ~1 us of a 5 us kernel
    if (S > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid * 4 + j * (NT * 4);
            if (i < H) *(f32x4*)(xr + i) = v[j];
        }
    }
    if (!xn) return;
    const float rstd = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid * 4 + j * (NT * 4);
        if (i < H) {
            float o[4];
            if constexpr (sizeof(T) == 2) {
                const u32x2 wv = wv2[j];
                o[0] = bf16_lo(wv.x) * (v[j].x * rstd); o[1] = bf16_hi(wv.x) * (v[j].y * rstd);
                o[2] = bf16_lo(wv.y) * (v[j].z * rstd); o[3] = bf16_hi(wv.y) * (v[j].w * rstd);
                u32x2 ov; ov.x = pack_bf16x2(o[0], o[1]); ov.y = pack_bf16x2(o[2], o[3]);
                // SC1: write-through (device-scope) store: the row is consumed by other XCDs inside this launch
                if constexpr (SC1) __hip_atomic_store((uint64_t*)(xn + (long)m * H + i), ((uint64_t)ov.y << 32) | ov.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *(u32x2*)(xn + (long)m * H + i) = ov;
            } else {
                const f32x4 wv = wv4[j];
                f32x4 ov = {wv.x * (v[j].x * rstd), wv.y * (v[j].y * rstd), wv.z * (v[j].z * rstd), wv.w * (v[j].w * rstd)};
                *(f32x4*)(xn + (long)m * H + i) = ov;
            }
        }
    }
}

// rotate_half RoPE of one (x[j], x[j+64]) pair with EXPLICIT contraction, shared by rope_kv_kernel and the 256x256 GEMM's RoPE epilogue so
// the fused and the unfused prefill write identical bits (left to the compiler, `a*c - b*s` may contract either product)
__device__ __forceinline__ float rope_lo(float x0, float x1, float c, float s) { return __builtin_fmaf(x0, c, -(x1 * s)); }   // out[j]    = x0 cos - x1 sin
__device__ __forceinline__ float rope_hi(float x0, float x1, float c, float s) { return __builtin_fmaf(x1, c, x0 * s); }      // out[j+64] = x1 cos + x0 sin

// Compute units of the device the calling thread is on (cached per ordinal): the persistent GEMM sizes its grid and its tile-height cost model
// with it instead of a literal 256.  Falls back to MI355X's 256 if the query fails.
static inline int pg_cu_count() {
    static int cached_[64] = {};
    int d_ = 0;
    (void)hipGetDevice(&d_);
    int& c = cached_[d_ & 63];
    if (c <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d_) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        c = v;
    }
    return c;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): set once per device the calling thread is on, result
// checked (a process-wide "done" flag left a second GPU's handle launching without it).  One call site = one flag word, one bit
// per device ordinal; evaluates to false when the attribute call fails (the caller falls back or lets the launch report the error).
#define PG_DYN_LDS(kfn, bytes)                                                                                                  \
    ([&]() -> bool {                                                                                                            \
        static unsigned long long done_ = 0;                                                                                    \
        int d_ = 0;                                                                                                             \
        (void)hipGetDevice(&d_);                                                                                                \
        const unsigned long long bit_ = 1ull << (d_ & 63);                                                                      \
        if (done_ & bit_) return true;                                                                                          \
        if (hipFuncSetAttribute((const void*)(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) {  \
            (void)hipGetLastError();                                                                                            \
            return false;                                                                                                       \
        }                                                                                                                       \
        done_ |= bit_;                                                                                                          \
        return true;                                                                                                            \
    }())
