// Persistent decode chain for <= 16 rows (round 4): o_proj -> (+residual, RMSNorm) -> gate|up + SwiGLU -> down_proj ->
// (+residual, RMSNorm) -> next layer's qkv in ONE launch of 256 workgroups (one per CU), with a RUN-AHEAD weight stream: three compute
// waves per CU load their share of the layer's tiled weights into a register ring that keeps running across the three grid-wide seams,
// a fourth "service" wave gathers the activations (LDS-DMA), publishes the results (write-through stores) and runs the grid barrier.
// Reference arithmetic: transformers LlamaDecoderLayer as driven by plangen_base.py:571-577.
//
// THIS FILE: the measured SKELETON of that kernel (same launch geometry, same per-CU byte schedule, same barriers, gathers and publishes,
// dummy arithmetic) -- it answers "what is the floor of this structure on MI355X" before the arithmetic is written
// (profiles/r04_c_persistent_chain_skeleton.md).  Work split at 16 rows (tiled weights, one n-tile = 16 columns, one chunk = 128 k = 4 KiB):
//   even CU b = 2t : o_proj n-tile t (16 chunks) | gate|up units b, b+256, b+512 (16 chunks each) | qkv half-units 3b..3b+2 (8 chunks each)
//   odd  CU b = 2t+1:                             | gate|up units (as above)                      | down n-tile t (44 chunks, two K halves) | qkv
// i.e. 352 / 464 KiB of weights per CU and layer; every unit's K range is split round-robin over the three compute waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/plangen_hip.h"
#include "kernels.h"
#include "gemm_common.h"

namespace {

struct ChainBar {                      // every word on its own 128-byte line; monotonic counters, never reset
    unsigned cnt[8][32];               // arrivals per group (group = blockIdx & 7 = the XCD a block lands on in practice; correctness does not depend on it)
    unsigned top[32];                  // groups complete
    unsigned gen[32];                  // released epoch
    unsigned err[32];                  // a spin gave up
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid barrier, called by ONE lane of the service wave after its published stores have drained (s_waitcnt vmcnt(0)).  epoch counts
// barriers since the state was zeroed (1, 2, 3, ...).  Bounded spin (~20 ms): on give-up err is set and the kernel carries on.
__device__ __forceinline__ void chain_grid_barrier(ChainBar* b, unsigned epoch, unsigned per_group) {
    const int g = blockIdx.x & 7;
    const unsigned old = __hip_atomic_fetch_add(&b->cnt[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * per_group) {
        const unsigned o2 = __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o2 + 1 == epoch * 8u) __hip_atomic_store(&b->gen[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t0 = wall_clock64();
    while (ld_relaxed(&b->gen[0]) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 2000000ull) { __hip_atomic_store(&b->err[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

struct ChainArgs {
    const bf16 *wo, *wgu, *wd, *wqkv;          // tiled decode copies: o(l), gate|up(l), down(l), qkv(l+1)
    const bf16 *a_o, *a_x, *a_h;               // gather sources: attention output [16][2048], normalised-input proxy [16][2048], h [16][5632]
    char* pub;                                 // publish target (>= 256 * 4 KiB)
    ChainBar* bar; unsigned epoch0;            // barriers of this launch use epochs epoch0 + 1 .. epoch0 + 3
    int mode;                                  // 1 grid barriers, 2 weight stream, 4 gathers + publishes
};

constexpr int cnt3(int n, int w) { return (n - w + 2) / 3; }                     // chunks of an n-chunk unit owned by compute wave w (round-robin)
// segments of a role, in program order: chunks per segment and the phase it belongs to (0 o, 1 gate|up, 2 down half 0, 3 down half 1, 4 qkv)
template <int ROLE> struct RoleDef;
template <> struct RoleDef<0> { static constexpr int NSEG = 7; static constexpr int n[7] = {16, 16, 16, 16, 8, 8, 8}; static constexpr int ph[7] = {0, 1, 1, 1, 4, 4, 4}; };
template <> struct RoleDef<1> { static constexpr int NSEG = 8; static constexpr int n[8] = {16, 16, 16, 22, 22, 8, 8, 8}; static constexpr int ph[8] = {1, 1, 1, 2, 3, 4, 4, 4}; };
template <int ROLE, int W> constexpr int total_items() { int t = 0; for (int s = 0; s < RoleDef<ROLE>::NSEG; ++s) t += cnt3(RoleDef<ROLE>::n[s], W); return t; }
struct ItemRef { int seg, j; };
template <int ROLE, int W> constexpr ItemRef item_ref(int i) {
    for (int s = 0; s < RoleDef<ROLE>::NSEG; ++s) { const int c = cnt3(RoleDef<ROLE>::n[s], W); if (i < c) return ItemRef{s, i}; i -= c; }
    return ItemRef{-1, 0};
}
template <int ROLE, int W> constexpr bool first_of_phase(int i) {
    if (i == 0) return true;
    return RoleDef<ROLE>::ph[item_ref<ROLE, W>(i).seg] != RoleDef<ROLE>::ph[item_ref<ROLE, W>(i - 1).seg];
}
template <int ROLE, int W> constexpr bool last_of_phase(int i) {
    if (i + 1 == total_items<ROLE, W>()) return true;
    return RoleDef<ROLE>::ph[item_ref<ROLE, W>(i).seg] != RoleDef<ROLE>::ph[item_ref<ROLE, W>(i + 1).seg];
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int N_> __device__ __forceinline__ void wait_ring(u32x4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N_ > 63 ? 63 : N_) : "memory");
}

__device__ __forceinline__ void ring_load(u32x4& dst, const bf16* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory"); }
constexpr int RING = 15;               // chunks per compute wave held / in flight: 60 loads (vmcnt is 6 bits), 240 VGPRs

// One compute wave: its items in program order, loads RING items ahead of the consumer, block barriers at the phase edges.
// (MODE is a template parameter: a RUNTIME branch around the asm loads makes hipcc copy ring registers at the joins -- the copy reads a
// register whose load is still in flight and the original register gets re-used, here for an address: memory aperture violation.)
template <int ROLE, int W, int MODE>
__device__ __forceinline__ void chain_compute_wave(const ChainArgs& a, const bf16* const (&segbase)[8], int lane, unsigned& sink) {
    constexpr int N = total_items<ROLE, W>();
    u32x4 ring[RING][4];
    constexpr bool stream = (MODE & 2) != 0;
    auto issue = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr ItemRef r = item_ref<ROLE, W>(i);
        constexpr int chunk = W + 3 * r.j;                                       // round-robin over the unit's chunks
        const bf16* p = segbase[r.seg] + (long)chunk * 2048 + lane * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) ring_load(ring[i % RING][q], p + q * 512);
    };
    if constexpr (stream) static_for<0, (RING < N ? RING : N)>(issue);
    static_for<0, N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (first_of_phase<ROLE, W>(i)) __builtin_amdgcn_s_barrier();           // A: the service wave has gathered this phase's activations
        if constexpr (stream) {
            constexpr int younger = ((i + RING - 1 < N - 1) ? (i + RING - 1) : (N - 1)) - i;
            wait_ring<4 * younger>(ring[i % RING]);
#pragma unroll
            for (int q = 0; q < 4; ++q) sink ^= ring[i % RING][q][0] ^ ring[i % RING][q][3];    // stand-in for the four MFMAs of the chunk
            if constexpr (i + RING < N) issue(std::integral_constant<int, i + RING>{});
        }
        if constexpr (last_of_phase<ROLE, W>(i)) __builtin_amdgcn_s_barrier();            // B: this wave's partial sums of the phase are in LDS
    });
}

template <int ROLE, int MODE>
__device__ __forceinline__ void chain_service_wave(const ChainArgs& a, char* lds, int lane, unsigned& sink) {
    constexpr bool grid = (MODE & 1) != 0, xfer = (MODE & 4) != 0;
    // gather `chunks` 4 KiB chunk slots of a [16][K] bf16 matrix (rows 2K bytes apart) into LDS, one LDS-DMA instruction per 4 rows
    auto gather = [&](const bf16* src, int K, int chunk0, int chunks) {
        if (!xfer) return;
        for (int c = 0; c < chunks; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = q * 4 + (lane >> 4);
                glds16(src + (long)row * K + (long)(chunk0 + c) * 128 + (((lane & 15) ^ (row & 15)) << 3), lds + (c * 4 + q) * 1024);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto publish = [&](int bytes) {                                                  // write-through stores of this block's results
        if (!xfer) return;
        char* p = a.pub + (long)blockIdx.x * 4096;
        for (int o = lane * 16; o < bytes; o += 1024) {
            u32x4 v = {sink, (unsigned)o, 1u, 2u};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p + o), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto gbar = [&](unsigned k) { if (grid && lane == 0) chain_grid_barrier(a.bar, a.epoch0 + k, gridDim.x / 8); };
    auto phase = [&](const bf16* src, int K, int chunk0, int chunks, int pub_bytes) {
        gather(src, K, chunk0, chunks);
        __builtin_amdgcn_s_barrier();            // A
        __builtin_amdgcn_s_barrier();            // B
        if (xfer) sink ^= *(volatile unsigned*)(lds + lane * 4);                     // stand-in for the cross-wave reduction + epilogue arithmetic
        publish(pub_bytes);
    };
    if constexpr (ROLE == 0) {
        phase(a.a_o, 2048, 0, 16, 1024 + 512 + 64);            // o_proj: x_new fp32 tile, bf16(x_new * w) tile, sums of squares
        gbar(1);
        phase(a.a_x, 2048, 0, 16, 3 * 256);                     // gate|up x 3 units -> h tiles
        gbar(2);
        gbar(3);
        phase(a.a_x, 2048, 0, 16, 3 * 1024);                    // next layer's qkv half-units -> fp32 slabs
    } else {
        gbar(1);
        phase(a.a_x, 2048, 0, 16, 3 * 256);
        gbar(2);
        gather(a.a_h, 5632, 0, 22);                             // down_proj, K half 0
        __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
        phase(a.a_h, 5632, 22, 22, 1024 + 512 + 64);            // K half 1, then the residual / norm epilogue
        gbar(3);
        phase(a.a_x, 2048, 0, 16, 3 * 1024);
    }
}

template <int ROLE, int MODE>
__device__ __forceinline__ void chain_block(const ChainArgs& a, char* lds, unsigned* out) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x, t = b >> 1;
    const bf16* segbase[8];
    auto gu_unit = [&](int j) { int u = b + 256 * j; if (u >= 704) u = b; return a.wgu + (long)u * 16 * 2048; };    // dummy unit: a re-read (L2 hit)
    auto qkv_unit = [&](int j) { const int q = 3 * b + j; return a.wqkv + ((long)(q >> 1) * 16 + (q & 1) * 8) * 2048; };
    if constexpr (ROLE == 0) {
        segbase[0] = a.wo + (long)t * 16 * 2048;
        segbase[1] = gu_unit(0); segbase[2] = gu_unit(1); segbase[3] = gu_unit(2);
        segbase[4] = qkv_unit(0); segbase[5] = qkv_unit(1); segbase[6] = qkv_unit(2); segbase[7] = segbase[6];
    } else {
        segbase[0] = gu_unit(0); segbase[1] = gu_unit(1); segbase[2] = gu_unit(2);
        segbase[3] = a.wd + (long)t * 44 * 2048; segbase[4] = segbase[3] + (long)22 * 2048;
        segbase[5] = qkv_unit(0); segbase[6] = qkv_unit(1); segbase[7] = qkv_unit(2);
    }
    unsigned sink = 0;
    if (w == 0) chain_compute_wave<ROLE, 0, MODE>(a, segbase, lane, sink);
    else if (w == 1) chain_compute_wave<ROLE, 1, MODE>(a, segbase, lane, sink);
    else if (w == 2) chain_compute_wave<ROLE, 2, MODE>(a, segbase, lane, sink);
    else chain_service_wave<ROLE, MODE>(a, lds, lane, sink);
    if (sink == 0x9e3779b9u) out[threadIdx.x] = sink;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void chain_skel_kernel(ChainArgs a, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (blockIdx.x & 1) chain_block<1, MODE>(a, lds, out); else chain_block<0, MODE>(a, lds, out);
}
template <int MODE> void launch_chain_skel(hipStream_t s, const ChainArgs& a, unsigned* out, int LDS) {
    (void)PG_DYN_LDS(chain_skel_kernel<MODE>, LDS);
    hipLaunchKernelGGL(chain_skel_kernel<MODE>, dim3(256), dim3(256), LDS, s, a, out);
}

}  // namespace

// Measurement entry (tools/chain_skel.py): `iters` back-to-back launches of the skeleton over the rotating weights of `nl` layers
// (tiled decode copies allocated and filled here: 103 MB per layer, > 256 MB in total so the Infinity Cache cannot hold them).
extern "C" int pg_bench_chain_skeleton(int nl, int iters, int mode, float* us_out, unsigned* err_out) {
    if (mode < 0 || mode > 5 || nl < 1) return -1;
    const long no = 2048L * 2048, ngu = 11264L * 2048, nd = 2048L * 5632, nq = 6144L * 2048;
    std::vector<bf16*> wo(nl), wgu(nl), wd(nl), wq(nl);
    auto alloc = [&](bf16** p, long n) { if (hipMalloc((void**)p, n * 2) != hipSuccess) return false; hipMemset(*p, 0x11, n * 2); return true; };
    for (int l = 0; l < nl; ++l) if (!alloc(&wo[l], no) || !alloc(&wgu[l], ngu) || !alloc(&wd[l], nd) || !alloc(&wq[l], nq)) return -2;
    bf16 *ao, *ax, *ah; char* pub; ChainBar* bar; unsigned* out;
    hipMalloc((void**)&ao, 16 * 2048 * 2); hipMalloc((void**)&ax, 16 * 2048 * 2); hipMalloc((void**)&ah, 16 * 5632 * 2);
    hipMalloc((void**)&pub, 256 * 4096); hipMalloc((void**)&bar, sizeof(ChainBar)); hipMalloc((void**)&out, 4096);
    hipMemset(ao, 0, 16 * 2048 * 2); hipMemset(ax, 0, 16 * 2048 * 2); hipMemset(ah, 0, 16 * 5632 * 2); hipMemset(bar, 0, sizeof(ChainBar));
    const int LDS = 96 * 1024;
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned epoch = 0;
    for (int it = -10; it < iters; ++it) {
        if (it == 0) hipEventRecord(e0, s);
        const int l = (it + 10) % nl;
        ChainArgs a{wo[l], wgu[l], wd[l], wq[(l + 1) % nl], ao, ax, ah, pub, bar, epoch, mode};
        switch (mode & 7) {
            case 0: launch_chain_skel<0>(s, a, out, LDS); break; case 1: launch_chain_skel<1>(s, a, out, LDS); break;
            case 2: launch_chain_skel<2>(s, a, out, LDS); break; case 3: launch_chain_skel<3>(s, a, out, LDS); break;
            case 4: launch_chain_skel<4>(s, a, out, LDS); break; case 5: launch_chain_skel<5>(s, a, out, LDS); break;
            default: break;          // modes 6 / 7 (weight stream AND activation transfers together) faulted with a memory aperture violation in round 4
                                     // (profiles/r04_c) and were never root-caused: not instantiated any more, rejected below
        }
        epoch += 3;
    }
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    int rc = hipGetLastError() == hipSuccess ? 0 : -2;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (us_out) *us_out = ms * 1e3f / iters;
    ChainBar hb; hipMemcpy(&hb, bar, sizeof(hb), hipMemcpyDeviceToHost);
    if (err_out) *err_out = hb.err[0];
    for (int l = 0; l < nl; ++l) { hipFree(wo[l]); hipFree(wgu[l]); hipFree(wd[l]); hipFree(wq[l]); }
    hipFree(ao); hipFree(ax); hipFree(ah); hipFree(pub); hipFree(bar); hipFree(out);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
    return rc;
}
