"""GPU: operator-level parity of the HIP kernels (through the C ABI) against plain torch
fp32 references on the same seeded inputs.  Asymmetric random data so transposed MFMA
layouts cannot pass."""
import pytest
import torch
import torch.nn.functional as F

from conftest import get_engine

pytestmark = pytest.mark.gpu


def _eng(tiny_cfg, tiny_weights, dtype):
    return get_engine(tiny_cfg, tiny_weights, dtype)


def _round(t, dtype):
    return t.to(torch.bfloat16).float() if dtype == "bf16" else t


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_rmsnorm_with_splitk_partials(tiny_cfg, tiny_weights, dtype):
    e = _eng(tiny_cfg, tiny_weights, dtype)
    g = torch.Generator().manual_seed(0)
    M, H = 5, 256
    x = torch.randn(M, H, generator=g)
    part = torch.randn(3, M, H, generator=g)
    w = _round(1 + 0.1 * torch.randn(H, generator=g), dtype)
    xnew, out = e.op_rmsnorm(x, w, 1e-6, part)
    xr = x + part.sum(0)
    ref = w * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert torch.allclose(xnew.cpu(), xr, atol=1e-5)
    tol = 1e-5 if dtype == "f32" else 2e-2
    assert (out.float().cpu() - ref).abs().max() < tol * ref.abs().max()


@pytest.mark.parametrize("dtype,kind,M,N,K", [
    ("f32", 3, 37, 200, 72), ("f32", 3, 130, 64, 256),
    ("bf16", 1, 16, 768, 256), ("bf16", 1, 6, 256, 512), ("bf16", 1, 128, 512, 512), ("bf16", 1, 100, 1024, 256),
    ("bf16", 1, 200, 256, 256),
    ("bf16", 2, 300, 384, 192), ("bf16", 2, 128, 128, 64), ("bf16", 2, 1000, 200, 576), ("bf16", 2, 2048, 1024, 256),
])
def test_gemm_kinds(tiny_cfg, tiny_weights, dtype, kind, M, N, K):
    e = _eng(tiny_cfg, tiny_weights, dtype)
    g = torch.Generator().manual_seed(M * 7 + N)
    a = _round(torch.randn(M, K, generator=g), dtype)
    w = _round(torch.randn(N, K, generator=g) * torch.linspace(0.5, 2.0, N)[:, None], dtype)   # asymmetric
    out = e.op_gemm(a, w, kind).cpu()
    ref = a.double() @ w.double().t()
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-4 * ref.abs().max().item() + 1e-4, err


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("up,stride2,res", [(0, 0, False), (0, 0, True), (1, 0, False), (0, 1, False)])
def test_conv3x3_implicit_gemm(tiny_cfg, tiny_weights, dtype, up, stride2, res):
    e = _eng(tiny_cfg, tiny_weights, dtype)
    g = torch.Generator().manual_seed(3 + up + 2 * stride2)
    B, Hs, Ws, Cin, Cout = 2, 10, 12, 64, 128
    x = _round(torch.randn(B, Cin, Hs, Ws, generator=g), dtype)
    w = _round(torch.randn(Cout, Cin, 3, 3, generator=g) / 24, dtype)
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    if stride2:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, b, stride=2)
    else:
        ref = F.conv2d(xin, w, b, padding=1)
    r = None
    if res:
        r = _round(torch.randn(ref.shape, generator=g), dtype)
        ref = ref + r
    out = e.op_conv3x3(x.permute(0, 2, 3, 1).contiguous(), w, b, None if r is None else r.permute(0, 2, 3, 1).contiguous(),
                       up, stride2)
    out = out.float().cpu().permute(0, 3, 1, 2)
    tol = 1e-4 if dtype == "f32" else 2e-2
    assert (out - ref).abs().max() < tol * ref.abs().max()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("C,swish", [(64, True), (128, False), (512, True)])
def test_groupnorm_swish(tiny_cfg, tiny_weights, dtype, C, swish):
    e = _eng(tiny_cfg, tiny_weights, dtype)
    g = torch.Generator().manual_seed(C)
    B, Hs, Ws = 2, 9, 7
    x = _round(torch.randn(B, C, Hs, Ws, generator=g) * 2 + 0.5, dtype)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    out = e.op_groupnorm(x.permute(0, 2, 3, 1).contiguous(), gamma, beta, swish).float().cpu().permute(0, 3, 1, 2)
    tol = 2e-5 if dtype == "f32" else 2e-2
    assert (out - ref).abs().max() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 512), (8200, 2040, 192), (13000, 1024, 128), (32768, 512, 1024)])
def test_gemm_256_tile_path(tiny_cfg, tiny_weights, M, N, K):
    """Shapes that fill >= 200 tiles of 256x256 go through the persistent eight-phase kernel
    (gemm256.hip); ragged M / N exercise the clamped staging rows and the masked epilogue.
    Checked against fp64 and bit-for-bit against the 128x128 kernel (same K order)."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(M + N + K)
    a = _round(torch.randn(M, K, generator=g), "bf16")
    w = _round(torch.randn(N, K, generator=g) * torch.linspace(0.5, 2.0, N)[:, None], "bf16")
    e.set_option("gemm256", 1)
    out = e.op_gemm(a, w, 2).cpu()
    e.set_option("gemm256", 0)
    out128 = e.op_gemm(a, w, 2).cpu()
    e.set_option("gemm256", 1)
    ref = a.double() @ w.double().t()
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-4 * ref.abs().max().item() + 1e-4, err
    assert torch.equal(out, out128)


@pytest.mark.parametrize("up,stride2,res,B,Hs,Cin,Cout", [(0, 0, True, 6, 96, 64, 256), (1, 0, False, 6, 48, 128, 256),
                                                         (0, 1, False, 8, 192, 64, 256), (0, 0, False, 3, 96, 64, 512)])
def test_conv3x3_256_tile_path(tiny_cfg, tiny_weights, up, stride2, res, B, Hs, Cin, Cout):
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(11 + up + 2 * stride2 + Cin)
    x = _round(torch.randn(B, Cin, Hs, Hs, generator=g), "bf16")
    w = _round(torch.randn(Cout, Cin, 3, 3, generator=g) / 24, "bf16")
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, b, stride=2) if stride2 else F.conv2d(xin, w, b, padding=1)
    r = None
    if res:
        r = _round(torch.randn(ref.shape, generator=g), "bf16")
        ref = ref + r
    xn = x.permute(0, 2, 3, 1).contiguous()
    rn = None if r is None else r.permute(0, 2, 3, 1).contiguous()
    e.set_option("gemm256", 1)
    out = e.op_conv3x3(xn, w, b, rn, up, stride2)
    e.set_option("gemm256", 0)
    out128 = e.op_conv3x3(xn, w, b, rn, up, stride2)
    e.set_option("gemm256", 1)
    assert torch.equal(out, out128)
    o = out.float().cpu().permute(0, 3, 1, 2)
    assert (o - ref).abs().max() < 2e-2 * ref.abs().max()


@pytest.mark.parametrize("res,B,Hs,Ws,up", [(False, 4, 64, 128, 0), (True, 3, 96, 160, 0), (True, 1, 192, 192, 0),
                                            (False, 4, 32, 64, 1), (False, 2, 96, 96, 1)])
def test_conv3x3_halo_tile_path(tiny_cfg, tiny_weights, res, B, Hs, Ws, up):
    """Cin = Cout = 128, sides multiple of the 8 x 32 tile, >= 128 tiles: the direct convolution with the
    LDS-resident input halo tile (conv_halo.hip).  Image borders (zero page), tile seams and the residual
    epilogue against torch; bit-for-bit against the implicit-GEMM kernel (same K order)."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(7 + Hs)
    x = _round(torch.randn(B, 128, Hs, Ws, generator=g), "bf16")
    w = _round(torch.randn(128, 128, 3, 3, generator=g) / 34, "bf16")
    b = torch.randn(128, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, w, b, padding=1)
    r = None
    if res:
        r = _round(torch.randn(ref.shape, generator=g), "bf16")
        ref = ref + r
    xn = x.permute(0, 2, 3, 1).contiguous()
    rn = None if r is None else r.permute(0, 2, 3, 1).contiguous()
    e.set_option("conv_halo", 1)
    out = e.op_conv3x3(xn, w, b, rn, up, 0)
    e.set_option("conv_halo", 0)
    out_gemm = e.op_conv3x3(xn, w, b, rn, up, 0)
    e.set_option("conv_halo", 1)
    assert torch.equal(out, out_gemm)
    o = out.float().cpu().permute(0, 3, 1, 2)
    assert (o - ref).abs().max() < 2e-2 * ref.abs().max()


def test_sampler_uniform_strictly_inside_unit_interval(tiny_cfg, tiny_weights):
    """with 24 random bits (x + 0.5) / 2^24 rounds to exactly 1.0 for the top counter value
    and the Gumbel term becomes +inf.  Sweep the top / bottom counter values of the generator output through
    the sampler's own transform: u must stay in (0, 1) and the Gumbel noise finite."""
    e = _eng(tiny_cfg, tiny_weights, "f32")
    top = [(1 << 64) - 1 - i for i in range(4096)] + [((1 << 23) - 1 - i) << 41 for i in range(64)]
    bot = list(range(4096)) + [i << 41 for i in range(64)]
    g = torch.Generator().manual_seed(0)
    rnd = torch.randint(-(1 << 62), 1 << 62, (1 << 16,), generator=g, dtype=torch.int64).tolist()
    bits = torch.tensor([b - (1 << 64) if b >= (1 << 63) else b for b in top + bot] + rnd, dtype=torch.int64)
    u, gum = e.op_uniform(bits)
    u, gum = u.cpu(), gum.cpu()
    assert (u > 0).all() and (u < 1).all()
    assert torch.isfinite(gum).all()
    assert u.max().item() == 1.0 - 2.0 ** -24 and u.min().item() == 2.0 ** -24
    # the transform is the stated one: 23 high bits, centred
    want = ((bits.numpy().astype("uint64") >> 41).astype("float64") + 0.5) / 2 ** 23
    assert (u.double().numpy() == want).all()
    assert abs(u[-(1 << 16):].mean().item() - 0.5) < 0.01


@pytest.mark.parametrize("M,N,K", [(128, 6144, 2048), (128, 2048, 2048), (128, 2048, 5632), (128, 16384, 2048),
                                   (64, 6144, 2048), (16, 2048, 5632), (8, 2048, 2048), (100, 512, 1408)])
def test_skinny_gemm_on_tiled_decode_weights(tiny_cfg, tiny_weights, M, N, K):
    """The decode layout: W re-tiled [n-tile 16][k-chunk 128][k-step][lane][8] exactly like
    pg_finalize_weights does, streamed by gemm_skinny3_kernel<.., TILED> at the Janus-Pro shapes the bs=64 bench
    runs (qkv NCK 8, o NCK 4, down NCK 11, gen_head) -- against fp64 and bit-for-bit against the row-major path."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(M + N + K)
    a = _round(torch.randn(M, K, generator=g), "bf16")
    w = _round(torch.randn(N, K, generator=g) * torch.linspace(0.5, 2.0, N)[:, None] * 0.05, "bf16")
    out_t = e.op_gemm(a, w, 4).cpu()
    out_r = e.op_gemm(a, w, 1).cpu()
    ref = a.double() @ w.double().t()
    def where(d, tol):                                             # forensics for a rare failure: which rows / 16-column tiles are off
        bad = (d > tol).nonzero()
        return {"n": int(bad.shape[0]), "rows": sorted(set(bad[:, 0].tolist()))[:16], "n_tiles": sorted(set((bad[:, 1] // 16).tolist()))[:16],
                "max": float(d.max())}
    tol = 2e-4 * ref.abs().max().item() + 1e-4
    d_t, d_r = (out_t.double() - ref).abs(), (out_r.double() - ref).abs()
    assert d_r.max().item() < tol, ("row-major path vs fp64", where(d_r, tol))
    assert d_t.max().item() < tol, ("tiled path vs fp64", where(d_t, tol))
    tol2 = 1e-4 * ref.abs().max().item() + 1e-5
    d2 = (out_t - out_r).abs()
    assert d2.max().item() < tol2, ("tiled vs row-major", where(d2, tol2))     # same products, M-block geometry may differ


@pytest.mark.parametrize("M,I,K", [(128, 5632, 2048), (64, 5632, 2048), (16, 5632, 2048), (8, 512, 256)])
def test_decode_swiglu_gemm_epilogue(tiny_cfg, tiny_weights, M, I, K):
    """gate|up GEMM with SwiGLU fused into the epilogue (gemm_skinny3_kernel<.., NCK 16, EPI 1, TILED>) at the
    bench shape vs torch: h = silu(x Wg^T) * (x Wu^T), bf16 out."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(M + I)
    a = _round(torch.randn(M, K, generator=g), "bf16")
    wg = _round(torch.randn(I, K, generator=g) * 0.03, "bf16")
    wu = _round(torch.randn(I, K, generator=g) * torch.linspace(0.5, 2.0, I)[:, None] * 0.03, "bf16")
    out = e.op_swiglu_gemm(a, wg, wu).float().cpu()
    ref = (F.silu(a.double() @ wg.double().t()) * (a.double() @ wu.double().t())).float()
    assert (out - ref).abs().max() < 1e-2 * ref.abs().max()          # one bf16 rounding of h


@pytest.mark.parametrize("H,NV", [(2048, 2), (4096, 4)])
def test_rmsnorm_wide_rows(tiny_cfg, tiny_weights, H, NV):
    """rmsnorm512_kernel (512 threads per 2048-wide row: the instantiation every decode step runs at H = 2048) and the generic
    rmsnorm_kernel at H = 4096, with 4 split-K slabs."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(H)
    M = 128
    x = torch.randn(M, H, generator=g)
    part = torch.randn(4, M, H, generator=g)
    w = _round(1 + 0.1 * torch.randn(H, generator=g), "bf16")
    xnew, out = e.op_rmsnorm(x, w, 1e-6, part)
    xr = x + part.sum(0)
    ref = w * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert torch.allclose(xnew.cpu(), xr, atol=1e-5)
    assert (out.float().cpu() - ref).abs().max() < 1e-2 * ref.abs().max()


@pytest.mark.parametrize("M,N,K", [(128, 2048, 2048), (128, 2048, 5632), (64, 6144, 2048), (64, 2048, 5632), (16, 6144, 2048), (8, 2048, 2048)])
def test_stream_gemm_is_deterministic_under_repetition(tiny_cfg, tiny_weights, M, N, K):
    """The v4 decode GEMM (x tile by LDS-DMA, counted vmcnt + raw s_barrier) on the shapes the default dispatch sends to
    it: 60 launches must give bit-identical slabs (a stale LDS read would differ by a whole 128-wide K chunk) and agree
    with fp64."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a = _round(torch.randn(M, K, generator=g), "bf16")
    w = _round(torch.randn(N, K, generator=g) * 0.05, "bf16")
    first = e.op_gemm(a, w, 4)
    ref = a.double() @ w.double().t()
    assert (first.cpu().double() - ref).abs().max().item() < 2e-4 * ref.abs().max().item() + 1e-4
    for _ in range(60):
        assert torch.equal(e.op_gemm(a, w, 4), first)


@pytest.mark.parametrize("M,N,K,S", [(128, 6144, 2048, 2), (128, 2048, 2048, 4), (128, 11264, 2048, 2), (128, 2048, 5632, 4),
                                     (64, 6144, 2048, 2), (32, 2048, 5632, 4), (16, 6144, 2048, 4), (16, 2048, 2048, 8)])
def test_decode_gemm_back_to_back_race_screen(M, N, K, S):
    """The production decode-GEMM dispatch (tiled weight copy) launched back to back with NO host sync between launches -- the
    harness that exposed a hazard of two co-resident 512-thread LDS-DMA blocks (profiles/r02_d_*) while synchronous
    per-launch checks stayed clean.  Every launch of 4 x 100 must match the 1-deep reference kernel."""
    import ctypes as C
    import os
    from plangen_amd import _lib
    lib = _lib.load_diag()          # the production object code + the pg_bench_* harness (libplangen_diag.so)
    lib.pg_bench_skinny_verify.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)] * 2
    for _ in range(4):
        md, mr = C.c_float(0), C.c_float(0)
        rc = lib.pg_bench_skinny_verify(M, N, K, 2, S, 1, 100, C.byref(md), C.byref(mr))
        assert rc == 0
        assert md.value <= 2e-3 * mr.value, (md.value, mr.value)


@pytest.mark.parametrize("bg_mode", [2, 0])
@pytest.mark.parametrize("M,N,K,S", [(128, 2048, 2048, 4), (128, 2048, 5632, 4), (64, 6144, 2048, 2), (64, 11264, 2048, 1), (32, 2048, 5632, 4),
                                     (16, 6144, 2048, 2)])
def test_decode_gemm_under_concurrent_memory_load(M, N, K, S, bg_mode):
    """Round 4: the production decode-GEMM dispatch while ANOTHER STREAM streams 768 MB (bg_mode 2: register loads, 0: LDS-DMA).
    This is the condition that turned the write-through epilogue's missing store-data wait states (an inline-asm
    `global_store_dwordx4` whose data register hipcc recycled one instruction later) from a ~0.5 % cold-launch fault into a
    100 % one: rows 12-15 of every m-tile but the block's last lost a split's contribution (what rounds 2-3 read as a stale
    LDS-DMA piece).  Before the fix 6 of 6 batches failed here at 32 / 64 / 128 rows; every launch must match the reference kernel."""
    import ctypes as C
    import os
    from plangen_amd import _lib
    lib = _lib.load_diag()          # the production object code + the pg_bench_* harness (libplangen_diag.so)
    lib.pg_bench_skinny_verify.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float)] * 2
    for _ in range(3):
        assert lib.pg_bench_background(768, 40, 256, 16, bg_mode) == 0
        md, mr = C.c_float(0), C.c_float(0)
        rc = lib.pg_bench_skinny_verify(M, N, K, 2, S, 1, 100, C.byref(md), C.byref(mr))
        lib.pg_bench_background_join()
        assert rc == 0
        assert md.value <= 2e-3 * mr.value, (md.value, mr.value)


@pytest.mark.parametrize("M,N,K,iters", [(64, 6144, 2048, 120), (128, 2048, 2048, 60), (128, 2048, 5632, 60), (16, 6144, 2048, 40), (16, 2048, 5632, 40)])
def test_decode_gemm_cold_launches(tiny_cfg, tiny_weights, M, N, K, iters):
    """(round 3: also the M = 128 narrow-N shapes the bs=64 loop sends to gemm_sk4_kernel<4,...> -- o 128x2048x2048 and down
    128x2048x5632, S = 4 -- and the 16-row blocks of BASELINE configs[1].)  ONE launch of the tiled decode GEMM at a time on an otherwise IDLE GPU (fresh device copies of host operands, 0.15 s pause between
    launches), 120 times: the shape and path that read a wave's last LDS-DMA piece stale in ~0.5 % of such cold launches before every chunk
    was retired one barrier ahead of its first read.  Back-to-back launches never showed it, with or without fresh
    memory; with the previous kernel this test fails in roughly one run of three at 400 iterations -- the structural guard is
    tests/test_isa_check.py, this is the empirical one."""
    e = _eng(tiny_cfg, tiny_weights, "bf16")
    g = torch.Generator().manual_seed(11)
    sets = []
    for _ in range(6):
        a = _round(torch.randn(M, K, generator=g), "bf16")
        w = _round(torch.randn(N, K, generator=g) * 0.05, "bf16")
        sets.append((a, w, (a.cuda().double() @ w.cuda().double().t())))
    junk = []
    import time
    for it in range(iters):
        a, w, ref = sets[it % len(sets)]
        if it % 3 == 0:
            junk.append(torch.empty((1 + it % 7) * 1_000_003, device="cuda"))
        if len(junk) > 4:
            junk.pop(0)
        time.sleep(0.15)                                           # the GPU goes idle between launches: that, not fresh memory, is what 'cold' needs
        out = e.op_gemm(a, w, 4)                                   # host tensors: fresh device copies every call
        d = (out.double() - ref).abs()
        tol = 2e-4 * ref.abs().max().item() + 1e-4
        if d.max().item() >= tol:
            idx = (d >= tol).nonzero()
            raise AssertionError((it, d.max().item(), sorted(set(idx[:, 0].tolist()))[:16], sorted(set((idx[:, 1] // 16).tolist()))[:16]))


def test_hardware_bf16_conversion_equals_software_rne_for_every_float():
    """Round 4: every fp32 -> bf16 conversion of the library (epilogues, norms, attention outputs, P in the flash kernels, VQ activations) is
    gfx950's v_cvt_pk_bf16_f32 instead of a 6-instruction software round-to-nearest-even.  All 2^32 bit patterns: identical for every
    non-NaN input (single and packed form), NaNs stay NaNs."""
    import ctypes as C
    import os
    from plangen_amd import _lib
    lib = _lib.load_diag()          # the production object code + the pg_bench_* harness (libplangen_diag.so)
    bad, nan_lost = C.c_ulonglong(1), C.c_ulonglong(1)
    assert lib.pg_bench_bf16_cvt_check(C.byref(bad), C.byref(nan_lost)) == 0
    assert bad.value == 0 and nan_lost.value == 0, (bad.value, nan_lost.value)


@pytest.mark.parametrize("mode", [4, 5, 6, 1, 12, 13, 9])
@pytest.mark.parametrize("M,N,K", [(13285, 2048, 2048), (6600, 2048, 5632), (6100, 6144, 2048)])
def test_gemm256_tile_heights_are_bit_identical_to_the_128_tile_kernel(M, N, K, mode):
    """Round 5: gemm256_kernel's tile height is chosen per launch (256 / 224 / 192 rows = `gemm256` 4 / 5 / 6; 1 = pick_tile_height()) and a K tile
    runs in two phases (default) or four (+8).  Same K order in every form: each of 3 launches must equal the 128x128 kernel bit for bit (ragged
    last tile rows included)."""
    import ctypes as C
    from plangen_amd import _lib
    lib = _lib.load_diag()
    lib.pg_bench_gemm.argtypes = [C.c_int] * 10 + [C.POINTER(C.c_float)] * 2
    us, md = C.c_float(0), C.c_float(-1)
    assert lib.pg_bench_gemm(M, N, K, 0, 0, 0, 0, mode, 2, 3, C.byref(us), C.byref(md)) == 0
    assert md.value == 0.0, md.value


@pytest.mark.parametrize("M,N,K,S", [(128, 2048, 2048, 4), (128, 2048, 5632, 4), (64, 2048, 5632, 11), (16, 2048, 2048, 4)])
def test_fused_norm_producer_experiment_matches_the_production_pair(M, N, K, S):
    """Round 5 experiment kept in libplangen_diag.so (profiles/r05_b, measured slower and NOT integrated): the split-K producer whose last-arrival
    block reduces the slabs, adds the residual and emits bf16(x w) + fixed-point row sums of squares.  Its cross-XCD publication (write-through
    stores, device-scope ticket, L2-bypassing reloads) must reproduce the production GEMM + rmsnorm512 arithmetic: x bit-identical, xw and ssq exact."""
    import ctypes as C
    from plangen_amd import _lib
    lib = _lib.load_diag()
    lib.pg_bench_fused_norm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)] * 3 + [C.POINTER(C.c_uint)]
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    bad = (C.c_uint * 4)()
    assert lib.pg_bench_fused_norm(M, N, K, S, 20, C.byref(a), C.byref(b), C.byref(c), bad) == 0
    assert (bad[0], bad[1], bad[2]) == (0, 0, 0), list(bad)


@pytest.mark.parametrize("B,Hi,Cin,Cout,up,plain,res", [
    (48, 24, 512, 512, 0, 0, 1),    # 24^2: HW = 576 = 9 x 64 (a 256-row tile straddles images; 64-row chunks do not), 16 channels per group, 216 tiles
    (48, 24, 512, 512, 0, 1, 1),    # 1x1 (AttnBlock.proj_out, vq_model.py:388) with the fp32 residual
    (24, 48, 512, 256, 0, 0, 0),    # 8 channels per group
    (12, 24, 512, 512, 1, 0, 0),    # nearest-2x upsample folded into the loader (vq_model.py:417-427): output 48^2
    (6, 96, 256, 256, 0, 0, 1),     # 96^2, Cin 256
])
def test_gemm256_epilogue_groupnorm_statistics_match_the_statistics_kernel(B, Hi, Cin, Cout, up, plain, res):
    """Round 6: the 256x256 kernel's convolutions emit the GroupNorm(32) partial sums of the tensor they store (as the halo
    kernel does), which deletes the separate gn_stats pass.  (mean, rstd) of every (image, group) from those partials must equal the statistics
    kernel's on the stored tensor; a slot nobody wrote would surface as NaN (the workspace is pre-filled with 0xff)."""
    import ctypes as C
    from plangen_amd import _lib
    lib = _lib.load_diag()
    lib.pg_bench_conv_gn_check.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    ns, dm, dr = C.c_int(0), C.c_float(0), C.c_float(0)
    assert lib.pg_bench_conv_gn_check(B, Hi, Hi, Cin, Cout, up, plain, res, 1, C.byref(ns), C.byref(dm), C.byref(dr)) == 0
    hw = (Hi << (0 if plain else up)) ** 2
    assert ns.value == hw // 64, (ns.value, hw)                     # the 256x256 kernel took the shape and reported one split per 64-row chunk
    print(f"epilogue GroupNorm statistics: {ns.value} splits per image, max |mean diff| {dm.value:.2e}, max rel rstd diff {dr.value:.2e}")
    assert dm.value < 2e-5 and dr.value < 2e-5, (dm.value, dr.value)
    # the 128x128 kernel emits no partials: the caller then runs the statistics kernel (nsplit 0)
    assert lib.pg_bench_conv_gn_check(B, Hi, Hi, Cin, Cout, up, plain, res, 0, C.byref(ns), C.byref(dm), C.byref(dr)) == 0
    assert ns.value == 0
