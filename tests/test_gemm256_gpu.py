"""GPU parity tests of the 256x256 eight-phase GEMM kernel (csrc/gemm256.hip), pinned with tile_hint=256, against a
plain PyTorch fp32 reference of the same op (bf16 tolerance 1e-2 of max|ref|) and, bit for bit where the arithmetic is
the same, against repeated launches of itself (race screen of the LDS-DMA ring).  Run on the MI355X box: pytest -m gpu."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert math.isfinite(e) and e <= tol, f"{what}: max-abs error / max-abs ref = {e:.3e} > {tol}"


def rn(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


# K-tile counts 1, 2, 3, 4, 5, 12 (odd / even ring parity, prologue longer than the problem), ragged M / N edges
SHAPES = [(256, 256, 64), (256, 256, 128), (512, 256, 192), (256, 512, 256), (300, 264, 320), (1576, 768, 768), (1000, 2304, 768),
          (520, 1032, 2048), (264, 264, 64)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm256_forward(dev, M, N, K):
    from youku_mplug_amd import ops
    a, w = rn(M, K, dev=dev, seed=1), rn(N, K, dev=dev, seed=2)
    out = ops.gemm(a, w, M, N, K, tile_hint=256)
    close(out, a.float() @ w.float().t(), 1e-2, "Y = X W^T")


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm256_dgrad(dev, M, N, K):
    from youku_mplug_amd import ops
    dy, w = rn(M, K, dev=dev, seed=3), rn(K, N, dev=dev, seed=4)
    out = ops.gemm(dy, w, M, N, K, trans_b=True, tile_hint=256)
    close(out, dy.float() @ w.float(), 1e-2, "dX = dY W")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (768, 2304, 1600), (264, 328, 192), (768, 768, 6272), (2304, 768, 12608), (3072, 768, 1024)])
def test_gemm256_wgrad(dev, M, N, K):
    from youku_mplug_amd import ops
    dy, x = rn(K, M, dev=dev, seed=5), rn(K, N, dev=dev, seed=6)
    out = ops.gemm(dy, x, M, N, K, trans_a=True, trans_b=True, tile_hint=256)
    close(out, dy.float().t() @ x.float(), 1e-2, "dW = dY^T X")


def test_gemm256_matches_128_kernel(dev):
    """Same inputs through both tile kernels: the fp32 accumulation order differs (16x16x32 vs 32x32x16 MFMA blocks,
    split-K partition), so the bf16 results agree to about one ulp."""
    from youku_mplug_amd import ops
    for (M, N, K, ta, tb) in [(1576, 768, 768, 0, 0), (1024, 512, 2048, 0, 1), (768, 512, 3200, 1, 1)]:
        a = rn(K, M, dev=dev, seed=40) if ta else rn(M, K, dev=dev, seed=40)
        b = rn(K, N, dev=dev, seed=41) if tb else rn(N, K, dev=dev, seed=41)
        o1 = ops.gemm(a, b, M, N, K, trans_a=bool(ta), trans_b=bool(tb), tile_hint=128)
        o2 = ops.gemm(a, b, M, N, K, trans_a=bool(ta), trans_b=bool(tb), tile_hint=256)
        close(o2, o1, 8e-3, "256 vs 128 kernel")   # one bf16 ulp at the largest magnitudes


def test_gemm256_bitwise_deterministic(dev):
    """Race screen: 40 launches per shape must be bit-identical (an early LDS read of a ring unit whose DMA has not
    landed shows up as a rare differing tile)."""
    from youku_mplug_amd import ops
    for (M, N, K, ta, tb) in [(1576, 768, 768, 0, 0), (4096, 2304, 768, 0, 0), (1024, 8192, 2048, 0, 0), (1576, 768, 2304, 0, 1),
                              (768, 2304, 1600, 1, 1), (4096, 4096, 4096, 0, 0)]:
        a = rn(K, M, dev=dev, seed=90) if ta else rn(M, K, dev=dev, seed=90)
        b = rn(K, N, dev=dev, seed=91) if tb else rn(N, K, dev=dev, seed=91)
        ref = ops.gemm(a, b, M, N, K, trans_a=bool(ta), trans_b=bool(tb), tile_hint=256).clone()
        for _ in range(40):
            out = ops.gemm(a, b, M, N, K, trans_a=bool(ta), trans_b=bool(tb), tile_hint=256)
            assert torch.equal(out, ref), (M, N, K, ta, tb)


@pytest.mark.parametrize("M,N,K,ta,tb", [(8192, 2304, 768, 0, 0), (10240, 2048, 128, 0, 0), (9000, 2304, 256, 0, 1), (5120, 6144, 2048, 0, 0),
                                          (2304, 768, 50432, 1, 1), (8200, 8200, 128, 0, 0)])
def test_gemm256_persistent_multi_item(dev, M, N, K, ta, tb):
    """More than 256 (tile, split) items: a workgroup walks several tiles and its DMA ring runs across the tile
    boundaries (csrc/gemm256.hip).  Every element against fp32, then a 20-launch bitwise race screen."""
    from youku_mplug_amd import ops
    a = rn(K, M, dev=dev, seed=60) if ta else rn(M, K, dev=dev, seed=60)
    b = rn(K, N, dev=dev, seed=61, scale=0.1) if tb else rn(N, K, dev=dev, seed=61, scale=0.1)
    bias, res = rn(N, dev=dev, seed=62), rn(M, N, dev=dev, seed=63)
    af = a.float().t() if ta else a.float()
    bf = b.float() if tb else b.float().t()
    if ta:
        out = ops.gemm(a, b, M, N, K, trans_a=True, trans_b=True, tile_hint=256)
        close(out, af @ bf, 1e-2, "persistent wgrad")
        kw = dict(trans_a=True, trans_b=True)
    else:
        kw = dict(trans_b=bool(tb), bias=bias, residual=res)
        out = ops.gemm(a, b, M, N, K, tile_hint=256, **kw)
        close(out, af @ bf + bias.float() + res.float(), 1e-2, "persistent + bias + residual")
    first = out.clone()
    for _ in range(20):
        assert torch.equal(ops.gemm(a, b, M, N, K, tile_hint=256, **kw), first)


def test_gemm256_large_vs_fp32(dev):
    """Full-size check against fp32 matmul on the device (config-B shapes, every output element)."""
    from youku_mplug_amd import ops
    for (M, N, K) in [(50432, 768, 768), (5120, 6144, 2048), (4096, 4096, 4096)]:
        a, w = rn(M, K, dev=dev, seed=21), rn(N, K, dev=dev, seed=22, scale=0.05)
        out = ops.gemm(a, w, M, N, K, tile_hint=256)
        ref = a.float() @ w.float().t()
        close(out, ref, 1e-2, f"{M}x{N}x{K}")


def test_gemm256_epilogues(dev):
    from youku_mplug_amd import ops
    M, N, K = 600, 512, 192
    a, w, bias, res = rn(M, K, dev=dev, seed=7), rn(N, K, dev=dev, seed=8, scale=0.1), rn(N, dev=dev, seed=9), rn(M, N, dev=dev, seed=10)
    z_ref = (a.float() @ w.float().t() + bias.float())
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out = ops.gemm(a, w, M, N, K, bias=bias, act=ops.ACT_GELU_ERF, preact_out=pre, tile_hint=256)
    close(pre, z_ref, 1e-2, "preact")
    close(out, F.gelu(pre.float()), 1e-2, "gelu_erf(bf16(z))")
    out = ops.gemm(a, w, M, N, K, bias=bias, act=ops.ACT_GELU_TANH, tile_hint=256)
    close(out, F.gelu(z_ref.bfloat16().float(), approximate="tanh"), 1e-2, "gelu_tanh")
    out = ops.gemm(a, w, M, N, K, bias=bias, act=ops.ACT_RELU, tile_hint=256)
    close(out, F.relu(z_ref), 1e-2, "relu")
    out = ops.gemm(a, w, M, N, K, bias=bias, residual=res, tile_hint=256)
    close(out, z_ref + res.float(), 1e-2, "bias+residual")
    z = rn(M, N, dev=dev, seed=11)
    for act, approx in ((ops.ACT_GELU_ERF, "none"), (ops.ACT_GELU_TANH, "tanh")):
        zz = z.float().requires_grad_(True)
        F.gelu(zz, approximate=approx).sum().backward()
        out = ops.gemm(a, w, M, N, K, act_bwd_z=z, act_bwd=act, tile_hint=256)
        close(out, (a.float() @ w.float().t()) * zz.grad, 1e-2, f"gelu' {approx}")
    alpha = torch.tensor(0.5, device=dev)
    base = res.clone()
    ops.gemm(a, w, M, N, K, out=base, alpha_dev=alpha, accumulate=True, tile_hint=256)
    close(base, res.float() + 0.5 * (a.float() @ w.float().t()), 1e-2, "alpha_dev+accumulate")
    o32 = ops.gemm(a, w, M, N, K, out_f32=True, tile_hint=256)
    close(o32, a.float() @ w.float().t(), 2e-3, "fp32 output")
    # dropout: pure function of (seed, offset, index), same mask as the 128 kernel
    d256 = ops.gemm(a, w, M, N, K, residual=res, dropout_p=0.25, seed=1234, offset=77, tile_hint=256).float() - res.float()
    d128 = ops.gemm(a, w, M, N, K, residual=res, dropout_p=0.25, seed=1234, offset=77, tile_hint=128).float() - res.float()
    assert torch.equal(d256 == 0, d128 == 0) or ((d256 == 0) != (d128 == 0)).float().mean().item() < 1e-4


def test_gemm256_row_maps_and_fused_bias_grad(dev):
    from youku_mplug_amd import ops
    B, T, N1, D, Nout = 4, 8, 17, 256, 512          # token rows of a [B*T, 1+16, D] stream
    n = N1 - 1
    rows = B * T * n                                 # 512
    tok = (n, N1, 1)
    x, w = rn(B * T * N1, D, dev=dev, seed=12), rn(Nout, D, dev=dev, seed=13)
    out = torch.zeros(B * T * N1, Nout, dtype=torch.bfloat16, device=dev)
    ops.gemm(x, w, rows, Nout, D, out=out, amap=tok, cmap=tok, tile_hint=256)
    ref = x.float() @ w.float().t()
    mask = torch.ones(B * T * N1, dtype=torch.bool, device=dev)
    mask[::N1] = False
    close(out[mask], ref[mask], 1e-2, "mapped rows")
    assert out[~mask].abs().max().item() == 0.0, "cls slots must be untouched"
    dy = rn(B * T * N1, Nout, dev=dev, seed=14)
    bsum = torch.empty(Nout, dtype=torch.bfloat16, device=dev)
    dw = ops.gemm(dy, x, Nout, D, rows, trans_a=True, trans_b=True, lda=Nout, ldb=D, kmap=tok, colsum_out=bsum, tile_hint=256)
    close(dw, dy.float()[mask].t() @ x.float()[mask], 1e-2, "wgrad with kmap")
    close(bsum, dy.float()[mask].sum(0), 1e-2, "fused bias grad with kmap")
    # dgrad with a mapped A operand and mapped reduction rows is not a Linear pass; plain kmap-free fused bias grad, split-K
    K = 6272
    dy2, x2 = rn(K, 768, dev=dev, seed=95), rn(K, 512, dev=dev, seed=96)
    bs2 = torch.empty(768, dtype=torch.bfloat16, device=dev)
    dw2 = ops.gemm(dy2, x2, 768, 512, K, trans_a=True, trans_b=True, colsum_out=bs2, tile_hint=256)
    close(dw2, dy2.float().t() @ x2.float(), 1e-2, "dW split-K")
    close(bs2, dy2.float().sum(0), 1e-2, "fused bias grad split-K")


@pytest.mark.parametrize("rows", [192, 160])
@pytest.mark.parametrize("M,N,K", [(160, 256, 64), (192, 256, 128), (5120, 2048, 256), (1000, 520, 192), (333, 264, 320), (2560, 768, 2048)])
def test_gemm256_short_tile_rows(dev, rows, M, N, K):
    """The 192- and 160-row tile variants of the eight-phase kernel (same ring, the second A unit of a wave row partly
    dead): every element against fp32, bit-identical to the 256-row tile (same K order per element), ragged M / N,
    and the fused epilogues the GPT shapes use (tanh-GELU + pre-activation, GELU', dropout + residual)."""
    from youku_mplug_amd import ops
    from youku_mplug_amd.ops import ACT_GELU_TANH
    a, w = rn(M, K, dev=dev, seed=71), rn(N, K, dev=dev, seed=72, scale=0.2)
    bias, res = rn(N, dev=dev, seed=73), rn(M, N, dev=dev, seed=74)
    out = ops.gemm(a, w, M, N, K, tile_hint=rows)
    close(out, a.float() @ w.float().t(), 1e-2, f"{rows}-row tile")
    assert torch.equal(out, ops.gemm(a, w, M, N, K, tile_hint=256))
    for _ in range(10):
        assert torch.equal(out, ops.gemm(a, w, M, N, K, tile_hint=rows))
    z1, z2 = torch.empty_like(out), torch.empty_like(out)
    h1 = ops.gemm(a, w, M, N, K, bias=bias, act=ACT_GELU_TANH, preact_out=z1, tile_hint=rows)
    h2 = ops.gemm(a, w, M, N, K, bias=bias, act=ACT_GELU_TANH, preact_out=z2, tile_hint=256)
    assert torch.equal(h1, h2) and torch.equal(z1, z2)
    g1 = ops.gemm(a, w, M, N, K, act_bwd_z=res, act_bwd=ACT_GELU_TANH, tile_hint=rows)
    g2 = ops.gemm(a, w, M, N, K, act_bwd_z=res, act_bwd=ACT_GELU_TANH, tile_hint=256)
    assert torch.equal(g1, g2)
    d1 = ops.gemm(a, w, M, N, K, bias=bias, residual=res, dropout_p=0.1, seed=11, offset=5, tile_hint=rows)
    d2 = ops.gemm(a, w, M, N, K, bias=bias, residual=res, dropout_p=0.1, seed=11, offset=5, tile_hint=256)
    assert torch.equal(d1, d2)
    r1 = ops.gemm(a, w, M, N, K, bias=bias, residual=res, tile_hint=rows)
    assert torch.equal(r1, ops.gemm(a, w, M, N, K, bias=bias, residual=res, tile_hint=256))


@pytest.mark.parametrize("M,N,K,tb", [(50432, 768, 768, 0), (50176, 768, 768, 0), (50432, 768, 2304, 1), (5120, 8192, 2048, 0),
                                       (5120, 8192, 2048, 1), (12352, 768, 192, 1), (40000, 520, 128, 0)])
def test_gemm256_row_bands(dev, M, N, K, tb):
    """Auto tile choice cuts the rows into bands of 256- / 192- / 160-row tiles, one launch each (csrc/gemm256.hip,
    plan_bands).  Every element keeps its K order, so each epilogue must be bit-identical to the pinned 256-row launch;
    fp32 check on top, and the plan the library reports must be a multi-band one for the ViT / GPT shapes it exists for."""
    from youku_mplug_amd import _lib, ops
    from youku_mplug_amd.ops import ACT_GELU_ERF
    import ctypes
    out6 = (ctypes.c_int * 6)()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nb = _lib.lib().mpv_gemm_plan_bands(M, N, K, ncu, 0, 0, out6)
    if (M, N) in ((50432, 768), (5120, 8192)):
        assert nb >= 2, f"expected a multi-band plan for {M}x{N}x{K}: {list(out6)}"
    a = rn(M, K, dev=dev, seed=81)
    w = rn(K, N, dev=dev, seed=82, scale=0.1) if tb else rn(N, K, dev=dev, seed=82, scale=0.1)
    bias, res = rn(N, dev=dev, seed=83), rn(M, N, dev=dev, seed=84)
    kw = dict(trans_b=bool(tb))
    o0 = ops.gemm(a, w, M, N, K, **kw)
    o256 = ops.gemm(a, w, M, N, K, tile_hint=256, **kw)
    assert torch.equal(o0, o256), "plain"
    close(o0, a.float() @ (w.float() if tb else w.float().t()), 1e-2, "bands vs fp32")
    assert torch.equal(ops.gemm(a, w, M, N, K, bias=bias, residual=res, **kw), ops.gemm(a, w, M, N, K, bias=bias, residual=res, tile_hint=256, **kw))
    d0 = ops.gemm(a, w, M, N, K, bias=bias, residual=res, dropout_p=0.1, seed=5, offset=9, **kw)
    assert torch.equal(d0, ops.gemm(a, w, M, N, K, bias=bias, residual=res, dropout_p=0.1, seed=5, offset=9, tile_hint=256, **kw))
    assert torch.equal(ops.gemm(a, w, M, N, K, act_bwd_z=res, act_bwd=ACT_GELU_ERF, **kw),
                       ops.gemm(a, w, M, N, K, act_bwd_z=res, act_bwd=ACT_GELU_ERF, tile_hint=256, **kw))
    z0, z1 = torch.empty_like(o0), torch.empty_like(o0)
    h0 = ops.gemm(a, w, M, N, K, bias=bias, act=ACT_GELU_ERF, preact_out=z0, **kw)
    h1 = ops.gemm(a, w, M, N, K, bias=bias, act=ACT_GELU_ERF, preact_out=z1, tile_hint=256, **kw)
    assert torch.equal(h0, h1) and torch.equal(z0, z1)
    for _ in range(5):
        assert torch.equal(o0, ops.gemm(a, w, M, N, K, **kw))


def test_gemm256_row_bands_mapped_rows_and_tap(dev):
    """Bands with gathered A rows / scattered C rows (token rows around the per-frame cls slot) and the row tap of the
    spatial projection: band boundaries are logical rows, the maps apply per row as in a single launch."""
    from youku_mplug_amd import ops
    BT, N1, D, Nout = 256, 197, 256, 768
    n = N1 - 1
    rows = BT * n                                   # 50176 logical rows in a 50432-row stream
    tok = (n, N1, 1)
    x, w = rn(BT * N1, D, dev=dev, seed=85), rn(Nout, D, dev=dev, seed=86, scale=0.1)
    res = rn(BT * N1, Nout, dev=dev, seed=87)
    o0 = torch.zeros(BT * N1, Nout, dtype=torch.bfloat16, device=dev)
    o1 = torch.zeros_like(o0)
    ops.gemm(x, w, rows, Nout, D, out=o0, amap=tok, cmap=tok, residual=res)
    ops.gemm(x, w, rows, Nout, D, out=o1, amap=tok, cmap=tok, residual=res, tile_hint=256)
    assert torch.equal(o0, o1)
    assert o0[::N1].abs().max().item() == 0.0, "cls slots must be untouched"
    M = BT * N1
    t0 = torch.zeros(BT, Nout, dtype=torch.bfloat16, device=dev)
    t1 = torch.zeros_like(t0)
    p0 = ops.gemm(x, w, M, Nout, D, residual=res, row_tap_out=t0, row_tap_group=N1)
    p1 = ops.gemm(x, w, M, Nout, D, residual=res, row_tap_out=t1, row_tap_group=N1, tile_hint=256)
    assert torch.equal(p0, p1) and torch.equal(t0, t1)
    close(t0, (x.float() @ w.float().t())[::N1], 1e-2, "row tap")


@pytest.mark.parametrize("rows", [192, 160])
def test_gemm256_short_tile_rows_dgrad(dev, rows):
    """192- / 160-row tiles with the reduction-slow B operand (dgrad form), bit-identical to the 256-row tile."""
    from youku_mplug_amd import ops
    for (M, N, K) in [(160, 256, 64), (1000, 520, 192), (2560, 768, 2304)]:
        dy, w = rn(M, K, dev=dev, seed=75), rn(K, N, dev=dev, seed=76, scale=0.2)
        out = ops.gemm(dy, w, M, N, K, trans_b=True, tile_hint=rows)
        close(out, dy.float() @ w.float(), 1e-2, f"{rows}-row dgrad tile")
        assert torch.equal(out, ops.gemm(dy, w, M, N, K, trans_b=True, tile_hint=256))


@pytest.mark.parametrize("hint", [256, 128, 160])
def test_gemm_gelu_derivative_parked_by_the_forward(dev, hint):
    """preact_deriv / MPV_ACT_DERIV (include/mpv.h): the MLP's first product parks bf16(GELU'(bf16(acc + bias))) instead of the
    pre-activation, and the matching dgrad epilogue is one multiply by that tensor.  (a) the parked tensor against autograd's
    GELU' of the rounded pre-activation, both GELU kinds, every tile kernel; (b) the dgrad through it against the dgrad that
    evaluates GELU'(z) itself: they differ by ONE bf16 rounding of the derivative (<= 2^-8 relative per element)."""
    from youku_mplug_amd import ops
    M, N, K = 600, 512, 192
    a, w, bias = rn(M, K, dev=dev, seed=7), rn(N, K, dev=dev, seed=8, scale=0.1), rn(N, dev=dev, seed=9)
    dy, w2 = rn(M, K, dev=dev, seed=12), rn(K, N, dev=dev, seed=13, scale=0.1)
    for act, approx in ((ops.ACT_GELU_ERF, "none"), (ops.ACT_GELU_TANH, "tanh")):
        z = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        d = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        h0 = ops.gemm(a, w, M, N, K, bias=bias, act=act, preact_out=z, tile_hint=hint)
        h1 = ops.gemm(a, w, M, N, K, bias=bias, act=act, preact_out=d, preact_deriv=True, tile_hint=hint)
        assert torch.equal(h0, h1), "the activation output does not depend on what is parked"
        zz = z.float().requires_grad_(True)
        F.gelu(zz, approximate=approx).sum().backward()
        if ops.GELU_DERIV_FWD:
            assert (d.float() - zz.grad).abs().max().item() <= 6e-3, "parked GELU'(z): polynomial + one bf16 rounding"
            g_ref = ops.gemm(dy, w2, M, N, K, trans_b=True, act_bwd_z=z, act_bwd=act, tile_hint=hint)
            g_new = ops.gemm(dy, w2, M, N, K, trans_b=True, act_bwd_z=d, act_bwd=act, z_is_deriv=True, tile_hint=hint)
            close(g_new, (dy.float() @ w2.float()) * zz.grad, 1e-2, "dgrad x parked derivative vs fp32")
            assert rel_err(g_new, g_ref) <= 6e-3
        else:
            assert torch.equal(d, z)


def test_parked_gelu_derivative_does_not_depend_on_the_tile_kernel(dev):
    """The 256x256 kernel evaluates tanh-GELU and its derivative from ONE polynomial (mpv_gelu_tanh_both_t), the 128x128 kernel from
    the stand-alone derivative (gelu_tanh_grad_f): a product whose tiles are split between the two (row bands, fall-backs) must not
    see two functions.  The pre-activations of the two kernels are the same bf16 values (same K order per tile is NOT promised, so
    elements whose z differs are left out); on equal z the parked derivatives are the same bf16 values, bit for bit."""
    from youku_mplug_amd import ops
    M, N, K = 1024, 768, 256
    a, w, bias = rn(M, K, dev=dev, seed=17), rn(N, K, dev=dev, seed=18, scale=0.1), rn(N, dev=dev, seed=19)
    for act in (ops.ACT_GELU_ERF, ops.ACT_GELU_TANH):
        zs, ds = [], []
        for hint in (256, 128):
            z = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            d = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm(a, w, M, N, K, bias=bias, act=act, preact_out=z, tile_hint=hint)
            ops.gemm(a, w, M, N, K, bias=bias, act=act, preact_out=d, preact_deriv=True, tile_hint=hint)
            zs.append(z)
            ds.append(d.float())
        same = zs[0] == zs[1]
        assert same.float().mean().item() > 0.99
        if not ops.GELU_DERIV_FWD:
            continue
        assert torch.equal(ds[0][same], ds[1][same]), (act, (ds[0] - ds[1]).abs()[same].max().item(), (ds[0] != ds[1])[same].float().mean().item())


@pytest.mark.parametrize("M,N,K", [(2560, 2560, 10240), (1024, 2048, 8192), (512, 2560, 10240)])
def test_gemm256_split_forward_with_bias_and_dropout_in_the_reduce(dev, M, N, K):
    """Few output tiles, long reduction (the decoder's 4h -> h product at 2.7B dims: 100 tiles on 256 CUs): the forward product is split
    along K like a weight gradient and -- round 4 -- its bias / bias + dropout epilogue moves into the reduce kernel.  Against the
    unsplit 128x128 kernel: the SAME dropout mask (a function of seed / offset / element index only), values within the summation
    order; against fp32 for the bias form."""
    from youku_mplug_amd import ops
    a, w, bias = rn(M, K, dev=dev, seed=131), rn(N, K, dev=dev, seed=132, scale=0.05), rn(N, dev=dev, seed=133)
    ref = a.float() @ w.float().t() + bias.float()
    o = ops.gemm(a, w, M, N, K, bias=bias)
    close(o, ref, 1e-2, "split forward + bias")
    kw = dict(bias=bias, dropout_p=0.1, seed=77, offset=5 << 36)
    d_new = ops.gemm(a, w, M, N, K, **kw)
    d_128 = ops.gemm(a, w, M, N, K, tile_hint=128, **kw)
    assert torch.equal(d_new == 0, d_128 == 0) or ((d_new == 0) != (d_128 == 0)).float().mean().item() < 1e-5, "dropout masks differ"
    kept = d_new != 0
    assert abs(kept.float().mean().item() - 0.9) < 0.01
    close(d_new[kept], (ref / 0.9)[kept], 1e-2, "split forward + bias + dropout")
    for _ in range(5):
        assert torch.equal(d_new, ops.gemm(a, w, M, N, K, **kw))


@pytest.mark.parametrize("M,N,K", [(1576, 768, 768), (520, 1032, 2048)])
def test_gemm_bias_slice_of_any_alignment(dev, M, N, K):
    """The 256x256 kernel fetches a tile's bias slice by 16-byte LDS-DMA (csrc/gemm256.hip, round 4); a bias that is only 8-byte
    aligned -- a slice of a packed bias vector, as the packed q / k / v biases are -- is taken by the 128x128 kernel instead
    (include/mpv.h).  Both must give the fp32 answer, and a 16-byte aligned slice at a non-zero offset must be bit-identical to a
    private copy of it (ragged N: the slice's tail past column N is zero-filled by the DMA, never read from the neighbour)."""
    from youku_mplug_amd import ops
    a, w = rn(M, K, dev=dev, seed=31), rn(N, K, dev=dev, seed=32, scale=0.1)
    packed = rn(3 * N + 16, dev=dev, seed=33)
    ref = a.float() @ w.float().t()
    for off in (0, 8, 4, N + 8):                        # 16-byte aligned: 0, 8, N + 8; 8-byte aligned only: 4
        b = packed[off:off + N]
        o = ops.gemm(a, w, M, N, K, bias=b)
        close(o, ref + b.float(), 1e-2, f"bias slice at element {off}")
        if off % 8 == 0:
            assert torch.equal(o, ops.gemm(a, w, M, N, K, bias=b.clone())), off
