"""Per-kernel numerics: every HIP kernel of librlaifv_hip.so against a plain PyTorch fp32 reference of
the same op (inputs rounded to bf16 first so both sides see identical values).  Runs on the MI355X box:
    python -m pytest tests -m gpu -q
All calls go through the C ABI (rlaif_v_amd.ops -> rlaif_v_amd.hip -> ctypes)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0, dev=None):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev)


def close(got, ref, rel=1.6e-2, what=""):
    got = got.float()
    ref = ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs().max().item()
    mag = ref.abs().max().item()
    assert err <= rel * mag + 1e-6, f"{what}: max err {err:.4e} vs max |ref| {mag:.4e}"
    # mean error must be far below the max bound (catches a wrong row/column hiding under a loose max)
    merr = (got - ref).abs().mean().item()
    mmag = ref.abs().mean().item()
    assert merr <= 0.5 * rel * mmag + 1e-7, f"{what}: mean err {merr:.4e} vs mean |ref| {mmag:.4e}"


@pytest.fixture(scope="module")
def ops():
    _dev()
    from rlaif_v_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 132, 192), (1000, 520, 1024), (7, 4, 64)])
def test_gemm_nt(ops, variant, M, N, K):
    dev = _dev()
    a = rnd(M, K, seed=1, dev=dev)
    b = rnd(N, K, seed=2, dev=dev)
    out = ops.gemm_nt(a, b, variant=variant)
    close(out, a.float() @ b.float().t(), what=f"gemm v{variant} {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,lda_extra", [(4096, 64, 64, 0), (4100, 64, 128, 0), (29000, 64, 4096, 8192), (12345, 192, 1088, 64),
                                             (29000, 128, 11008, 0), (5000, 256, 192, 0)])
def test_gemm_nt_skinny(ops, M, N, K, lda_extra):
    """Many rows, <= 256 columns, plain store: rv_gemm_nt_bf16 takes the streaming kernel (gemm_nt_skinny_kernel: 3-stage ring, counted
    waits).  Against fp32 torch and, bit for bit, against the 128x128 kernel (explicit variant 1: same k order of the fp32 sums);
    activations as a column slice of a wider tensor, ragged last row tile, output into a column slice."""
    dev = _dev()
    wide = rnd(M, K + lda_extra, seed=5, dev=dev)
    a = wide[:, lda_extra // 2: lda_extra // 2 + K] if lda_extra else wide
    assert a.data_ptr() % 16 == 0
    b = rnd(N, K, seed=6, dev=dev, scale=0.1)
    out_wide = torch.full((M, N + 64), 7.0, dtype=BF, device=dev)
    got = ops.gemm_nt(a, b, out=out_wide[:, 32:32 + N], alpha=0.25)
    old = ops.gemm_nt(a, b, alpha=0.25, variant=1)
    close(got, 0.25 * (a.float() @ b.float().t()), what=f"skinny {M}x{N}x{K}")
    assert torch.equal(got, old)
    assert bool((out_wide[:, :32] == 7.0).all()) and bool((out_wide[:, 32 + N:] == 7.0).all())       # nothing outside the slice
    # identity check of the tile / lane mapping: A rows = unit vectors
    eye = torch.zeros(M, K, dtype=BF, device=dev)
    idx = torch.arange(M, device=dev) % K
    eye[torch.arange(M, device=dev), idx] = 1.0
    b2 = (torch.arange(N * K, device=dev).reshape(N, K) % 251).to(BF)
    assert torch.equal(ops.gemm_nt(eye, b2), b2.t()[idx].contiguous())


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_gemm_identity_asymmetric(ops, variant):
    """A = I with an asymmetric B catches transposed / permuted output tiles exactly."""
    dev = _dev()
    K = 256
    a = torch.eye(K, dtype=BF, device=dev)
    b = (torch.arange(384 * K, device=dev).reshape(384, K) % 251).to(BF)
    out = ops.gemm_nt(a, b, variant=variant)           # out[m][n] = b[n][m]
    assert torch.equal(out, b.t().contiguous())


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_epilogues(ops, variant, act):
    dev = _dev()
    M, N, K = 300, 256, 320
    a, b = rnd(M, K, seed=3, dev=dev, scale=0.5), rnd(N, K, seed=4, dev=dev, scale=0.5)
    bias, res = rnd(N, seed=5, dev=dev), rnd(M, N, seed=6, dev=dev)
    out = ops.gemm_nt(a, b, bias=bias, residual=res, act=act, alpha=0.5, variant=variant)
    z = 0.5 * (a.float() @ b.float().t()) + bias.float()
    if act == 1:
        z = z * torch.sigmoid(1.702 * z)
    elif act == 2:
        z = F.gelu(z)
    close(out, z + res.float(), what=f"gemm epilogue act={act}")


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_gemm_strided_views(ops, variant):
    dev = _dev()
    big_a, big_b = rnd(130, 512, seed=7, dev=dev), rnd(96, 448, seed=8, dev=dev)
    a, b = big_a[:, 128:384], big_b[:, 64:320]          # K = 256 windows, ld > K
    outbuf = torch.zeros(130, 200, dtype=BF, device=dev)
    out = outbuf[:, 8:104]
    ops.gemm_nt(a, b, out=out, variant=variant)
    close(out, a.float() @ b.float().t(), what="gemm strided")
    assert outbuf[:, :8].abs().sum() == 0 and outbuf[:, 104:].abs().sum() == 0


@pytest.mark.parametrize("v", [2, 3])
@pytest.mark.parametrize("M,N,K", [(512, 768, 64), (300, 520, 128), (700, 1000, 2048), (4096, 4096, 4096)])
def test_gemm_pingpong_large(ops, M, N, K, v):
    """256x256x32 ping-pong kernel: K = 1 / 2 / many ring tiles, ragged M and N edges, bit-stable across runs."""
    dev = _dev()
    a, b = rnd(M, K, seed=11, dev=dev, scale=0.5), rnd(N, K, seed=12, dev=dev, scale=0.5)
    out = ops.gemm_nt(a, b, variant=v)
    ref = ops.gemm_nt(a, b, variant=1)
    close(out, a.float() @ b.float().t(), what=f"gemm256 {M}x{N}x{K}")
    assert torch.equal(out, ref)                      # same accumulation order as the 128x128 kernel
    for _ in range(3):                                # race screen: repeated launches must agree exactly
        assert torch.equal(ops.gemm_nt(a, b, variant=v), out)


@pytest.mark.parametrize("R,I,J", [(32, 256, 256), (64, 512, 264), (100, 136, 520), (1000, 768, 256), (4096, 1024, 2048),
                                   (7, 8, 8)])
def test_gemm_tn(ops, R, I, J):
    """dW = dY^T X through the transposing-LDS-read kernel: ragged R (zero-row redirect), ragged I / J."""
    dev = _dev()
    p, q = rnd(R, I, seed=21, dev=dev, scale=0.5), rnd(R, J, seed=22, dev=dev, scale=0.5)
    out = ops.gemm_tn(p, q)
    close(out, p.float().t() @ q.float(), what=f"gemm_tn {R}x{I}x{J}")
    # identical to the explicit-transpose + NT path (same accumulation order when R % 32 == 0)
    ref = ops.gemm_nt(ops.transpose(p), ops.transpose(q))
    close(out, ref, rel=4e-3, what="gemm_tn vs transpose+nt")
    res = rnd(I, J, seed=23, dev=dev)
    out2 = ops.gemm_tn(p, q, residual=res, alpha=0.5)
    close(out2, 0.5 * (p.float().t() @ q.float()) + res.float(), what="gemm_tn epilogue")
    for _ in range(2):
        assert torch.equal(ops.gemm_tn(p, q), out)


def test_gemm_tn_identity_asymmetric(ops):
    dev = _dev()
    R = 256
    p = torch.eye(R, dtype=BF, device=dev)                                   # P[r][i] = delta(r, i)
    q = (torch.arange(R * 384, device=dev).reshape(R, 384) % 251).to(BF)
    assert torch.equal(ops.gemm_tn(p, q), q)                                 # out[i][j] = Q[i][j]
    assert torch.equal(ops.gemm_tn(q, p), q.t().contiguous())                # out[i][j] = Q[j][i]
    strided = rnd(300, 1024, seed=5, dev=dev)
    a, b = strided[:, 128:384], strided[:, 512:1024]
    close(ops.gemm_tn(a, b), a.float().t() @ b.float(), what="gemm_tn strided views")


@pytest.mark.parametrize("M,N,K,K2,gc", [(300, 768, 256, 64, 256), (200, 1024, 512, 64, 512), (130, 256, 128, 128, 0),
                                          (27000, 3072, 1024, 64, 1024), (26000, 1024, 2048, 192, 0)])
def test_gemm_nt_lora(ops, M, N, K, K2, gc):
    """Fused LoRA GEMM: second K segment fetched from (a2, b2), column-grouped for fused q|k|v / gate|up; both the
    128x128 kernel (small problems) and the 256x256 ping-pong kernel (>= 192 tiles)."""
    dev = _dev()
    groups = N // gc if gc else 1
    a, b = rnd(M, K, seed=31, dev=dev), rnd(N, K, seed=32, dev=dev)
    a2, b2 = rnd(M, groups * K2, seed=33, dev=dev), rnd(N, K2, seed=34, dev=dev)
    res = rnd(M, N, seed=35, dev=dev)
    ref = a.float() @ b.float().t()
    for g in range(groups):
        cols = slice(g * gc, (g + 1) * gc) if gc else slice(0, N)
        ref[:, cols] += a2[:, g * K2:(g + 1) * K2].float() @ b2[cols].float().t()
    close(ops.gemm_nt_lora(a, b, a2, b2, group_cols=gc), ref, what=f"gemm_nt_lora {M}x{N}x{K}+{K2}")
    close(ops.gemm_nt_lora(a, b, a2, b2, group_cols=gc, residual=res), ref + res.float(), what="gemm_nt_lora + residual")
    # zero adapter == plain GEMM, bit for bit (the extra K steps add exact zeros)
    z = torch.zeros_like(a2)
    assert torch.equal(ops.gemm_nt_lora(a, b, z, b2, group_cols=gc), ops.gemm_nt(a, b, variant=3 if M > 20000 else 1))


@pytest.mark.parametrize("M,N,K,K2,gc", [(27000, 3072, 1024, 64, 1024), (26000, 1024, 2048, 192, 0), (300, 512, 128, 64, 256)])
def test_gemm_nn_lora(ops, M, N, K, K2, gc):
    """NN form of the fused LoRA GEMM: bit-identical to the NT form on the transposed weight operands."""
    dev = _dev()
    groups = N // gc if gc else 1
    a, b = rnd(M, K, seed=61, dev=dev), rnd(N, K, seed=62, dev=dev)
    a2, b2 = rnd(M, groups * K2, seed=63, dev=dev), rnd(N, K2, seed=64, dev=dev)
    res = rnd(M, N, seed=65, dev=dev)
    ref = ops.gemm_nt_lora(a, b, a2, b2, group_cols=gc, residual=res)
    got = ops.gemm_nn_lora(a, b.t().contiguous(), a2, b2.t().contiguous(), group_cols=gc, residual=res)
    if M > 20000:
        assert torch.equal(got, ref)
    else:
        close(got, ref, rel=4e-3, what="gemm_nn_lora vs nt_lora (different tile kernels)")
    assert torch.equal(ops.linear_lora(a, b, b.t().contiguous(), a2, b2, b2.t().contiguous(), group_cols=gc, residual=res),
                       got if M > 20000 and gc % 256 == 0 else ref)


@pytest.mark.parametrize("R,I,J", [(5000, 64, 1024), (27664, 192, 4096), (3000, 1024, 64), (777, 64, 64)])
def test_gemm_tn_skinny(ops, R, I, J):
    """Split-K TN GEMM (LoRA weight gradients): ragged last chunk, deterministic fp32 second pass."""
    dev = _dev()
    p, q = rnd(R, I, seed=41, dev=dev, scale=0.3), rnd(R, J, seed=42, dev=dev, scale=0.3)
    out = ops.gemm_tn_skinny(p, q, alpha=0.25)
    close(out, 0.25 * (p.float().t() @ q.float()), what=f"gemm_tn_skinny {R}x{I}x{J}")
    assert torch.equal(ops.gemm_tn_skinny(p, q, alpha=0.25), out)
    close(ops.gemm_tn_skinny(p, q, splits=1), ops.gemm_tn(p, q), rel=4e-3, what="splits=1 vs gemm_tn")
    view = torch.zeros(I, J + 64, dtype=BF, device=dev)
    ops.gemm_tn_skinny(p, q, out=view[:, 32:32 + J], splits=7)
    close(view[:, 32:32 + J], p.float().t() @ q.float(), what="gemm_tn_skinny strided out")
    assert view[:, :32].abs().sum() == 0 and view[:, 32 + J:].abs().sum() == 0


@pytest.mark.parametrize("M,N,K", [(300, 512, 64), (1000, 1288, 160), (27000, 4096, 1024), (26000, 3072, 4096)])
def test_gemm_nn(ops, M, N, K):
    """NN GEMM (weight operand as [K][N], fetched in full 512-byte segments, transposing LDS reads): ragged M / N,
    strided views, residual + alpha; bit-identical to the NT kernel on the transposed operand (same MFMA sequence)."""
    dev = _dev()
    a, b = rnd(M, K, seed=51, dev=dev), rnd(K, N, seed=52, dev=dev)
    out = ops.gemm_nn(a, b)
    close(out, a.float() @ b.float(), what=f"gemm_nn {M}x{N}x{K}")
    bt = b.t().contiguous()
    if K % 64 == 0:
        assert torch.equal(out, ops.gemm_nt(a, bt, variant=3))
    res = rnd(M, N, seed=53, dev=dev)
    close(ops.gemm_nn(a, b, residual=res, alpha=0.5), 0.5 * (a.float() @ b.float()) + res.float(), what="gemm_nn epilogue")
    wide = rnd(K, N + 128, seed=54, dev=dev)
    close(ops.gemm_nn(a, wide[:, 64:64 + N]), a.float() @ wide[:, 64:64 + N].float(), what="gemm_nn strided b")
    assert torch.equal(ops.gemm_nn(a, b), out)


def test_gemm_mfma_shape_variants_agree(ops):
    """The 16x16x32-MFMA main loops (default) and the 32x32x16 loops (rv_set_gemm_mi16(0)) of the NN-A64 and TN kernels
    compute the same products: both against fp32 torch, and against each other to bf16 rounding of the output."""
    from rlaif_v_amd import hip
    dev = _dev()
    a, b = rnd(3000, 1024, seed=61, dev=dev), rnd(1024, 1536, seed=62, dev=dev)
    p, q = rnd(5000, 768, seed=63, dev=dev), rnd(5000, 1280, seed=64, dev=dev)
    outs = {}
    try:
        for mode in (1, 0):
            hip.lib().call("rv_set_gemm_mi16", mode)
            outs[mode] = (ops.gemm_nn(a, b), ops.gemm_tn(p, q))
    finally:
        hip.lib().call("rv_set_gemm_mi16", 1)
    for mode in (1, 0):
        close(outs[mode][0], a.float() @ b.float(), what=f"gemm_nn mi16={mode}")
        close(outs[mode][1], p.float().t() @ q.float(), what=f"gemm_tn mi16={mode}")
    for x, y in zip(outs[1], outs[0]):
        d = (x.float() - y.float()).abs().max().item()
        assert d <= 2.0 ** -7 * y.float().abs().max().item(), d       # one bf16 ulp of the largest magnitude


def test_gemm_f32_out(ops):
    dev = _dev()
    a, b = rnd(190, 128, seed=9, dev=dev), rnd(260, 128, seed=10, dev=dev)
    out = ops.gemm_nt_f32(a, b)
    torch.testing.assert_close(out, a.float() @ b.float().t(), rtol=1e-4, atol=1e-3)


def test_transpose(ops):
    dev = _dev()
    for R, C in [(64, 64), (100, 72), (577, 128), (1, 8), (130, 200)]:
        x = rnd(R, C, seed=R + C, dev=dev)
        t = ops.transpose(x)
        assert t.shape == (C, ops.round_up(R, 64))
        assert torch.equal(t[:, :R], x.t())
        assert t[:, R:].abs().sum() == 0


@pytest.mark.parametrize("n,n_wg", [(8, 1), (8 * 1000 + 8, 4), (1 << 22, 8), (3 * (1 << 20) + 40, 16)])
def test_reduce_copy_persistent(ops, n, n_wg):
    """The data-parallel stand-in kernel (bench.py's single-GPU probe): dst = a + b, any length, any workgroup count,
    also while another stream is busy."""
    dev = _dev()
    a, b = rnd(n, seed=n, dev=dev), rnd(n + 64, seed=n + 1, dev=dev)
    dst = torch.full((n + 8,), 7.0, dtype=BF, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        ops.reduce_copy_persistent(a, b, dst, n_wg)
    busy = ops.gemm_nt(rnd(512, 256, seed=3, dev=dev), rnd(512, 256, seed=4, dev=dev))      # main stream keeps working
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(dst[:n], (a.float() + b[:n].float()).to(BF)) and float(dst[n:].float().min()) == 7.0
    assert torch.isfinite(busy.float()).all()


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,d", [(333, 512), (1000, 1280), (5000, 4096), (3, 4096), (700, 5120)])
def test_rmsnorm_fwd_bwd(ops, rows, d):
    """Row lengths from one 2048-column pass to three (d = 5120), ragged row counts."""
    dev = _dev()
    x, w, dy, dres = rnd(rows, d, seed=1, dev=dev), (1 + 0.1 * rnd(d, seed=2, dev=dev).float()).to(BF), \
        rnd(rows, d, seed=3, dev=dev), rnd(rows, d, seed=4, dev=dev)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    ref = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5))
    close(y, ref.detach(), what="rmsnorm fwd")
    torch.testing.assert_close(rstd, torch.rsqrt(x.float().pow(2).mean(-1) + 1e-5), rtol=1e-5, atol=1e-6)
    ref.backward(dy.float())
    dw = torch.zeros(d, dtype=BF, device=dev)
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dw, dres=dres)
    close(dx, xf.grad + dres.float(), what="rmsnorm dx")
    close(dw, wf.grad, rel=2e-2, what="rmsnorm dw")
    # accumulate flag adds onto the existing dw
    dw2 = dw.clone()
    ops.rmsnorm_bwd(dy, x, w, rstd, dw2, dw_accumulate=True)
    close(dw2, 2 * wf.grad, rel=3e-2, what="rmsnorm dw accumulate")


@pytest.mark.parametrize("rows,d", [(333, 512), (1000, 4096), (5, 8192)])
def test_rmsnorm_fp32_stream_kernels(ops, rows, d):
    """The kernels of the opt-in fp32 residual stream (RV_RESID_FP32): on a bf16-representable stream they are bit-identical to the
    bf16 kernels (same arithmetic, wider load); the fused add + norm equals add then norm; row gather; backward with fp32 x."""
    dev = _dev()
    xb = rnd(rows, d, seed=1, dev=dev)
    w = rnd(d, seed=2, dev=dev, scale=0.3) + 1
    y_b, r_b = ops.rmsnorm_fwd(xb, w, 1e-5)
    y_f, r_f = ops.rmsnorm_fwd(xb.float(), w, 1e-5)
    assert torch.equal(y_b, y_f) and torch.equal(r_b, r_f)
    # fused add + norm on a genuinely fp32 stream
    x32 = torch.randn(rows, d, generator=torch.Generator().manual_seed(3)).to(dev) * 2.0
    br = rnd(rows, d, seed=4, dev=dev, scale=0.2)
    xo, y, rstd = ops.add_rmsnorm_fwd(x32, br, w, 1e-5)
    ref_sum = x32 + br.float()
    assert torch.equal(xo, ref_sum) and torch.equal(ops.add_f32_bf16(x32, br), ref_sum)
    y2, rstd2 = ops.rmsnorm_fwd(ref_sum, w, 1e-5)
    assert torch.equal(y, y2) and torch.equal(rstd, rstd2)
    xh = ref_sum * torch.rsqrt(ref_sum.pow(2).mean(-1, keepdim=True) + 1e-5)
    close(y, xh * w.float(), rel=1e-2, what="add + rmsnorm fp32 stream")
    xo3, none_y, none_r = ops.add_rmsnorm_fwd(x32, br, w, 1e-5, want_norm=False)
    assert none_y is None and none_r is None and torch.equal(xo3, ref_sum)
    # row gather (the final norm on the selected rows)
    idx = torch.randperm(rows, generator=torch.Generator().manual_seed(5))[: max(1, rows // 3)].to(torch.int32).to(dev)
    yg, rg = ops.rmsnorm_fwd(ref_sum, w, 1e-5, row_idx=idx)
    assert torch.equal(yg, y2[idx.long()]) and torch.equal(rg, rstd2[idx.long()])
    # backward: fp32 x == bf16 x when the stream is bf16-representable; and against torch on a genuinely fp32 stream
    dy, dres = rnd(rows, d, seed=6, dev=dev), rnd(rows, d, seed=7, dev=dev)
    dw_b, dw_f = torch.empty(d, dtype=BF, device=dev), torch.empty(d, dtype=BF, device=dev)
    dx_b = ops.rmsnorm_bwd(dy, xb, w, r_b, dw_b, dres=dres)
    dx_f = ops.rmsnorm_bwd(dy, xb.float(), w, r_b, dw_f, dres=dres)
    assert dx_f.dtype == BF and torch.equal(dx_b, dx_f) and torch.equal(dw_b, dw_f)
    xr = ref_sum.clone().requires_grad_(True)
    wr = w.float().clone().requires_grad_(True)
    (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * wr).backward(dy.float())
    dw = torch.empty(d, dtype=BF, device=dev)
    dx = ops.rmsnorm_bwd(dy, ref_sum, w, rstd2, dw)
    close(dx, xr.grad, rel=2e-2, what="rmsnorm bwd fp32 x")
    close(dw, wr.grad, rel=2e-2, what="rmsnorm bwd fp32 x: dw")


def test_rmsnorm_row_gather_scatter(ops):
    dev = _dev()
    rows, d = 64, 256
    x, w = rnd(rows, d, seed=1, dev=dev), rnd(d, seed=2, dev=dev)
    idx = torch.tensor([3, 60, 7, 8, 21], dtype=torch.int32, device=dev)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5, row_idx=idx)
    xs = x[idx.long()].float()
    close(y, w.float() * xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-5), what="rmsnorm gather")
    dy = rnd(5, d, seed=3, dev=dev)
    dw = torch.zeros(d, dtype=BF, device=dev)
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dw, row_idx=idx)
    xf = x.float().requires_grad_(True)
    sel = xf[idx.long()]
    (w.float() * sel * torch.rsqrt(sel.pow(2).mean(-1, keepdim=True) + 1e-5)).backward(dy.float())
    close(dx, xf.grad, what="rmsnorm scatter dx")


def test_layernorm(ops):
    dev = _dev()
    x, w, b = rnd(577, 1024, seed=1, dev=dev), rnd(1024, seed=2, dev=dev), rnd(1024, seed=3, dev=dev)
    close(ops.layernorm_fwd(x, w, b, 1e-5), F.layer_norm(x.float(), (1024,), w.float(), b.float(), 1e-5), what="ln")


def _rope_ref(x, L, H, hd, theta, inverse=False):
    n = x.shape[0]
    xf = x.float().view(n, H, hd)
    pos = torch.arange(n, device=x.device) % L
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=x.device) / hd))
    fr = pos[:, None].float() * inv[None]
    cos, sin = torch.cat([fr, fr], -1).cos()[:, None], torch.cat([fr, fr], -1).sin()[:, None]
    if inverse:
        sin = -sin
    rot = torch.cat([-xf[..., hd // 2:], xf[..., :hd // 2]], -1)
    return (xf * cos + rot * sin).view(n, H * hd)


def test_rope(ops):
    dev = _dev()
    S, L, H, hd = 3, 37, 4, 128
    buf = rnd(S * L, 3 * H * hd, seed=1, dev=dev)
    orig = buf.clone()
    cos, sin = ops.rope_tables(L, hd, 10000.0, dev)
    ops.rope_inplace(buf, cos, sin, L, 2 * H, hd)
    close(buf[:, :2 * H * hd], _rope_ref(orig[:, :2 * H * hd], L, 2 * H, hd, 10000.0), what="rope fwd")
    assert torch.equal(buf[:, 2 * H * hd:], orig[:, 2 * H * hd:])      # v untouched
    ops.rope_inplace(buf, cos, sin, L, 2 * H, hd, backward=True)
    close(buf, orig, rel=2e-2, what="rope inverse")


def test_swiglu_gelu(ops):
    dev = _dev()
    rows, f = 77, 264
    gu, da = rnd(rows, 2 * f, seed=1, dev=dev), rnd(rows, f, seed=2, dev=dev)
    guf = gu.float().requires_grad_(True)
    ref = F.silu(guf[:, :f]) * guf[:, f:]
    close(ops.swiglu_fwd(gu), ref.detach(), what="swiglu fwd")
    ref.backward(da.float())
    close(ops.swiglu_bwd(da, gu), guf.grad, what="swiglu bwd")
    x, dy = rnd(40, 256, seed=3, dev=dev), rnd(40, 256, seed=4, dev=dev)
    xf = x.float().requires_grad_(True)
    r = F.gelu(xf)
    close(ops.gelu_fwd(x), r.detach(), what="gelu fwd")
    r.backward(dy.float())
    close(ops.gelu_bwd(dy, x), xf.grad, what="gelu bwd")


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, causal):
    """q,k,v fp32 [S,H,L,hd] -> (out, lse)"""
    hd = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        L = q.shape[2]
        s = s + torch.full((L, L), float("-inf"), device=q.device).triu(1)
    return torch.softmax(s, -1) @ v, torch.logsumexp(s, -1)


@pytest.mark.parametrize("hd,causal,L", [(128, True, 200), (128, True, 64), (128, False, 130), (64, False, 577),
                                         (64, True, 33), (128, True, 1)])
def test_attn_fwd(ops, hd, causal, L):
    dev = _dev()
    S, H = 2, 3
    qkv = rnd(S * L, 3 * H * hd, seed=L + hd, dev=dev)
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, causal, 0, H * hd, 2 * H * hd)
    q, k, v = [qkv[:, i * H * hd:(i + 1) * H * hd].float().view(S, L, H, hd).transpose(1, 2) for i in range(3)]
    ro, rl = _attn_ref(q, k, v, causal)
    close(out, ro.transpose(1, 2).reshape(S * L, H * hd), rel=2e-2, what=f"attn fwd hd{hd} L{L}")
    torch.testing.assert_close(lse, rl, rtol=1e-3, atol=2e-3)


def test_attn_fwd_online_softmax_rescale(ops):
    """One key far down the sequence dominates: forces the running-max rescale branch."""
    dev = _dev()
    S, H, L, hd = 1, 1, 256, 128
    qkv = rnd(S * L, 3 * hd, seed=5, dev=dev, scale=0.3)
    qkv[200, hd:2 * hd] = 6.0 * qkv[255, :hd].sign()      # key 200 aligned with query 255
    qkv[255, :hd] = qkv[255, :hd].sign() * 1.0
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, hd, 2 * hd)
    q, k, v = [qkv[:, i * hd:(i + 1) * hd].float().view(S, L, H, hd).transpose(1, 2) for i in range(3)]
    ro, rl = _attn_ref(q, k, v, True)
    close(out, ro.transpose(1, 2).reshape(S * L, hd), rel=2e-2, what="attn rescale")
    torch.testing.assert_close(lse, rl, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("case", ["causal L1", "causal L257", "causal L640", "causal L1000", "full L577", "gqa4 causal L333", "packed L900",
                                  "packed L1300", "pad-free rows", "late dominating key", "early dominating key", "causal L2048 H32"])
def test_attn_fwd3_matches_reference(ops, case):
    """attn_fwd3_kernel (round 6: one wave per SIMD, two software-pipelined 32-query blocks per wave, O / Q / K fragments in
    hard-numbered AGPRs, lazy rescale; csrc/attn_fwd3.inc) behind the test knob rv_set_attn_fwd_version(3): same bars as version 2
    against fp32 torch attention.  It is NOT the default (profiles/r06_attn_fwd3_negative_result.log); the test keeps it correct."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.exp_attn_fwd3 import ref_attn
    from rlaif_v_amd import hip
    dev = _dev()
    cases = {"causal L1": dict(S=2, L=1, H=3, causal=True), "causal L257": dict(S=2, L=257, H=3, causal=True),
             "causal L640": dict(S=2, L=640, H=3, causal=True), "causal L1000": dict(S=2, L=1000, H=3, causal=True),
             "full L577": dict(S=1, L=577, H=2, causal=False), "gqa4 causal L333": dict(S=2, L=333, H=8, G=4, causal=True),
             "packed L900": dict(S=2, L=900, H=2, causal=True, seg=([100, 257], [500, 600])),
             "packed L1300": dict(S=1, L=1300, H=2, causal=True, seg=([64], [700])),
             "pad-free rows": dict(S=3, L=700, H=2, causal=True, rows=[700, 130, 513], seg=([40, 10, 300], [400, 60, 400])),
             "late dominating key": dict(S=1, L=512, H=1, causal=True, spike=(300, 511, 6.0)),
             "early dominating key": dict(S=1, L=512, H=1, causal=True, spike=(3, 511, 6.0)),
             "causal L2048 H32": dict(S=1, L=2048, H=32, causal=True)}
    c = cases[case]
    S, L, H, G, hd = c["S"], c["L"], c["H"], c.get("G", 1), 128
    d, dk = H * hd, (H // G) * hd
    g = torch.Generator().manual_seed(len(case) * 7 + L)
    rows, ntok = None, S * L
    if "rows" in c:
        lens = torch.tensor(c["rows"], dtype=torch.int32)
        rows = ((torch.cumsum(lens, 0, dtype=torch.int32) - lens).to(dev), lens.to(dev))
        ntok = int(lens.sum())
    qkv = (torch.randn(ntok, d + 2 * dk, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    if "spike" in c:
        kj, qi, amp = c["spike"]
        qkv[qi, :hd] = qkv[qi, :hd].sign()
        qkv[kj, d:d + hd] = amp * qkv[qi, :hd].sign()
    seg = None
    if "seg" in c:
        seg = tuple(torch.tensor(x, dtype=torch.int32, device=dev) for x in c["seg"])
    ro, rl = ref_attn(qkv, S, L, H, hd, c["causal"], G, seg, rows)
    lib = hip.lib()
    lib.call("rv_set_attn_fwd_version", 3)
    try:
        for rep in range(3 if L >= 1000 else 1):          # the kernel's first build was wrong NON-deterministically at many-tile shapes
            out, lse = ops.attn_fwd(qkv, S, L, H, hd, c["causal"], 0, d, d + dk, seg=seg, kv_group=G, rows=rows)
            torch.cuda.synchronize()
            close(out, ro, rel=2e-2, what=f"attn fwd3 {case}")
            for s_ in range(S):
                n = int(rows[1][s_]) if rows is not None else L
                torch.testing.assert_close(lse[s_, :, :n], rl[s_, :, :n], rtol=1e-3, atol=3e-3)
    finally:
        lib.call("rv_set_attn_fwd_version", 0)


@pytest.mark.parametrize("causal,L", [(True, 200), (True, 128), (False, 97), (True, 1)])
def test_attn_bwd(ops, causal, L):
    dev = _dev()
    S, H, hd = 2, 2, 128
    qkv = rnd(S * L, 3 * H * hd, seed=L, dev=dev, scale=0.7)
    do = rnd(S * L, H * hd, seed=L + 1, dev=dev)
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, causal, 0, H * hd, 2 * H * hd)
    dqkv = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, causal, 0, H * hd, 2 * H * hd)
    qf = qkv.float().requires_grad_(True)
    q, k, v = [qf[:, i * H * hd:(i + 1) * H * hd].view(S, L, H, hd).transpose(1, 2) for i in range(3)]
    ro, _ = _attn_ref(q, k, v, causal)
    ro.transpose(1, 2).reshape(S * L, H * hd).backward(do.float())
    for i, nm in enumerate("qkv"):
        sl = slice(i * H * hd, (i + 1) * H * hd)
        close(dqkv[:, sl], qf.grad[:, sl], rel=2.5e-2, what=f"attn bwd d{nm} L{L} causal={causal}")


@pytest.mark.parametrize("H,G,causal,L", [(4, 2, True, 200), (8, 4, True, 130), (4, 4, False, 97), (6, 3, True, 257)])
def test_attn_gqa_fwd_bwd(ops, H, G, causal, L):
    """Grouped-query attention (HF repeat_kv): H query heads on H/G key/value heads; dK/dV sum over the group."""
    dev = _dev()
    S, hd = 2, 128
    Hkv = H // G
    width = (H + 2 * Hkv) * hd
    qkv = rnd(S * L, width, seed=L + H, dev=dev, scale=0.7)
    do = rnd(S * L, H * hd, seed=L + 1, dev=dev)
    kc, vc = H * hd, (H + Hkv) * hd
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, causal, 0, kc, vc, kv_group=G)
    dqkv = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, causal, 0, kc, vc, kv_group=G)
    qf = qkv.float().requires_grad_(True)
    q = qf[:, :kc].view(S, L, H, hd).transpose(1, 2)
    k = qf[:, kc:vc].view(S, L, Hkv, hd).transpose(1, 2).repeat_interleave(G, dim=1)
    v = qf[:, vc:].view(S, L, Hkv, hd).transpose(1, 2).repeat_interleave(G, dim=1)
    ro, rl = _attn_ref(q, k, v, causal)
    close(out, ro.transpose(1, 2).reshape(S * L, H * hd), rel=2e-2, what=f"gqa fwd H{H} G{G}")
    torch.testing.assert_close(lse, rl.detach(), rtol=1e-3, atol=2e-3)
    ro.transpose(1, 2).reshape(S * L, H * hd).backward(do.float())
    for nm, sl in (("q", slice(0, kc)), ("k", slice(kc, vc)), ("v", slice(vc, width))):
        close(dqkv[:, sl], qf.grad[:, sl], rel=2.5e-2, what=f"gqa bwd d{nm} H{H} G{G} L{L}")
    assert torch.equal(ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, causal, 0, kc, vc, kv_group=G), dqkv)


@pytest.mark.parametrize("G,use_pos", [(1, False), (1, True), (2, True)])
def test_attn_bwd_fused_inverse_rope(ops, G, use_pos):
    """rv_attn_bwd with rope tables writes dQ / dK already rotated back: identical (to bf16 rounding of the fp32 rotation)
    to the unfused rv_attn_bwd followed by rv_rope_inplace(backward); dV untouched.  Packed rows + position table + GQA."""
    dev = _dev()
    S, L, H, hd = 2, 333, 4, 128
    Hkv = H // G
    width = (H + 2 * Hkv) * hd
    kc, vc = H * hd, (H + Hkv) * hd
    qkv = rnd(S * L, width, seed=7, dev=dev, scale=0.7)
    do = rnd(S * L, H * hd, seed=8, dev=dev)
    sh = torch.tensor([130, 0], dtype=torch.int32, device=dev)
    e1 = torch.tensor([260, 100], dtype=torch.int32, device=dev)
    cos, sin = ops.rope_tables(512, hd, 10000.0, dev)
    pos = (torch.randint(0, 512, (S * L,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev)) if use_pos else None
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, kc, vc, seg=(sh, e1), kv_group=G)
    # (RV_ATTN_DKV is read once per process: the round-2/3 dK/dV kernel is covered when the whole file runs with RV_ATTN_DKV=3)
    ref = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, True, 0, kc, vc, seg=(sh, e1), kv_group=G)
    ops.rope_inplace(ref, cos, sin, L, H + Hkv, hd, backward=True, pos=pos)
    got = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, True, 0, kc, vc, seg=(sh, e1), kv_group=G, rope=(cos, sin, pos))
    assert torch.equal(got[:, vc:], ref[:, vc:])
    d = (got[:, :vc].float() - ref[:, :vc].float()).abs().max().item()
    mag = ref[:, :vc].float().abs().max().item()
    print(f"fused inverse rope G={G} pos={use_pos}: max |diff| {d:.3e} of {mag:.3e}")
    assert d <= 2.0 ** -7 * mag                      # one bf16 ulp of the largest magnitude (fp32 contraction order)


def _packed_mask(L, sh, e1, dev):
    i = torch.arange(L, device=dev)[:, None]
    j = torch.arange(L, device=dev)[None, :]
    ok = (j <= i) & ~((i >= e1) & (j >= sh) & (j < e1))
    return torch.where(ok, 0.0, float("-inf"))


@pytest.mark.parametrize("L,segs", [(200, [(40, 120), (0, 64)]), (333, [(130, 260), (129, 131)]), (64, [(10, 30), (63, 64)]),
                                    (300, [(0, 40), (0, 100)]),
                                    # whole 128-query blocks in the rejected branch / whole key blocks in the chosen branch:
                                    # the block-uniform tile skipping (incl. a skip range that starts at tile 0)
                                    (900, [(100, 500), (64, 448), (0, 256), (130, 131)]), (1100, [(638, 900), (640, 1024)])])
def test_attn_packed_pairs_fwd_bwd(ops, L, segs):
    """[shared | chosen | rejected] rows: rejected-branch queries must not see chosen-branch keys."""
    dev = _dev()
    S, H, hd = len(segs), 2, 128
    qkv = rnd(S * L, 3 * H * hd, seed=L, dev=dev, scale=0.7)
    do = rnd(S * L, H * hd, seed=L + 1, dev=dev)
    sh = torch.tensor([a for a, _ in segs], dtype=torch.int32, device=dev)
    e1 = torch.tensor([b for _, b in segs], dtype=torch.int32, device=dev)
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, H * hd, 2 * H * hd, seg=(sh, e1))
    dqkv = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, True, 0, H * hd, 2 * H * hd, seg=(sh, e1))
    qf = qkv.float().requires_grad_(True)
    q, k, v = [qf[:, i * H * hd:(i + 1) * H * hd].view(S, L, H, hd).transpose(1, 2) for i in range(3)]
    mask = torch.stack([_packed_mask(L, a, b, dev) for a, b in segs])[:, None]          # [S,1,L,L]
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask
    ro = torch.softmax(sc, -1) @ v
    close(out, ro.detach().transpose(1, 2).reshape(S * L, H * hd), rel=2e-2, what="packed attn fwd")
    torch.testing.assert_close(lse, torch.logsumexp(sc, -1).detach(), rtol=1e-3, atol=2e-3)
    ro.transpose(1, 2).reshape(S * L, H * hd).backward(do.float())
    for i, nm in enumerate("qkv"):
        sl = slice(i * H * hd, (i + 1) * H * hd)
        close(dqkv[:, sl], qf.grad[:, sl], rel=2.5e-2, what=f"packed attn bwd d{nm}")


@pytest.mark.parametrize("name,L,G,segs", [("plain causal L=4096", 4096, 1, None),
                                            # BASELINE config 5's packed rows: [shared 639 | chosen to 4096 | rejected tail], 7,396 and 5,239 tokens
                                            ("packed L=7396", 7396, 1, [(639, 4096), (639, 3739)]),
                                            ("gqa G=4 packed L=4500", 4500, 4, [(210, 2048), (146, 3000)]),
                                            ("gqa G=4 plain L=4096", 4096, 4, None)])
def test_attn_long_rows_fwd_bwd(ops, name, L, G, segs):
    """The attention kernels at config 5's length (VERDICT r4 missing 1: no attention test ran a row longer than 3,458 tokens):
    64 key tiles per chosen branch, packed rows past 7,000 tokens, kv_group 4 (config 4's head arrangement) - forward output,
    lse and dQ / dK / dV against fp32 torch attention on the same bf16-rounded inputs (the reference evaluated per head in
    query chunks: an [L, L] fp32 score matrix per head at a time)."""
    dev = _dev()
    S, H, hd = 2, 4, 128
    Hkv = H // G
    width, kc, vc = (H + 2 * Hkv) * hd, H * hd, (H + Hkv) * hd
    qkv = rnd(S * L, width, seed=L + G, dev=dev, scale=0.7)
    do = rnd(S * L, H * hd, seed=L + 1, dev=dev)
    seg = None
    if segs is not None:
        seg = (torch.tensor([a for a, _ in segs], dtype=torch.int32, device=dev), torch.tensor([b for _, b in segs], dtype=torch.int32, device=dev))
    out, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, kc, vc, seg=seg, kv_group=G)
    dqkv = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, True, 0, kc, vc, seg=seg, kv_group=G)
    assert torch.equal(ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, True, 0, kc, vc, seg=seg, kv_group=G), dqkv)     # deterministic
    qf = qkv.float().requires_grad_(True)
    ro_rows, lse_rows = [], []
    for s_ in range(S):
        rows = slice(s_ * L, (s_ + 1) * L)
        mask = _packed_mask(L, segs[s_][0], segs[s_][1], dev) if segs is not None else _packed_mask(L, L, L, dev)
        heads_o, heads_l = [], []
        for h in range(H):
            q = qf[rows, h * hd:(h + 1) * hd]
            k = qf[rows, kc + (h // G) * hd:kc + (h // G + 1) * hd]
            v = qf[rows, vc + (h // G) * hd:vc + (h // G + 1) * hd]
            sc = (q @ k.t()) / math.sqrt(hd) + mask
            o = torch.softmax(sc, -1) @ v
            (o * do[rows, h * hd:(h + 1) * hd].float()).sum().backward()
            heads_o.append(o.detach())
            heads_l.append(torch.logsumexp(sc.detach(), -1))
            del sc, o
        ro_rows.append(torch.cat(heads_o, 1))
        lse_rows.append(torch.stack(heads_l))
    close(out, torch.cat(ro_rows, 0), rel=2e-2, what=f"{name}: fwd")
    torch.testing.assert_close(lse, torch.stack(lse_rows), rtol=1e-3, atol=2e-3)
    for nm, sl in (("q", slice(0, kc)), ("k", slice(kc, vc)), ("v", slice(vc, width))):
        close(dqkv[:, sl], qf.grad[:, sl], rel=2.5e-2, what=f"{name}: d{nm}")


@pytest.mark.parametrize("G", [1, 2])
def test_attn_pad_free_rows_bit_identical_to_rectangular(ops, G):
    """rv_attn_fwd / rv_attn_bwd with (row_off, row_len): every row of a CONCATENATED buffer gives bit-identical out / lse / dQ /
    dK / dV to the same row evaluated alone as a rectangular S = 1 launch (tiles are laid relative to the row's first token, so a
    row's arithmetic does not depend on where it starts; blocks past a row's end leave at once).  Packed segments, fused inverse
    RoPE through the position table, grouped-query attention; row lengths from 1 tile to the launch maximum."""
    dev = _dev()
    H, hd = 4, 128
    Hkv = H // G
    width, kc, vc = (H + 2 * Hkv) * hd, H * hd, (H + Hkv) * hd
    lens = [1100, 64, 333, 897, 130]
    segs = [(638, 900), (10, 30), (129, 131), (0, 256), (130, 130)]
    off = [sum(lens[:i]) for i in range(len(lens))]
    N, S, Lmax = sum(lens), len(lens), max(lens)
    qkv = rnd(N, width, seed=3, dev=dev, scale=0.7)
    do = rnd(N, H * hd, seed=4, dev=dev)
    cos, sin = ops.rope_tables(2048, hd, 10000.0, dev)
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32).to(dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)                       # noqa: E731
    rows = (i32(off), i32(lens))
    seg = (i32([a for a, _ in segs]), i32([b for _, b in segs]))
    out, lse = ops.attn_fwd(qkv, S, Lmax, H, hd, True, 0, kc, vc, seg=seg, kv_group=G, rows=rows)
    dqkv = ops.attn_bwd(qkv, out, do, lse, S, Lmax, H, hd, True, 0, kc, vc, seg=seg, kv_group=G, rope=(cos, sin, pos), rows=rows)
    assert out.shape == (N, H * hd)
    for s_, (a, n) in enumerate(zip(off, lens)):
        sl = slice(a, a + n)
        seg1 = (seg[0][s_:s_ + 1], seg[1][s_:s_ + 1])
        o1, l1 = ops.attn_fwd(qkv[sl], 1, n, H, hd, True, 0, kc, vc, seg=seg1, kv_group=G)
        d1 = ops.attn_bwd(qkv[sl], o1, do[sl], l1, 1, n, H, hd, True, 0, kc, vc, seg=seg1, kv_group=G, rope=(cos, sin, pos[sl].contiguous()))
        assert torch.equal(out[sl], o1), f"row {s_}: forward output"
        assert torch.equal(lse[s_, :, :n], l1[0]), f"row {s_}: lse"
        assert torch.equal(dqkv[sl], d1), f"row {s_}: dqkv"
    assert torch.equal(ops.attn_bwd(qkv, out, do, lse, S, Lmax, H, hd, True, 0, kc, vc, seg=seg, kv_group=G, rope=(cos, sin, pos), rows=rows), dqkv)


def test_rope_position_table(ops):
    dev = _dev()
    n, H, hd, L = 150, 2, 128, 64
    buf = rnd(n, 2 * H * hd, seed=2, dev=dev)
    orig = buf.clone()
    pos = torch.randint(0, L, (n,), generator=torch.Generator().manual_seed(0)).to(torch.int32).to(dev)
    cos, sin = ops.rope_tables(L, hd, 10000.0, dev)
    ops.rope_inplace(buf, cos, sin, L, 2 * H, hd, pos=pos)
    xf = orig.float().view(n, 2 * H, hd)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32, device=dev) / hd))
    fr = pos[:, None].float() * inv[None]
    c, s_ = torch.cat([fr, fr], -1).cos()[:, None], torch.cat([fr, fr], -1).sin()[:, None]
    rot = torch.cat([-xf[..., hd // 2:], xf[..., :hd // 2]], -1)
    close(buf, (xf * c + rot * s_).view(n, 2 * H * hd), what="rope pos table")


# ------------------------------------------------------------------------------------------- LM head / loss
def test_lmhead_logp_fwd_bwd(ops):
    dev = _dev()
    n, d, V = 150, 256, 512
    npad = ops.round_up(n, 64)
    h = torch.zeros(npad, d, dtype=BF, device=dev)
    h[:n] = rnd(n, d, seed=1, dev=dev)
    w = rnd(V, d, seed=2, dev=dev, scale=0.2)
    tgt = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(dev)
    tgt[0], tgt[1] = 0, V - 1
    logp, lse = ops.lmhead_logp_fwd(h, w, tgt, n)
    hf = h[:n].float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    logits = hf @ wf.t()
    ref_lp = logits.log_softmax(-1).gather(1, tgt.long()[:, None])[:, 0]
    torch.testing.assert_close(lse, torch.logsumexp(logits, -1).detach(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(logp, ref_lp.detach(), rtol=1e-4, atol=1e-3)
    coef = torch.randn(n, generator=torch.Generator().manual_seed(4)).to(dev)
    dlog = ops.lmhead_logp_bwd(h, w, tgt, lse, coef, n)
    (ref_lp * coef).sum().backward()
    p = torch.softmax(logits.detach(), -1)
    onehot = F.one_hot(tgt.long(), V).float()
    close(dlog[:n], coef[:, None] * (onehot - p), rel=1e-2, what="dlogits")
    assert dlog[n:].abs().sum() == 0
    # the two GEMMs the model runs on dlogits reproduce autograd's dh and dW
    dh = ops.gemm_nt(dlog, ops.transpose(w)[:, :V])           # [npad, d] = dlog @ w
    close(dh[:n], hf.grad, rel=2e-2, what="lm head dh")
    dW = ops.gemm_nt(ops.transpose(dlog), ops.transpose(h))   # [V, d] = dlog^T @ h
    close(dW, wf.grad, rel=2e-2, what="lm head dW")


@pytest.mark.parametrize("v_valid", [449, 457, 511])
def test_lmhead_logp_padded_vocabulary(ops, v_valid):
    """A tokenizer with added tokens (OmniLMM: 32000 + 9): the stored head has zero rows up to a multiple of 64; those columns
    must not enter the softmax (forward) and get zero dlogits (backward) - compared with the UNPADDED fp32 computation."""
    dev = _dev()
    n, d, V = 100, 256, 512
    npad = ops.round_up(n, 64)
    h = torch.zeros(npad, d, dtype=BF, device=dev)
    h[:n] = rnd(n, d, seed=1, dev=dev)
    w = rnd(V, d, seed=2, dev=dev, scale=0.2)
    w[v_valid:] = 0
    tgt = torch.randint(0, v_valid, (n,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(dev)
    tgt[0], tgt[1] = 0, v_valid - 1
    logp, lse = ops.lmhead_logp_fwd(h, w, tgt, n, v_valid=v_valid)
    logits = h[:n].float() @ w[:v_valid].float().t()
    ref_lp = logits.log_softmax(-1).gather(1, tgt.long()[:, None])[:, 0]
    torch.testing.assert_close(lse, torch.logsumexp(logits, -1), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(logp, ref_lp, rtol=1e-4, atol=1e-3)
    coef = torch.randn(n, generator=torch.Generator().manual_seed(4)).to(dev)
    dlog = ops.lmhead_logp_bwd(h, w, tgt, lse, coef, n, v_valid=v_valid)
    p = torch.softmax(logits, -1)
    close(dlog[:n, :v_valid], coef[:, None] * (F.one_hot(tgt.long(), v_valid).float() - p), rel=1e-2, what="dlogits (valid columns)")
    assert dlog[:, v_valid:].abs().sum() == 0 and dlog[n:].abs().sum() == 0


def test_seq_sum_and_dpo_loss(ops):
    dev = _dev()
    from oracle import dpo_oracle as O
    B = 5
    g = torch.Generator().manual_seed(0)
    lens = [7, 1, 12, 3, 9, 4, 0, 6, 2, 11]
    import itertools
    off = torch.tensor([0] + list(itertools.accumulate(lens)), dtype=torch.int32, device=dev)
    logp = (-torch.rand(sum(lens), generator=g) * 5).to(dev)
    s, c = ops.seq_sum(logp, off, 2 * B)
    ref_s = torch.stack([logp[off[i]:off[i + 1]].sum() for i in range(2 * B)])
    torch.testing.assert_close(s, ref_s, rtol=1e-6, atol=1e-5)
    assert c.tolist() == [float(x) for x in lens]
    ref_win = (-20 * torch.rand(B, generator=g)).to(dev)
    ref_rej = (-20 * torch.rand(B, generator=g)).to(dev)
    for use_avg, sft, dpo in [(False, 0.0, 1.0), (False, 0.3, 0.7)]:
        per_pair, scal, coef = ops.dpo_loss(s, c, ref_win, ref_rej, 0.1, use_avg, sft, dpo)
        sv = s.clone().requires_grad_(True)
        pw, pr = sv[:B], sv[B:]
        losses, cw, cr = O.dpo_loss(pw, pr, ref_win, ref_rej, 0.1)
        loss = dpo * losses.mean() - sft * pw.mean()
        loss.backward()
        torch.testing.assert_close(per_pair[0], losses.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(per_pair[1], cw, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(per_pair[2], cr, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(scal[0], loss.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(scal[3], (cw > cr).float().mean(), rtol=0, atol=1e-6)
        torch.testing.assert_close(coef, sv.grad, rtol=1e-4, atol=1e-7)
    # average log-prob variant (dpo_use_average): row with zero targets gives NaN like the reference
    cnt_ok = c.clone()
    cnt_ok[6] = 1.0
    per_pair, scal, coef = ops.dpo_loss(s, cnt_ok, ref_win, ref_rej, 0.1, True, 0.0, 1.0)
    sv = s.clone().requires_grad_(True)
    avg = sv / cnt_ok
    losses, _, _ = O.dpo_loss(avg[:B], avg[B:], ref_win, ref_rej, 0.1)
    losses.mean().backward()
    torch.testing.assert_close(per_pair[0], losses.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(coef, sv.grad, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------- data movement
def test_splice_and_grads(ops):
    dev = _dev()
    d, V, nfeat = 256, 50, 12
    embed, feats = rnd(V, d, seed=1, dev=dev), rnd(nfeat, d, seed=2, dev=dev)
    src = torch.tensor([3, 3, -2, -3, -13, -1, 49, 0, -1, 3], dtype=torch.int32, device=dev)
    out = ops.splice_fwd(src, embed, feats, d)
    for i, sidx in enumerate(src.tolist()):
        exp = embed[sidx] if sidx >= 0 else (torch.zeros(d, dtype=BF, device=dev) if sidx == -1 else feats[-2 - sidx])
        assert torch.equal(out[i], exp)
    dx = rnd(10, d, seed=3, dev=dev)
    # embedding backward: rows {3: [0,1,9], 49: [6], 0: [7]}
    uniq = torch.tensor([0, 3, 49], dtype=torch.int32, device=dev)
    seg = torch.tensor([0, 1, 4, 5], dtype=torch.int32, device=dev)
    pos = torch.tensor([7, 0, 1, 9, 6], dtype=torch.int32, device=dev)
    dW = torch.zeros(V, d, dtype=BF, device=dev)
    ops.embed_bwd(uniq, seg, pos, dx, dW)
    ref = torch.zeros(V, d, device=dev)
    ref.index_add_(0, torch.tensor([3, 3, 49, 0, 3], device=dev), dx[[0, 1, 6, 7, 9]].float())
    close(dW, ref, rel=1e-2, what="embed bwd")
    a = torch.tensor([2, -1, 4], dtype=torch.int32, device=dev)
    b = torch.tensor([3, -1, -1], dtype=torch.int32, device=dev)
    fg = ops.feat_grad(a, b, dx, d)
    close(fg[0], dx[2].float() + dx[3].float(), rel=1e-2, what="feat grad")
    assert fg[1].abs().sum() == 0 and torch.equal(fg[2], dx[4])
    idx = torch.tensor([9, 0, 4], dtype=torch.int32, device=dev)
    gth = ops.gather_rows(dx, idx)
    assert torch.equal(gth, dx[idx.long()])
    sc = torch.zeros(10, d, dtype=BF, device=dev)
    ops.scatter_rows(gth, idx, sc)
    assert torch.equal(sc[idx.long()], gth)
    mask = torch.ones(10, dtype=torch.bool, device=dev)
    mask[idx.long()] = False
    assert sc[mask].abs().sum() == 0
    close(ops.colsum(dx), dx.float().sum(0), rel=1e-2, what="colsum")


def test_clip_front_end(ops):
    dev = _dev()
    B, HW, ps, cd = 2, 56, 14, 128
    px = torch.randn(B, 3, HW, HW, generator=torch.Generator().manual_seed(0)).to(dev)
    Kp = ops.round_up(3 * ps * ps, 64)
    cols = ops.im2col_patches(px, ps, Kp)
    ref = F.unfold(px.to(BF).float(), ps, stride=ps).transpose(1, 2).reshape(B * 16, 3 * ps * ps)
    assert torch.equal(cols[:, :588].float(), ref) and cols[:, 588:].abs().sum() == 0
    wconv = rnd(cd, 3, ps, ps, seed=1, dev=dev, scale=0.05)
    wpad = torch.zeros(cd, Kp, dtype=BF, device=dev)
    wpad[:, :588] = wconv.reshape(cd, 588)
    pe = ops.gemm_nt(cols, wpad)
    conv = F.conv2d(px.to(BF).float(), wconv.float(), stride=ps).flatten(2).transpose(1, 2).reshape(B * 16, cd)
    close(pe, conv, what="patch embed")
    cls, pos = rnd(cd, seed=2, dev=dev), rnd(17, cd, seed=3, dev=dev)
    x = ops.clip_assemble(pe, cls, pos, B, 16)
    refx = torch.cat([cls.float().expand(B, 1, cd), pe.float().view(B, 16, cd)], 1) + pos.float()[None]
    close(x, refx.reshape(B * 17, cd), rel=1e-2, what="clip assemble")


@pytest.mark.parametrize("M,N,K", [(1154, 1024, 1024), (1154, 1024, 4096), (300, 256, 128), (4617, 1024, 1024)])
def test_gemm_nt_f32_residual_and_layernorm_f32in(ops, M, N, K):
    """The two kernels of the CLIP tower's fp32 residual stream (RV_CLIP_FP32_RESID): x32 += a @ w^T + bias in place
    (rv_gemm_nt_bf16_f32res, residual aliasing the output, every GEMM variant the dispatcher may pick) and LayerNorm of an fp32
    input with bf16 output (rv_layernorm_fwd_f32in), against fp32 torch."""
    dev = _dev()
    a, w, bias = rnd(M, K, seed=1, dev=dev, scale=0.5), rnd(N, K, seed=2, dev=dev, scale=0.05), rnd(N, seed=3, dev=dev)
    x32 = torch.randn(M, N, generator=torch.Generator().manual_seed(4)).to(dev) * 3.0
    ref = x32 + a.float() @ w.float().t() + bias.float()
    for variant in (-1, 0, 1, 2):
        got = ops.gemm_nt_f32res(a, w, bias, x32.clone(), variant=variant)
        assert got.dtype == torch.float32
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-3)
    got = ops.gemm_nt_f32res(a, w, None, x32.clone())
    torch.testing.assert_close(got, ref - bias.float(), rtol=1e-4, atol=2e-3)
    g, b = rnd(N, seed=5, dev=dev, scale=0.3) + 1, rnd(N, seed=6, dev=dev, scale=0.1)
    y = ops.layernorm_fwd_f32in(ref, g, b, 1e-5)
    yr = F.layer_norm(ref, (N,), g.float(), b.float(), 1e-5)
    close(y, yr, rel=1e-2, what="layernorm fp32 in")
    # and it is the bf16-input kernel's arithmetic: identical output on an input that IS bf16-representable
    xb = ref.to(BF)
    assert torch.equal(ops.layernorm_fwd_f32in(xb.float(), g, b, 1e-5), ops.layernorm_fwd(xb, g, b, 1e-5))


# ------------------------------------------------------------------------------------------- optimizer
def test_adamw_and_gradnorm(ops):
    dev = _dev()
    n = 8 * 1000
    g = torch.Generator().manual_seed(0)
    master = torch.randn(n, generator=g).to(dev)
    p = master.to(BF)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    ref = torch.nn.Parameter(master.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for step in range(1, 4):
        grad = (torch.randn(n, generator=g) * 3).to(BF).to(dev)
        out2 = ops.grad_norm(grad, 1.0)
        nrm = grad.float().norm()
        torch.testing.assert_close(out2[0], nrm, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out2[1], torch.clamp(1.0 / (nrm + 1e-6), max=1.0), rtol=1e-4, atol=1e-6)
        ops.adamw_step(p, master, m, v, grad, 1e-2, 0.9, 0.999, 1e-8, 0.01, step, clip=out2)
        ref.grad = grad.float().clone()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        torch.testing.assert_close(master, ref.detach(), rtol=2e-5, atol=2e-6)
        assert torch.equal(p, master.to(BF))


@pytest.mark.parametrize("M,d,f", [(300, 256, 512), (1000, 512, 776), (27000, 1024, 1024)])
def test_swiglu_fused_gemm_epilogues(ops, M, d, f):
    """SwiGLU in the gate|up GEMM epilogue and its backward in the down-projection input-gradient epilogue (interleaved gate /
    up columns).  Forward (round 6): the activation comes from the GEMM's fp32 ACCUMULATORS, so it is no longer bit-identical to the
    unfused composition GEMM -> bf16 -> swiglu kernel - it must be CLOSER to the fp32 product silu(x Wg) * (x Wu) than that
    composition is (the kept gate|up tile still is the rounded GEMM output, bit for bit).  Backward: bit-identical to the unfused
    composition, which is pinned against torch fp32."""
    dev = _dev()
    x = rnd(M, d, seed=71, dev=dev, scale=1.0)
    wguT = rnd(d, 2 * f, seed=72, dev=dev, scale=0.08)          # [in, 2f], column 2j = gate_j, 2j+1 = up_j
    gu, act = ops.linear_swiglu(x, wguT)
    gu_ref = ops.gemm_nn(x, wguT)
    assert torch.equal(gu, gu_ref)
    act_unfused = ops.swiglu_fwd(gu_ref, interleaved=True)
    g32, u32 = gu_ref.float()[:, 0::2], gu_ref.float()[:, 1::2]
    close(act, torch.nn.functional.silu(g32) * u32, what="fused swiglu act")
    # against the fp32 product of the same bf16 operands: the fused activation (one rounding) beats the unfused one (three)
    gu_f = x.double() @ wguT.double()
    exact = torch.nn.functional.silu(gu_f[:, 0::2]) * gu_f[:, 1::2]
    e_fused, e_unfused = (act.double() - exact).abs().mean().item(), (act_unfused.double() - exact).abs().mean().item()
    assert e_fused < 0.75 * e_unfused, (e_fused, e_unfused)
    close(act, exact.float(), rel=6e-3, what="fused swiglu act vs exact")
    # the interleaved kernels agree with the block-layout ones on the de-interleaved tensor
    blocks = torch.cat([gu_ref[:, 0::2], gu_ref[:, 1::2]], 1).contiguous()
    assert torch.equal(ops.swiglu_fwd(blocks), act_unfused)
    # backward: dgu = SwiGLU'(gu) o (dy @ W_down)
    dy = rnd(M, d, seed=73, dev=dev, scale=0.5)
    w_down = rnd(d, f, seed=74, dev=dev, scale=0.08)            # [out = d, in = f]: dy @ w_down = d act
    dgu = ops.linear_swiglu_bwd(dy, w_down, gu)
    dact = ops.gemm_nn(dy, w_down)
    dgu_ref = ops.swiglu_bwd(dact, gu, interleaved=True)
    assert torch.equal(dgu, dgu_ref)
    da, sg = dact.float(), torch.sigmoid(g32)
    close(dgu[:, 0::2], da * u32 * sg * (1 + g32 * (1 - sg)), what="fused swiglu d gate")
    close(dgu[:, 1::2], da * g32 * sg, what="fused swiglu d up")
    dblocks = ops.swiglu_bwd(dact, blocks)
    assert torch.equal(torch.cat([dgu_ref[:, 0::2], dgu_ref[:, 1::2]], 1), dblocks)


@pytest.mark.parametrize("M,N,K,act", [(8200, 1792, 2048, 0), (8200, 6144, 1792, 0), (8200, 15360, 1792, 2), (300, 264, 96, 1),
                                        (4100, 1024, 544, 2)])
def test_gemm_nn_bias_act(ops, M, N, K, act):
    """rv_gemm_nn_bias_act_bf16 (round 6): the NN kernels with the bias / activation epilogue of the NT ones - what the frozen EVA
    tower's linears run on with their weights kept only as W^T.  Against torch fp32 and against rv_gemm_nt_bf16 on the other
    orientation of the same weight (same products, another summation order)."""
    dev = _dev()
    x = rnd(M, K, seed=61, dev=dev, scale=1.0)
    w = rnd(N, K, seed=62, dev=dev, scale=0.05)
    bias = rnd(N, seed=63, dev=dev, scale=0.5)
    res = rnd(M, N, seed=64, dev=dev, scale=1.0)
    wT = w.t().contiguous()
    got = ops.gemm_nn(x, wT, bias=bias, act=act, residual=res)
    y = x.float() @ w.float().t() + bias.float()
    if act == 1:
        y = y * torch.sigmoid(1.702 * y)
    elif act == 2:
        y = torch.nn.functional.gelu(y)
    close(got, y + res.float(), what="gemm_nn bias/act")
    if K % 64 == 0:              # (the NT kernels step K by 64; K = 96 / 544 run the 32-deep NN kernel)
        nt = ops.gemm_nt(x, w, bias=bias, act=act, residual=res)
        assert (got.float() - nt.float()).abs().max().item() <= 2e-2 * y.abs().max().item()
        assert (got != nt).float().mean().item() < 0.2          # mostly the same bf16 values: only the summation order differs
    assert torch.equal(ops.gemm_nn(x, wT), ops.gemm_nn(x, wT, bias=None, act=0))


@pytest.mark.parametrize("M,d,f,p", [(6200, 1024, 2048, 0.0), (5003, 512, 3072, 0.05), (12345, 1088, 1536, 0.3)])
def test_lora_swiglu_fused_gemm_epilogues(ops, M, d, f, p):
    """The same two epilogues on the fused-LoRA GEMMs (rv_gemm_nn_lora_swiglu_bf16 / rv_gemm_nn_lora_swiglu_bwd_bf16, round 6;
    RV_LORA_FUSE_SWIGLU): gate|up = x W^T + [t_gate | t_up] bexp with the expanded adapter's zeros keeping the modules apart, the
    activation from the fp32 accumulators, dropout(act) = rv_dropout of the ROUNDED activation written by the same epilogue;
    backward d(gate|up) = SwiGLU'(gu) o (dy W_down + mask o (dt A_down) / (1 - p)).  Ragged row counts, three widths."""
    dev = _dev()
    rp, seed = 64, 12345
    assert ops.linear_lora_swiglu_ok(M, f, d, rp)
    x = rnd(M, d, seed=91, dev=dev, scale=1.0)
    wguT = rnd(d, 2 * f, seed=92, dev=dev, scale=0.06)
    t = rnd(M, 2 * rp, seed=93, dev=dev, scale=0.5)
    bexp = torch.zeros(2 * rp, 2 * f, dtype=torch.bfloat16, device=dev)
    bexp[:rp, 0::2] = rnd(rp, f, seed=94, dev=dev, scale=0.1)
    bexp[rp:, 1::2] = rnd(rp, f, seed=95, dev=dev, scale=0.1)
    gu, act, actd = ops.linear_lora_swiglu(x, wguT, t, bexp, p, seed)
    exact_gu = x.double() @ wguT.double() + t.double() @ bexp.double()
    close(gu, exact_gu.float(), rel=6e-3, what="lora swiglu gate|up")
    exact = torch.nn.functional.silu(exact_gu[:, 0::2]) * exact_gu[:, 1::2]
    close(act, exact.float(), rel=6e-3, what="lora swiglu act")
    unfused = ops.swiglu_fwd(gu, interleaved=True)
    e_f, e_u = (act.double() - exact).abs().mean().item(), (unfused.double() - exact).abs().mean().item()
    assert e_f < 0.75 * e_u, (e_f, e_u)
    if p > 0:
        assert torch.equal(actd, ops.dropout(act, p, seed))
        keep = float((actd != 0).float().mean()) / max(float((act != 0).float().mean()), 1e-9)
        assert abs(keep - (1 - p)) < 5e-3
    else:
        assert actd is None
    # backward
    dy = rnd(M, d, seed=96, dev=dev, scale=0.5)
    w_down = rnd(d, f, seed=97, dev=dev, scale=0.06)
    dt = rnd(M, rp, seed=98, dev=dev, scale=0.5)
    a_down = rnd(rp, f, seed=99, dev=dev, scale=0.1)
    dgu = ops.linear_lora_swiglu_bwd(dy, w_down, dt, a_down, gu, p, seed)
    ad = dt.double() @ a_down.double()
    if p > 0:
        mask = (ops.dropout(torch.ones(M, f, dtype=torch.bfloat16, device=dev), p, seed) != 0).double() / (1 - p)
        ad = ad * mask
    dact = (dy.double() @ w_down.double() + ad).float().bfloat16()          # the unfused path rounds d act once; so does the epilogue
    ref = ops.swiglu_bwd(dact, gu, interleaved=True)
    close(dgu, ref.float(), rel=1.6e-2, what="lora swiglu d(gate|up)")
    g32, u32, da = gu.float()[:, 0::2], gu.float()[:, 1::2], dact.float()
    sg = torch.sigmoid(g32)
    close(dgu[:, 0::2], da * u32 * sg * (1 + g32 * (1 - sg)), rel=1.6e-2, what="lora swiglu d gate")
    close(dgu[:, 1::2], da * g32 * sg, rel=1.6e-2, what="lora swiglu d up")


@pytest.mark.parametrize("M,d,kvd,use_pos", [(27000, 1024, 1024, True), (13000, 2048, 512, False), (300, 1024, 1024, True)])
def test_rope_fused_qkv_gemm_epilogue(ops, M, d, kvd, use_pos):
    """RoPE in the epilogue of the q|k|v projection (rv_gemm_nn_rope_bf16, round 6): the q and k heads leave rotated from the fp32
    accumulators.  Against the exact fp64 product + rotation it must be CLOSER than the unfused composition GEMM -> bf16 ->
    rv_rope_inplace (two roundings) is, and agree with that composition to bf16 rounding; the v columns are the plain GEMM's, bit
    for bit (grouped-query width kvd < d included); positions from a table (packed pairs) or token % L."""
    dev = _dev()
    hd, L = 128, 977
    H, Hkv = d // hd, kvd // hd
    N, rope_cols = d + 2 * kvd, d + kvd
    x = rnd(M, d, seed=81, dev=dev, scale=1.0)
    wT = rnd(d, N, seed=82, dev=dev, scale=0.06)
    cos, sin = ops.rope_tables(2048, hd, 10000.0, dev)
    pos = (torch.arange(M, device=dev, dtype=torch.int32) * 7 % 1500).contiguous() if use_pos else None
    assert ops.linear_rope_ok(M, N, d, rope_cols, hd) == (M > 1000)
    fused = ops.linear_rope(x, wT, cos, sin, pos, L, rope_cols, hd)
    plain = ops.gemm_nn(x, wT)
    assert torch.equal(fused[:, rope_cols:], plain[:, rope_cols:])
    unfused = ops.rope_inplace(plain.clone(), cos, sin, L, H + Hkv, hd, pos=pos)
    close(fused, unfused, rel=1.6e-2, what="fused rope vs gemm + rope_inplace")
    # exact reference
    y = (x.double() @ wT.double())
    p = pos.long() if pos is not None else torch.arange(M, device=dev) % L
    c, s_ = cos[p].double(), sin[p].double()                       # [M, 64]
    yr = y[:, :rope_cols].reshape(M, H + Hkv, 2, 64)
    x1, x2 = yr[:, :, 0], yr[:, :, 1]
    ex = torch.stack([x1 * c[:, None] - x2 * s_[:, None], x2 * c[:, None] + x1 * s_[:, None]], 2).reshape(M, rope_cols)
    e_f = (fused[:, :rope_cols].double() - ex).abs().mean().item()
    e_u = (unfused[:, :rope_cols].double() - ex).abs().mean().item()
    assert e_f < 0.85 * e_u, (e_f, e_u)
    close(fused[:, :rope_cols], ex.float(), rel=8e-3, what="fused rope vs exact")


@pytest.mark.parametrize("R,I,J", [(4096, 4096, 4608), (4352, 8192, 4352), (6000, 4104, 4600)])
def test_gemm_tn_tail_split(ops, R, I, J, monkeypatch):
    """Weight-gradient GEMM whose last round of 256 tiles is partly filled: the tail tiles are split over the token axis into
    fp32 slabs and summed in fixed order (rv_gemm_tn_bf16_ws).  Same result as the plain launch to fp32 summation order,
    deterministic, edge tiles included (I, J not multiples of 256)."""
    dev = _dev()
    from rlaif_v_amd import hip
    need = hip.lib().lib.rv_gemm_tn_workspace_floats(R, I, J)
    assert need > 0, "the case must trigger the tail split"
    p, q = rnd(R, I, seed=1, dev=dev, scale=0.5), rnd(R, J, seed=2, dev=dev, scale=0.5)
    a = ops.gemm_tn(p, q)                                   # tail split (workspace provided by ops)
    b = ops.gemm_tn(p, q)
    assert torch.equal(a, b)                                # deterministic
    plain = torch.empty(I, J, dtype=BF, device=dev)
    hip.call("rv_gemm_tn_bf16", p, p.stride(0), q, q.stride(0), plain, plain.stride(0), R, I, J, None, 0, 1.0)
    ref = p.float().t() @ q.float()
    close(a, ref, what=f"gemm_tn tail split {R}x{I}x{J}")
    close(plain, ref, what="gemm_tn plain")
    assert (a.float() - plain.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    frac_same = (a == plain).float().mean().item()
    assert frac_same > 0.7, frac_same                       # tiles of the full rounds are bit-identical, the tail differs in fp32 order only


def test_attn_bwd_dkv5_random_shapes_vs_dkv3_and_reference(ops):
    """Round 4: the default dK/dV kernel (version 5: generated tile bodies, four-stage LDS ring, counted waits, read-out through
    LDS) on a sweep of seeded random shapes - ragged lengths around the 64 / 128 tile edges, packed pair rows with arbitrary
    (shared, chosen-end) bounds incl. empty branches, grouped-query heads, causal and full - against the round-2/3 kernel in the
    SAME process (rv_set_attn_dkv_version) and against fp32 torch attention.  Also: every launch twice, bit-identical."""
    from rlaif_v_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(20260926)
    hd = 128
    cases = [(1, 1, 1, 1, True, None), (2, 2, 1, 63, True, None), (1, 2, 2, 64, False, None), (2, 4, 2, 65, True, None),
             (1, 2, 1, 127, True, None), (1, 2, 1, 129, False, None), (2, 2, 1, 191, True, None), (1, 6, 3, 320, True, None)]
    for _ in range(10):
        S = int(torch.randint(1, 4, (1,), generator=g))
        G = int([1, 1, 2, 4][int(torch.randint(0, 4, (1,), generator=g))])
        H = G * int(torch.randint(1, 3, (1,), generator=g))
        L = int(torch.randint(2, 700, (1,), generator=g))
        segs = []
        for _s in range(S):
            a = int(torch.randint(0, L, (1,), generator=g))
            b = int(torch.randint(a, L + 1, (1,), generator=g))
            segs.append((a, b))
        cases.append((S, H, G, L, True, segs))
    try:
        for (S, H, G, L, causal, segs) in cases:
            Hkv = H // G
            width = (H + 2 * Hkv) * hd
            kc, vc = H * hd, (H + Hkv) * hd
            qkv = rnd(S * L, width, seed=L + 7 * H + S, dev=dev, scale=0.7)
            do = rnd(S * L, H * hd, seed=L + 1, dev=dev)
            seg = None
            if segs is not None:
                seg = (torch.tensor([a for a, _ in segs], dtype=torch.int32, device=dev),
                       torch.tensor([b for _, b in segs], dtype=torch.int32, device=dev))
            out, lse = ops.attn_fwd(qkv, S, L, H, hd, causal, 0, kc, vc, seg=seg, kv_group=G)
            res = {}
            for ver in (5, 3):
                hip.call("rv_set_attn_dkv_version", ver)
                a = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, causal, 0, kc, vc, seg=seg, kv_group=G)
                b = ops.attn_bwd(qkv, out, do, lse, S, L, H, hd, causal, 0, kc, vc, seg=seg, kv_group=G)
                assert torch.equal(a, b), f"dkv{ver} not deterministic: S{S} H{H} G{G} L{L} causal={causal} segs={segs}"
                assert torch.isfinite(a.float()).all(), f"dkv{ver}: non-finite: S{S} H{H} G{G} L{L} segs={segs}"
                res[ver] = a
            # reference (fp32)
            qf = qkv.float().requires_grad_(True)
            q = qf[:, :kc].view(S, L, H, hd).transpose(1, 2)
            k = qf[:, kc:vc].view(S, L, Hkv, hd).transpose(1, 2).repeat_interleave(G, dim=1)
            v = qf[:, vc:].view(S, L, Hkv, hd).transpose(1, 2).repeat_interleave(G, dim=1)
            if segs is not None:
                mask = torch.stack([_packed_mask(L, a, b, dev) for a, b in segs])[:, None]
            elif causal:
                mask = torch.full((L, L), float("-inf"), device=dev).triu(1)[None, None]
            else:
                mask = torch.zeros(1, 1, L, L, device=dev)
            ro = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask, -1) @ v
            ro.transpose(1, 2).reshape(S * L, H * hd).backward(do.float())
            what = f"S{S} H{H} G{G} L{L} causal={causal} segs={segs}"
            for nm, sl in (("q", slice(0, kc)), ("k", slice(kc, vc)), ("v", slice(vc, width))):
                close(res[5][:, sl], qf.grad[:, sl], rel=2.5e-2, what=f"dkv5 d{nm} {what}")
                close(res[5][:, sl], res[3][:, sl], rel=1.0e-2, what=f"dkv5 vs dkv3 d{nm} {what}")
    finally:
        hip.call("rv_set_attn_dkv_version", 0)
