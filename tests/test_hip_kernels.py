"""GPU parity tests for the unit kernels behind the C-ABI (libbuddy_hip.so) against plain torch fp32 references
of the same op.  Tolerances: fp32 MFMA accumulates exactly like an fmaf chain, so differences to torch are pure
summation-order round-off: rel-to-absmax 2e-5 for K <= 4608 contractions, 1e-5 for normalisation."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.fixture(scope="module")
def lib():
    from buddy_amd import _lib
    return _lib.require_gpu()


def P(t):
    return None if t is None else t.data_ptr()


def S():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,batch", [(256, 128, 64, 1), (200, 72, 132, 3), (2048, 2048, 256, 2)])
def test_gemm(lib, tA, tB, M, N, K, batch):
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M + N + K + tA * 2 + tB)
    A = torch.randn(batch, M, K, generator=g).cuda()
    Bm = torch.randn(batch, N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    As = A.transpose(1, 2).contiguous() if tA else A
    Bs = Bm.transpose(1, 2).contiguous() if tB else Bm
    Cc = torch.full((batch, M, N), 7.0, device="cuda")
    ref = 0.5 * torch.einsum("bmk,bnk->bmn", A.double(), Bm.double()).float() + bias
    _lib.check(lib.buddy_gemm(P(As), M if tA else K, tA, P(Bs), N if tB else K, tB, P(Cc), N, M, N, K, 0.5, P(bias), 0, batch,
                              M * K, N * K, M * N, S()))
    torch.cuda.synchronize()
    assert rel(Cc, ref) < 2e-5
    # accumulate
    _lib.check(lib.buddy_gemm(P(As), M if tA else K, tA, P(Bs), N if tB else K, tB, P(Cc), N, M, N, K, 0.5, None, 1, batch,
                              M * K, N * K, M * N, S()))
    torch.cuda.synchronize()
    assert rel(Cc, 2 * ref - bias) < 2e-5


@pytest.mark.parametrize("tiles,Cout,Cin,P_", [(300, 128, 128, 3), (1000, 256, 384, 2), (129, 256, 32, 5), (4096, 128, 512, 2)])
def test_winograd_domain_gemm_bf16x3(lib, tiles, Cout, Cin, P_):
    """buddy_gemm_winograd_domain_bf16x3 (exact three-way bf16 split, six bf16 MFMA products, fp32 accumulate; csrc/wgemm.hip) against fp64:
    the SAME bound as the fp32-MFMA GEMM (2e-5 of the abs-max), measured next to it; ragged row counts, both column-block counts.  Values
    with a wide dynamic range (the Winograd-domain operands span ~4 decades) so that a dropped low-order term would show."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(tiles + Cout + Cin)
    V = (torch.randn(P_, tiles, Cin, generator=g) * torch.exp(2.0 * torch.randn(P_, tiles, 1, generator=g))).cuda()
    U = (torch.randn(P_, Cout, Cin, generator=g) * torch.exp(1.5 * torch.randn(P_, 1, Cin, generator=g))).cuda()
    ref = torch.einsum("pmk,pnk->pmn", V.double(), U.double())
    nbytes = int(lib.buddy_wgemm_packed_bytes(P_, Cout, Cin))
    assert nbytes == P_ * Cout * Cin * 6
    U3 = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_pack_weights(P(U), U3.data_ptr(), P_, Cout, Cin, S()))
    M3 = torch.full((P_, tiles, Cout), 7.0, device="cuda")
    _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(V), U3.data_ptr(), P(M3), tiles, Cout, Cin, P_, S()))
    M1 = torch.empty(P_, tiles, Cout, device="cuda")
    _lib.check(lib.buddy_gemm_winograd_domain(P(V), P(U), P(M1), tiles, Cout, Cin, P_, S()))
    torch.cuda.synchronize()
    e3, e1 = rel(M3, ref), rel(M1, ref)
    # relative to the row-wise scale as well (rows span decades): the worst row
    rr = lambda X: float(((X.double() - ref).abs().amax(dim=2) / ref.abs().amax(dim=2)).max())
    print(f"tiles={tiles} Cout={Cout} Cin={Cin}: bf16x3 {e3:.2e} (worst row {rr(M3):.2e}), fp32 MFMA {e1:.2e} (worst row {rr(M1):.2e})")
    assert e3 < 2e-5 and rr(M3) < 2e-5
    assert lib.buddy_wgemm_packed_bytes(2, 96, 128) == 0 and lib.buddy_wgemm_packed_bytes(2, 128, 48) == 0


@pytest.mark.parametrize("utts,tpu,Cout,Cin,P_", [(3, 100, 128, 128, 3), (2, 500, 256, 384, 2), (1, 129, 256, 64, 5), (4, 1024, 128, 512, 2)])
def test_winograd_domain_gemm_f16x2(lib, utts, tpu, Cout, Cin, P_):
    """buddy_gemm_winograd_domain_f16x2 (two-way f16 split of power-of-two-scaled operands, three f16 MFMA products, fp32 accumulate; csrc/wgemm.hip)
    against fp64.  The operands are good to 2^-22, so the bound is 4x the bf16x3 / fp32 kernels' (8e-5 of the abs-max; measured beside them); utterances of
    very different level in one batch (per-utterance scale), weights and rows spanning decades, ragged row counts, and the batch-independence of an
    utterance's result, bit for bit."""
    from buddy_amd import _lib
    tiles = utts * tpu
    g = torch.Generator(device="cpu").manual_seed(tiles + Cout + Cin)
    level = torch.tensor([1.0, 3e-4, 2e3, 17.0])[:utts].repeat_interleave(tpu)[None, :, None]
    V = (torch.randn(P_, tiles, Cin, generator=g) * torch.exp(2.0 * torch.randn(P_, tiles, 1, generator=g)) * level).cuda()
    U = (torch.randn(P_, Cout, Cin, generator=g) * torch.exp(1.5 * torch.randn(P_, 1, Cin, generator=g)) * 1e-2).cuda()
    ref = torch.einsum("pmk,pnk->pmn", V.double(), U.double())
    nbytes = int(lib.buddy_wgemm_f16x2_packed_bytes(P_, Cout, Cin))
    assert nbytes == P_ * Cout * Cin * 4 + 512
    U2 = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_f16x2_pack_weights(P(U), U2.data_ptr(), P_, Cout, Cin, S()))
    vmax = torch.empty(utts, 64, 32, dtype=torch.int32, device="cuda")      # 64 partial maxima per utterance, one per 128-byte line (include/buddy_hip.h)
    _lib.check(lib.buddy_abs_max_bits(P(V), P_, utts, tpu * Cin, vmax.data_ptr(), S()))
    assert torch.equal(vmax.view(torch.float32)[:, :, 0].amax(dim=1), V.reshape(P_, utts, -1).abs().amax(dim=(0, 2)))
    M2 = torch.full((P_, tiles, Cout), 7.0, device="cuda")
    _lib.check(lib.buddy_gemm_winograd_domain_f16x2(P(V), U2.data_ptr(), P(M2), tiles, Cout, Cin, P_, vmax.data_ptr(), tpu, S()))
    M1 = torch.empty(P_, tiles, Cout, device="cuda")
    _lib.check(lib.buddy_gemm_winograd_domain(P(V), P(U), P(M1), tiles, Cout, Cin, P_, S()))
    torch.cuda.synchronize()
    # every utterance against ITS abs-max (the levels differ by 7 decades)
    worst2 = worst1 = 0.0
    for u in range(utts):
        sl = slice(u * tpu, (u + 1) * tpu)
        worst2 = max(worst2, rel(M2[:, sl], ref[:, sl])); worst1 = max(worst1, rel(M1[:, sl], ref[:, sl]))
    rr = lambda X: float(((X.double() - ref).abs().amax(dim=2) / ref.abs().amax(dim=2)).max())
    print(f"utts={utts} tiles/utt={tpu} Cout={Cout} Cin={Cin}: f16x2 {worst2:.2e} (worst row {rr(M2):.2e}), fp32 MFMA {worst1:.2e} (worst row {rr(M1):.2e})")
    assert worst2 < 8e-5 and rr(M2) < 2e-3          # a row 2^-18 below its utterance's abs-max keeps >= 2^-22 * 2^... of absolute accuracy (see wgemm.hip)
    # utterance 0 alone == utterance 0 inside the batch
    V0 = V[:, :tpu].contiguous()
    M0 = torch.empty(P_, tpu, Cout, device="cuda")
    _lib.check(lib.buddy_gemm_winograd_domain_f16x2(P(V0), U2.data_ptr(), P(M0), tpu, Cout, Cin, P_, vmax.data_ptr(), tpu, S()))
    torch.cuda.synchronize()
    assert torch.equal(M0, M2[:, :tpu])
    assert lib.buddy_wgemm_f16x2_packed_bytes(2, 96, 128) == 0 and lib.buddy_wgemm_f16x2_packed_bytes(65, 128, 128) == 0 and lib.buddy_wgemm_f16x2_packed_bytes(2, 128, 96) == 0


def _packed_1x1(lib, W, arith):
    """stage image of a [N][K] weight for the general GEMM forms: bf16x3 (three exact bf16 planes) or f16x2 (two f16 planes + the matrix's power of two)"""
    from buddy_amd import _lib
    N, K = W.shape
    if arith == "f16x2":
        W3 = torch.empty(lib.buddy_wgemm_f16x2_packed_bytes(1, N, K) // 4, dtype=torch.int32, device="cuda")
        _lib.check(lib.buddy_wgemm_f16x2_pack_weights(P(W), W3.data_ptr(), 1, N, K, S()))
    else:
        W3 = torch.empty(N * K * 6 // 4, dtype=torch.int32, device="cuda")
        _lib.check(lib.buddy_wgemm_pack_weights(P(W), W3.data_ptr(), 1, N, K, S()))
    return W3


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("M,N,K,C0", [(1000, 128, 384, 256), (4097, 256, 256, 0), (300, 128, 64, 32)])
def test_gemm_bf16x3_general_form(lib, M, N, K, C0, arith):
    """buddy_gemm_bf16x3 / buddy_gemm_f16x2 (the 1x1 convolutions / NINs on the split-arithmetic kernels): two-source A (channel concatenation split at C0;
    0 = one source), bias, alpha, accumulate, ragged M -- against fp64, the fp32 GEMM's bound.  f16x2 (round 6): the A operand's power of two is taken per
    row inside the kernel, so rows ten decades apart in one launch keep their relative accuracy."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    if arith == "f16x2":
        A = A * torch.logspace(-6, 4, M).cuda()[:, None]
    W = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    W3 = _packed_1x1(lib, W, arith)
    gemm = lib.buddy_gemm_f16x2 if arith == "f16x2" else lib.buddy_gemm_bf16x3
    A0 = A[:, :C0].contiguous() if C0 else A
    A1 = A[:, C0:].contiguous() if C0 else None
    Cc = torch.full((M, N), 3.0, device="cuda")
    ref = 0.5 * (A.double() @ W.double().t()).float() + bias
    if arith == "f16x2":          # row by row, without the bias: every row against its own abs-max (the rows span ten decades)
        _lib.check(gemm(P(A0), A0.shape[1], P(A1), A1.shape[1] if C0 else 0, C0, W3.data_ptr(), P(Cc), N, M, N, K, None, 0.5, 0, S()))
        torch.cuda.synchronize()
        ref0 = 0.5 * (A.double() @ W.double().t())
        rowerr = ((Cc.double() - ref0).abs().amax(dim=1) / (ref0.abs().amax(dim=1) + 1e-300)).max()
        assert float(rowerr) < 2e-5, float(rowerr)
    _lib.check(gemm(P(A0), A0.shape[1], P(A1), A1.shape[1] if C0 else 0, C0, W3.data_ptr(), P(Cc), N, M, N, K, P(bias), 0.5, 0, S()))
    torch.cuda.synchronize()
    assert rel(Cc, ref) < 2e-5
    _lib.check(gemm(P(A0), A0.shape[1], P(A1), A1.shape[1] if C0 else 0, C0, W3.data_ptr(), P(Cc), N, M, N, K, None, 0.5, 1, S()))
    torch.cuda.synchronize()
    assert rel(Cc, 2 * ref - bias) < 2e-5


@pytest.mark.parametrize("M,N,K", [(40000, 128, 256), (40000, 256, 512), (333, 256, 128)])
def test_gemm_f16x2_running_row_scale(lib, M, N, K):
    """The f16x2 general form finds a row's power of two ON THE WAY (wgemm_f16x2_gen_kernel: a K-stage whose abs-max leaves the current scale's range
    rescales the row's accumulators by an exact power of two).  Rows whose magnitude climbs twelve decades along K (a rescale at almost every stage), rows
    that fall as far (the first stage fixes the scale), zero rows, one huge element in the last stage, and a zero first half: every row within 2e-5 of its
    own bound sum_k |a_k| |w_k| -- what a per-row scale known in advance would give.  The three shapes run the three kernel forms (64-row waves, two column
    blocks per workgroup, 32-row waves: M and N select them, wgemm.hip gen_rows64 / gen_colpair)."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(7 * M + N + K)
    A = torch.randn(M, K, generator=g)
    ramp = torch.logspace(-6, 6, K)
    kind = torch.arange(M) % 6
    A[kind == 0] *= ramp
    A[kind == 1] *= ramp.flip(0)
    A[kind == 2] = 0.0
    A[kind == 3, -1] = 3.0e7
    A[kind == 4, :K // 2] = 0.0
    A = A.cuda()
    W = torch.randn(N, K, generator=g).cuda()
    W2 = _packed_1x1(lib, W, "f16x2")
    Cc = torch.full((M, N), 3.0, device="cuda")
    _lib.check(lib.buddy_gemm_f16x2(P(A), K, None, 0, 0, W2.data_ptr(), P(Cc), N, M, N, K, None, 1.0, 0, S()))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t()
    bound = A.double().abs() @ W.double().abs().t()
    err = ((Cc.double() - ref).abs() / (bound + 1e-300)).max()
    assert float(err) < 2e-6, float(err)
    assert torch.equal(Cc[kind.cuda() == 2], torch.zeros_like(Cc[kind.cuda() == 2]))


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("B,HW,N,K,C0,silu,acc", [(2, 300, 128, 128, 0, 1, 0), (1, 1000, 384, 128, 256, 1, 1), (3, 77, 256, 64, 128, 0, 1), (1, 4096, 512, 256, 256, 1, 0)])
def test_gemm_bf16x3_gn_bwd(lib, B, HW, N, K, C0, silu, acc, arith):
    """buddy_gemm_bf16x3_gn_bwd: the skip path's 1x1 data-gradient GEMM with the GroupNorm_0 backward's apply pass as its epilogue (a ResBlock's input
    gradient, reference layerspp.py:242-274 backward) against fp64 autograd: dx = alpha * A W^T + d/dx [act(GroupNorm(x))] . da, two-source x,
    two-destination dx (the second accumulating), ragged M.  2e-5 for the GEMM term like the general form, 2e-4 of the abs-max overall (the
    normalisation backward's cancellation, as test_gnbwd_conv3x3_winograd6)."""
    from buddy_amd import _lib
    G = min(N // 4, 32)
    g = torch.Generator(device="cpu").manual_seed(B * 100 + HW + N + K)
    A = torch.randn(B * HW, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    x = (torch.randn(B, HW, N, generator=g) * 1.5 + 0.3).cuda()
    da = torch.randn(B, HW, N, generator=g).cuda()
    gamma = (1 + 0.2 * torch.randn(N, generator=g)).cuda()
    beta = (0.2 * torch.randn(N, generator=g)).cuda()
    prev = torch.randn(B, HW, N, generator=g).cuda()
    xd = x.double().permute(0, 2, 1).requires_grad_(True)                      # (B, C, HW) for F.group_norm
    z = F.group_norm(xd, G, gamma.double(), beta.double(), eps=1e-6)
    a = F.silu(z) if silu else z
    gx, = torch.autograd.grad(a, xd, da.double().permute(0, 2, 1))
    ref = 0.70710678 * (A.double() @ W.double().t()).reshape(B, HW, N) + gx.permute(0, 2, 1)
    xg = x.double().permute(0, 2, 1).reshape(B, G, N // G, HW)
    mean = xg.mean(dim=(2, 3)); rstd = 1.0 / torch.sqrt(xg.var(dim=(2, 3), unbiased=False) + 1e-6)
    stats = torch.stack([mean, rstd], dim=-1).float().contiguous()
    W3 = _packed_1x1(lib, W, arith)
    x0 = x[..., :C0].contiguous() if C0 else x
    x1 = x[..., C0:].contiguous() if C0 else None
    d0 = torch.full_like(x0, float("nan"))
    d1 = prev[..., C0:].contiguous() if C0 else None
    if C0 and not acc:
        d1.fill_(float("nan"))
    stat_scratch = torch.empty(B * 256 * N * 2, dtype=torch.float64, device="cuda")
    red = torch.empty(B, G, 2, device="cuda")
    _lib.check((lib.buddy_gemm_f16x2_gn_bwd if arith == "f16x2" else lib.buddy_gemm_bf16x3_gn_bwd)(P(A), K, W3.data_ptr(), P(x0), P(x1) if C0 else None, C0, P(da), P(stats), P(gamma), P(beta), G, silu, 0.70710678,
                                            P(d0), P(d1) if C0 else None, 0, acc if C0 else 0, stat_scratch.data_ptr(), P(red), B, HW, N, K, S()))
    torch.cuda.synchronize()
    out = torch.cat([d0, d1], dim=-1) if C0 else d0
    if C0 and acc:
        ref[..., C0:] += prev[..., C0:].double()
    e = rel(out, ref.float())
    print(f"1x1 data-gradient + GroupNorm backward apply {B}x{HW} K={K} -> {N}: {e:.2e}")
    assert e < 2e-4


def test_winograd_domain_gemm_bf16x3_layout(lib):
    """V = I-like selector against asymmetric weights: every (row, channel, k) lands where it should (exactly representable values)."""
    from buddy_amd import _lib
    tiles, Cout, Cin = 128, 256, 128
    V = torch.zeros(1, tiles, Cin, device="cuda")
    V[0, torch.arange(tiles), (torch.arange(tiles) * 37) % Cin] = 1.0
    U = ((torch.arange(Cout * Cin, device="cuda", dtype=torch.float32).reshape(1, Cout, Cin) % 251) + 0.5)
    U3 = torch.empty(Cout * Cin * 6 // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_pack_weights(P(U), U3.data_ptr(), 1, Cout, Cin, S()))
    Mo = torch.empty(1, tiles, Cout, device="cuda")
    _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(V), U3.data_ptr(), P(Mo), tiles, Cout, Cin, 1, S()))
    torch.cuda.synchronize()
    assert torch.equal(Mo[0], U[0][:, (torch.arange(tiles) * 37) % Cin].t().contiguous())


def test_gemm_asymmetric_layout(lib):
    """A = I against an asymmetric B catches a transposed C write (guide rule: always A=I with asymmetric B)."""
    from buddy_amd import _lib
    n = 128
    A = torch.eye(n, device="cuda")
    Bm = (torch.arange(n * n, device="cuda", dtype=torch.float32).reshape(n, n) % 97) + 0.25 * torch.arange(n, device="cuda")[:, None]
    Cc = torch.empty(n, n, device="cuda")
    _lib.check(lib.buddy_gemm(P(A), n, 0, P(Bm), n, 0, P(Cc), n, n, n, n, 1.0, None, 0, 1, 0, 0, 0, S()))
    torch.cuda.synchronize()
    assert torch.equal(Cc, Bm.t().contiguous())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 16, 24, 32, 64), (2, 33, 20, 96, 32), (1, 64, 128, 256, 128), (1, 8, 8, 384, 256)])
def test_conv3x3(lib, B, H, W, Cin, Cout):
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()        # [o][(ky*3+kx)*Cin + i]; here H is "our H"
    y = torch.empty(B, H, W, Cout, device="cuda")
    _lib.check(lib.buddy_conv3x3(P(x_nhwc), P(wt), P(b), P(y), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    assert rel(y.permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 16, 32, 32, 64), (2, 40, 64, 96, 32), (1, 64, 128, 256, 128), (3, 8, 32, 384, 256)])
def test_conv3x3_winograd(lib, B, H, W, Cin, Cout):
    """Fused Winograd F(2x2,3x3): same tolerance class as the direct kernel (fp32, different summation order)."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout + 1)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.double(), w.cuda().double(), b.double(), padding=1).float()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().numpy()
    U = np.empty(16 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd_transform_weights(wt.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    y = torch.empty(B, H, W, Cout, device="cuda")
    _lib.check(lib.buddy_conv3x3_winograd(P(x_nhwc), P(Ud), P(b), P(y), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    assert rel(y.permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 24, 16, 12), (1, 64, 32, 128, 256), (3, 8, 12, 40, 8), (2, 16, 16, 384, 128)])
def test_conv3x3_winograd4(lib, B, H, W, Cin, Cout):
    """Three-pass Winograd F(4x4,3x3) (input transform, 36 batched fp32 GEMMs, output transform).  Stated tolerance 1e-4 relative to the
    abs-max (the interpolation points 0, +-1, +-2, inf amplify fp32 round-off by about one decimal digit over the direct form: measured 3e-6)."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout + 4)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.double(), w.cuda().double(), b.double(), padding=1).float()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().numpy()
    U = np.empty(36 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd4_transform_weights(wt.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    y = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(36 * (B * H * W // 16) * (Cin + Cout), device="cuda")
    _lib.check(lib.buddy_conv3x3_winograd4(P(x_nhwc), P(Ud), P(b), P(y), P(scratch), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    assert rel(y.permute(0, 3, 1, 2), ref) < 1e-4


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 24, 16, 12), (1, 64, 32, 128, 256), (3, 8, 14, 40, 8), (2, 20, 16, 384, 128), (1, 6, 6, 8, 4),
                                            (2, 37, 50, 96, 64)])
def test_conv3x3_winograd6(lib, B, H, W, Cin, Cout):
    """Three-pass Winograd F(6x6,3x3) (separable transforms through LDS, 64 batched fp32 GEMMs), H / W not multiples of 6 (overhanging
    tiles), channel counts that do not fill a 32-quad chunk.  Stated tolerance 1e-4 of the abs-max as for F(4x4,3x3) (measured ~1e-5: about
    twice the round-off of F(4x4,3x3))."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout + 6)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.double(), w.cuda().double(), b.double(), padding=1).float()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().numpy()
    U = np.empty(64 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd6_transform_weights(wt.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    y = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    tiles = B * ((H + 5) // 6) * ((W + 5) // 6)
    scratch = torch.empty(64 * tiles * (Cin + Cout), device="cuda")
    _lib.check(lib.buddy_conv3x3_winograd6(P(x_nhwc), P(Ud), P(b), P(y), P(scratch), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    e = rel(y.permute(0, 3, 1, 2), ref)
    print(f"F(6x6,3x3) {B}x{H}x{W} {Cin}->{Cout}: {e:.2e}")
    assert e < 1e-4


@pytest.mark.parametrize("f6", [0, 1])
@pytest.mark.parametrize("B,H,W,C0,C1,Cout,silu,stat", [(2, 32, 32, 16, 0, 32, 1, 1), (1, 64, 32, 128, 0, 256, 1, 1), (3, 8, 12, 24, 16, 8, 1, 0),
                                                       (2, 16, 16, 256, 128, 128, 1, 1), (2, 16, 32, 64, 0, 64, 0, 1)])
def test_gn_conv3x3_winograd_fused(lib, B, H, W, C0, C1, Cout, silu, stat, f6):
    """conv3x3(act(GroupNorm(cat[x0, x1]))) with the normalisation + SiLU applied inside the F(4x4,3x3) input transform and the per-channel
    (sum, sum of squares) of the output left by the output transform (the next GroupNorm's statistics), vs fp64 torch.  1e-4 like the unfused
    convolution; the sums are fp64 sums of the fp32 outputs: 1e-6 relative to sum |y| / sum y^2."""
    from buddy_amd import _lib
    Cin = C0 + C1
    G = min(Cin // 4, 32)
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout + 7)
    x = (torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.3).cuda()
    gamma = (1 + 0.2 * torch.randn(Cin, generator=g)).cuda()
    beta = (0.2 * torch.randn(Cin, generator=g)).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin))
    b = torch.randn(Cout, generator=g).cuda()
    z = F.group_norm(x.double(), G, gamma.double(), beta.double(), eps=1e-6)
    if silu:
        z = F.silu(z)
    ref = F.conv2d(z, w.cuda().double(), b.double(), padding=1)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().numpy()
    U = np.empty((64 if f6 else 36) * Cin * Cout, dtype=np.float32)
    _lib.check((lib.buddy_winograd6_transform_weights if f6 else lib.buddy_winograd4_transform_weights)(wt.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    xn = x.permute(0, 2, 3, 1)
    x0 = xn[..., :C0].contiguous()
    x1 = xn[..., C0:].contiguous() if C1 else None
    y = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(64 * B * ((H + 5) // 6) * ((W + 5) // 6) * (Cin + Cout) if f6 else 36 * (B * H * W // 16) * (Cin + Cout), device="cuda")
    stats = torch.empty(B, G, 2, device="cuda")
    stat_scratch = torch.empty(B * 256 * 1024 * 2, dtype=torch.float64, device="cuda")
    csum = torch.zeros(B, Cout, 2, dtype=torch.float64, device="cuda")
    _lib.check((lib.buddy_gn_conv3x3_winograd6 if f6 else lib.buddy_gn_conv3x3_winograd4)(P(x0), P(x1) if C1 else None, C0, P(gamma), P(beta), G, silu, P(Ud), P(b), P(y), P(scratch), P(stats),
                                              stat_scratch.data_ptr(), csum.data_ptr() if stat else None, B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    assert rel(y.permute(0, 3, 1, 2), ref.float()) < 1e-4
    if stat:
        yd = y.double()
        s_ref, q_ref = yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))
        assert float((csum[..., 0] - s_ref).abs().max() / yd.abs().sum(dim=(1, 2)).max()) < 1e-6
        assert float((csum[..., 1] - q_ref).abs().max() / q_ref.max()) < 1e-6


@pytest.mark.parametrize("B,H,W,Cin,C0,C1,silu", [(2, 32, 32, 16, 32, 0, 1), (1, 64, 32, 256, 128, 0, 1), (2, 20, 16, 128, 256, 128, 1),
                                                  (2, 16, 33, 64, 64, 0, 0)])
def test_conv3x3_winograd6_gn_bwd_sums(lib, B, H, W, Cin, C0, C1, silu):
    """Data-gradient convolution (F(6x6,3x3) passes) whose output transform also leaves the GroupNorm-backward sums of the tensor it produces
    the gradient for: (sum dxhat, sum dxhat * xhat) per (utterance, channel), dxhat = da * act'(z) * gamma, against fp64 torch evaluated
    on the SAME da.  1e-5 relative to sum |dxhat| (fp32 products, v_exp_f32 / v_rcp_f32 in act', fp64 accumulation beyond one tile row)."""
    from buddy_amd import _lib
    Cout = C0 + C1
    G = min(Cout // 4, 32)
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout + 11)
    gr = torch.randn(B, H, W, Cin, generator=gen).cuda()
    x = (torch.randn(B, H, W, Cout, generator=gen) * 1.5 + 0.3).cuda()
    gamma = (1 + 0.2 * torch.randn(Cout, generator=gen)).cuda()
    beta = (0.2 * torch.randn(Cout, generator=gen)).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(9 * Cin))
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().numpy()
    U = np.empty(64 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd6_transform_weights(wt.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    xd = x.double()
    xg = xd.reshape(B, H * W, G, Cout // G)
    mean = xg.mean(dim=(1, 3)); var = xg.var(dim=(1, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    stats = torch.stack([mean, rstd], dim=-1).float().contiguous()
    x0 = x[..., :C0].contiguous()
    x1 = x[..., C0:].contiguous() if C1 else None
    da = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(64 * B * ((H + 5) // 6) * ((W + 5) // 6) * (Cin + Cout), device="cuda")
    stat_scratch = torch.empty(B * 256 * 1024 * 2, dtype=torch.float64, device="cuda")
    chsum = torch.zeros(B, Cout, 2, dtype=torch.float64, device="cuda")
    _lib.check(lib.buddy_conv3x3_winograd6_gn_bwd_sums(P(gr), P(Ud), P(da), P(scratch), P(x0), P(x1) if C1 else None, C0, P(stats), P(gamma), P(beta),
                                                       G, silu, stat_scratch.data_ptr(), chsum.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    ref = F.conv2d(gr.permute(0, 3, 1, 2).double(), w.cuda().double(), None, padding=1).permute(0, 2, 3, 1)
    assert rel(da, ref.float()) < 1e-4
    mean_c = stats[..., 0].double().repeat_interleave(Cout // G, dim=1)[:, None, None, :]
    rstd_c = stats[..., 1].double().repeat_interleave(Cout // G, dim=1)[:, None, None, :]
    xh = (xd - mean_c) * rstd_c
    z = xh * gamma.double() + beta.double()
    sg = torch.sigmoid(z)
    dact = sg * (1 + z * (1 - sg)) if silu else torch.ones_like(z)
    dxh = da.double() * dact * gamma.double()
    s_ref, t_ref = dxh.sum(dim=(1, 2)), (dxh * xh).sum(dim=(1, 2))
    assert float((chsum[..., 0] - s_ref).abs().max() / dxh.abs().sum(dim=(1, 2)).max()) < 1e-5
    assert float((chsum[..., 1] - t_ref).abs().max() / (dxh * xh).abs().sum(dim=(1, 2)).max()) < 1e-5


@pytest.mark.parametrize("B,H,W,C,Cout,silu", [(2, 32, 32, 32, 16, 1), (1, 64, 32, 128, 256, 1), (2, 20, 16, 256, 128, 1), (2, 16, 33, 64, 64, 0)])
def test_gnbwd_conv3x3_winograd6(lib, B, H, W, C, Cout, silu):
    """conv3x3(dx) with dx = the input-gradient of act(GroupNorm(x)) for an incoming gradient da, the GroupNorm backward's apply pass
    evaluated inside the F(6x6,3x3) input transform, against fp64 autograd.  2e-4 of the abs-max (the backward of a normalisation subtracts
    the two group means from dxhat: cancellation on top of the convolution's 1e-4)."""
    from buddy_amd import _lib
    G = min(C // 4, 32)
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + C + Cout + 13)
    x = (torch.randn(B, C, H, W, generator=gen) * 1.5 + 0.3).cuda()
    da = torch.randn(B, C, H, W, generator=gen).cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=gen)).cuda()
    beta = (0.2 * torch.randn(C, generator=gen)).cuda()
    w = (torch.randn(Cout, C, 3, 3, generator=gen) / np.sqrt(9 * C))
    xd = x.double().requires_grad_(True)
    z = F.group_norm(xd, G, gamma.double(), beta.double(), eps=1e-6)
    a = F.silu(z) if silu else z
    dx, = torch.autograd.grad(a, xd, da.double())
    ref = F.conv2d(dx, w.cuda().double(), None, padding=1)
    xg = x.double().reshape(B, G, C // G, H * W)
    mean = xg.mean(dim=(2, 3)); rstd = 1.0 / torch.sqrt(xg.var(dim=(2, 3), unbiased=False) + 1e-6)
    stats = torch.stack([mean, rstd], dim=-1).float().contiguous()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).contiguous().numpy()
    U = np.empty(64 * C * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd6_transform_weights(wt.ctypes.data, Cout, C, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    xn, dn = x.permute(0, 2, 3, 1).contiguous(), da.permute(0, 2, 3, 1).contiguous()
    y = torch.empty(B, H, W, Cout, device="cuda")
    scratch = torch.empty(64 * B * ((H + 5) // 6) * ((W + 5) // 6) * (C + Cout), device="cuda")
    stat_scratch = torch.empty(B * 256 * C * 2, dtype=torch.float64, device="cuda")
    red = torch.empty(B, G, 2, device="cuda")
    _lib.check(lib.buddy_gnbwd_conv3x3_winograd6(P(xn), P(gamma), P(beta), P(stats), P(dn), G, silu, P(Ud), P(y), P(scratch), stat_scratch.data_ptr(), P(red),
                                                 B, H, W, C, Cout, S()))
    torch.cuda.synchronize()
    e = rel(y.permute(0, 3, 1, 2), ref.float())
    print(f"gn-bwd + F(6x6,3x3) {B}x{H}x{W} {C}->{Cout}: {e:.2e}")
    assert e < 2e-4


@pytest.mark.parametrize("B,H,W,Cin,Cout,silu,stat", [(2, 16, 16, 32, 32, 1, 1), (1, 32, 20, 128, 256, 1, 1), (2, 13, 18, 64, 128, 0, 0),
                                                        (1, 24, 36, 256, 256, 1, 1)])
def test_gn_upconv3x3_winograd6(lib, B, H, W, Cin, Cout, silu, stat):
    """The up block's Conv_0(naive_upsample_2d(act(GroupNorm_0(x)))) (layerspp.py:243-257) in its sub-pixel form: one F(6x6,3x3) convolution on the
    low-resolution grid with four phase kernels (summed taps, buddy_conv3_weight_prep kind 61) and a depth-to-space output transform, against
    fp64 torch on the materialised upsampled tensor.  Network axis convention: H = time (kx), W = frequency (ky).  Same 1e-4 as the plain form."""
    from buddy_amd import _lib
    G = min(Cin // 4, 32)
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cin + Cout + 61)
    x = (torch.randn(B, H, W, Cin, generator=g) * 1.5 + 0.3).cuda()          # NHWC, H = time, W = frequency
    gamma = (1 + 0.2 * torch.randn(Cin, generator=g)).cuda()
    beta = (0.2 * torch.randn(Cin, generator=g)).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)).cuda()   # torch OIHW: [o][i][ky = frequency][kx = time]
    b = torch.randn(Cout, generator=g).cuda()
    X = x.permute(0, 3, 2, 1).double()                                        # (B, C, F, T)
    z = F.group_norm(X, G, gamma.double(), beta.double(), eps=1e-6)
    if silu:
        z = F.silu(z)
    ref = F.conv2d(F.interpolate(z, scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1).permute(0, 3, 2, 1)   # (B, 2H, 2W, Cout)
    U = torch.full((256 * Cin * Cout,), float("nan"), device="cuda")
    _lib.check(lib.buddy_conv3_weight_prep(P(w.contiguous()), Cout, Cin, 0, 61, P(U), S()))
    y = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    scratch = torch.empty(64 * B * ((H + 5) // 6) * ((W + 5) // 6) * (Cin + 4 * Cout), device="cuda")
    stats = torch.empty(B, G, 2, device="cuda")
    stat_scratch = torch.empty(B * 256 * 1024 * 2, dtype=torch.float64, device="cuda")
    csum = torch.zeros(B, Cout, 2, dtype=torch.float64, device="cuda")
    _lib.check(lib.buddy_gn_upconv3x3_winograd6(P(x), P(gamma), P(beta), G, silu, P(U), P(b), P(y), P(scratch), P(stats), stat_scratch.data_ptr(),
                                                csum.data_ptr() if stat else None, B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize()
    e = rel(y, ref.float())
    print(f"sub-pixel up conv {B}x{H}x{W} {Cin}->{Cout}: {e:.2e}")
    assert e < 1e-4
    if stat:
        yd = y.double()
        s_ref, q_ref = yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))
        assert float((csum[..., 0] - s_ref).abs().max() / yd.abs().sum(dim=(1, 2)).max()) < 1e-6
        assert float((csum[..., 1] - q_ref).abs().max() / q_ref.max()) < 1e-6


@pytest.mark.parametrize("B,H,W,C,Cout,silu", [(2, 16, 16, 32, 32, 1), (1, 32, 20, 256, 128, 1), (2, 13, 18, 64, 64, 0), (1, 24, 36, 256, 256, 1)])
def test_gnbwd_upconv3x3_winograd6(lib, B, H, W, C, Cout, silu):
    """Backward of the same piece of the up block: gradient w.r.t. the LOW-resolution input u of Conv_0(upsample(u)) for dx = the input-gradient of
    act(GroupNorm_1(h)) under the incoming da (h, da at (2H, 2W)): GroupNorm backward apply + space-to-depth inside the input transform, flipped
    phase kernels (kind 61, dgrad), output at (H, W) -- against fp64 autograd through upsample + conv + GroupNorm.  2e-4 like the plain form."""
    from buddy_amd import _lib
    G = min(C // 4, 32)
    gen = torch.Generator(device="cpu").manual_seed(B * 1000 + C + Cout + 62)
    h = (torch.randn(B, 2 * H, 2 * W, C, generator=gen) * 1.5 + 0.3).cuda()   # Conv_0's output (+ bias), NHWC
    da = torch.randn(B, 2 * H, 2 * W, C, generator=gen).cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=gen)).cuda()
    beta = (0.2 * torch.randn(C, generator=gen)).cuda()
    w = (torch.randn(C, Cout, 3, 3, generator=gen) / np.sqrt(9 * Cout)).cuda()   # Conv_0: Cout (low-res channels) -> C
    hd = h.permute(0, 3, 2, 1).double().requires_grad_(True)
    z = F.group_norm(hd, G, gamma.double(), beta.double(), eps=1e-6)
    a = F.silu(z) if silu else z
    dx, = torch.autograd.grad(a, hd, da.permute(0, 3, 2, 1).double())
    u = torch.zeros(B, Cout, W, H, dtype=torch.float64, device="cuda", requires_grad=True)
    hh = F.conv2d(F.interpolate(u, scale_factor=2, mode="nearest"), w.double(), None, padding=1)
    ref, = torch.autograd.grad(hh, u, dx)
    ref = ref.permute(0, 3, 2, 1)                                               # (B, H, W, Cout)
    hg = h.permute(0, 3, 1, 2).double().reshape(B, G, C // G, 4 * H * W)
    mean = hg.mean(dim=(2, 3)); rstd = 1.0 / torch.sqrt(hg.var(dim=(2, 3), unbiased=False) + 1e-6)
    stats = torch.stack([mean, rstd], dim=-1).float().contiguous()
    U = torch.full((256 * C * Cout,), float("nan"), device="cuda")
    _lib.check(lib.buddy_conv3_weight_prep(P(w.contiguous()), C, Cout, 1, 61, P(U), S()))
    y = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    scratch = torch.empty(64 * B * ((H + 5) // 6) * ((W + 5) // 6) * (4 * C + Cout), device="cuda")
    stat_scratch = torch.empty(B * 256 * C * 2, dtype=torch.float64, device="cuda")
    red = torch.empty(B, G, 2, device="cuda")
    _lib.check(lib.buddy_gnbwd_upconv3x3_winograd6(P(h), P(gamma), P(beta), P(stats), P(da), G, silu, P(U), P(y), P(scratch), stat_scratch.data_ptr(), P(red),
                                                   B, H, W, C, Cout, S()))
    torch.cuda.synchronize()
    e = rel(y, ref.float())
    print(f"gn-bwd + sub-pixel up data-gradient {B}x{H}x{W} {C}->{Cout}: {e:.2e}")
    assert e < 2e-4


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("C,silu", [(32, 1), (96, 1), (128, 0), (384, 1), (512, 1)])
def test_groupnorm_act_fwd_bwd(lib, mode, C, silu):
    from buddy_amd import _lib
    B, H, W = 2, 16, 24
    G = min(C // 4, 32)
    g = torch.Generator(device="cpu").manual_seed(C + mode)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).cuda().double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.2 * torch.randn(C, generator=g)).cuda()
    z = F.group_norm(x, G, gamma.double(), beta.double(), eps=1e-6)
    if silu:
        z = F.silu(z)
    if mode == 1:
        z = z.reshape(B, C, H // 2, 2, W // 2, 2).mean(dim=(3, 5))
    elif mode == 2:
        z = z.repeat_interleave(2, 2).repeat_interleave(2, 3)
    dy = torch.randn(z.shape, generator=g).cuda()
    gx_ref, = torch.autograd.grad(z, x, dy.double())
    x_nhwc = x.detach().float().permute(0, 2, 3, 1).contiguous()
    Ho, Wo = z.shape[2], z.shape[3]
    y = torch.empty(B, Ho, Wo, C, device="cuda")
    stats = torch.empty(B, G, 2, device="cuda")
    red = torch.empty(B, G, 2, device="cuda")
    scratch = torch.empty(B * 256 * C * 4, device="cuda")
    _lib.check(lib.buddy_groupnorm_act(P(x_nhwc), P(gamma), P(beta), P(y), P(stats), P(scratch), B, H, W, C, G, mode, silu, S()))
    torch.cuda.synchronize()
    assert rel(y.permute(0, 3, 1, 2), z.detach().float()) < 1e-5
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous()
    dx = torch.empty(B, H, W, C, device="cuda")
    _lib.check(lib.buddy_groupnorm_act_bwd(P(x_nhwc), P(gamma), P(beta), P(stats), P(dy_nhwc), P(dx), P(scratch), P(red), B, H, W, C, G,
                                           mode, silu, S()))
    torch.cuda.synchronize()
    assert rel(dx.permute(0, 3, 1, 2), gx_ref.float()) < 2e-5


@pytest.mark.parametrize("L,M", [(4096, 100), (16000, 3000), (5000, 1024)])
def test_fir_and_adjoint(lib, L, M):
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(L + M)
    x = torch.randn(2, L, generator=g).cuda()
    h = (torch.randn(M, generator=g) * torch.exp(-torch.arange(M) / (M / 5))).cuda()
    y = torch.empty_like(x)
    _lib.check(lib.buddy_fir(P(x), P(h), 0, P(y), 2, L, M, 0, S()))
    ref = F.conv1d(F.pad(x.double()[:, None], (M - 1, 0)), h.double().flip(0)[None, None])[:, 0].float()
    torch.cuda.synchronize()
    assert rel(y, ref) < 1e-5
    gy = torch.randn(2, L, generator=g).cuda()
    gx = torch.empty_like(x)
    _lib.check(lib.buddy_fir(P(gy), P(h), 0, P(gx), 2, L, M, 1, S()))
    torch.cuda.synchronize()
    # <y, gy> == <x, gx>
    lhs, rhs = float((ref.double() * gy.double()).sum()), float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_row_ops(lib):
    from buddy_amd import _lib
    x = torch.randn(3, 5001, device="cuda")
    y = torch.randn(3, 5001, device="cuda")
    a = torch.tensor([0.5, -1.0, 2.0], device="cuda")
    c = torch.tensor([1.5, 0.25, -3.0], device="cuda")
    out = torch.empty_like(x)
    _lib.check(lib.buddy_axpby_rows(P(x), P(y), P(a), P(c), P(out), 3, 5001, S()))
    mom = torch.empty(3, 2, device="cuda", dtype=torch.float64)
    _lib.check(lib.buddy_row_moments(P(x), P(mom), 3, 5001, S()))
    torch.cuda.synchronize()
    assert rel(out, a[:, None] * x + c[:, None] * y) < 1e-6
    assert rel(mom[:, 0], x.double().sum(1)) < 1e-9 and rel(mom[:, 1], (x.double() ** 2).sum(1)) < 1e-9


def test_wpe_kernel_vs_torch_restatement(lib):
    """buddy_wpe (one workgroup per frequency bin, Cholesky in LDS, complex128) vs the torch restatement of the same algorithm."""
    from oracle.batched import wpe
    from buddy_amd.utils.wpe import wpe_hip
    g = torch.Generator(device="cpu").manual_seed(11)
    L = 16000
    src = torch.randn(L, generator=g, dtype=torch.float64) * (torch.rand(L, generator=g, dtype=torch.float64) > 0.7)
    h = torch.randn(3000, generator=g, dtype=torch.float64) * torch.exp(-torch.arange(3000, dtype=torch.float64) / 600.0)
    y = torch.nn.functional.conv1d(src.view(1, 1, -1), h.flip(0).view(1, 1, -1), padding=2999)[0, 0, :L]
    Y = wpe.stft(y[None].cuda()).permute(2, 0, 1).contiguous()              # (F, 1, T)
    Zt = wpe.wpe(Y, taps=50, delay=2, iterations=5)
    Zh = wpe_hip(Y, taps=50, delay=2, iterations=5)
    err = float((torch.view_as_real(Zh) - torch.view_as_real(Zt)).abs().max() / torch.view_as_real(Zt).abs().max())
    # 50 taps on a reverberant signal: cond(R) ~ 1e9..1e10, so Cholesky (kernel) and LU (torch.linalg.solve) agree to cond * eps_fp64
    assert err < 1e-4, err
    # fewer taps / iterations, odd delay
    Zt = wpe.wpe(Y, taps=7, delay=3, iterations=2)
    Zh = wpe_hip(Y, taps=7, delay=3, iterations=2)
    assert float((torch.view_as_real(Zh) - torch.view_as_real(Zt)).abs().max() / torch.view_as_real(Zt).abs().max()) < 1e-9


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("B,T,Cc", [(2, 2048, 256), (3, 144, 64), (1, 320, 128), (2, 1000, 256)])
def test_flash_attention_fwd_bwd(lib, B, T, Cc, prec):
    """online-softmax attention (no T x T matrix) vs the reference formulation in fp64: w = softmax(q k^T C^-1/2), h = w v
    (networks/ncsnpp_utils/layerspp.py:82-86) and its three input gradients; ragged T (not a multiple of the 64-row / 32-column blocks).
    prec 0 = fp32 operands (the product path): 1e-5 / 2e-5.  prec 1 / 2 = bf16 / f16 MFMA operands (opt-in fast mode): operand rounding
    2^-9 / 2^-12 -> stated 2e-2 / 3e-3 of the tensor's abs-max."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T + Cc)
    q, k, v, dO = (torch.randn(B, T, Cc, generator=g).cuda() for _ in range(4))
    q = q * 1.5                                                   # some peaky rows
    scale = Cc ** -0.5
    O = torch.empty_like(q); lse = torch.empty(B, T, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    delta = torch.empty(B, T, device="cuda")
    if prec == 0:
        _lib.check(lib.buddy_flash_attention_fwd(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, 0, S()))
        _lib.check(lib.buddy_flash_attention_bwd(P(q), P(k), P(v), P(O), P(dO), P(lse), P(delta), P(dq), P(dk), P(dv), B, T, Cc, scale, 0, S()))
        assert lib.buddy_flash_attention_fwd(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, 1, S()) != 0       # 16-bit operands: the entries below
    else:
        ws = torch.empty(lib.buddy_flash_attention16_workspace(B, T, Cc), device="cuda")
        _lib.check(lib.buddy_flash_attention16_fwd(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, prec, P(ws), S()))
        _lib.check(lib.buddy_flash_attention16_bwd(P(q), P(k), P(v), P(O), P(dO), P(lse), P(delta), P(dq), P(dk), P(dv), B, T, Cc, scale, prec, P(ws), S()))
        assert lib.buddy_flash_attention16_fwd(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, prec, None, S()) != 0   # no workspace: argument error
    torch.cuda.synchronize()
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    w = torch.softmax(torch.einsum("bic,bjc->bij", qd, kd) * scale, dim=-1)
    ref = torch.einsum("bij,bjc->bic", w, vd)
    gq, gk, gv = torch.autograd.grad(ref, (qd, kd, vd), dO.double())
    lse_ref = torch.logsumexp(torch.einsum("bic,bjc->bij", qd, kd) * scale, dim=-1)
    tf, tb, tl = [(1e-5, 2e-5, 1e-4), (2e-2, 2e-2, 5e-2), (3e-3, 3e-3, 1e-2)][prec]
    errs = dict(O=rel(O, ref.detach()), lse=float((lse.double() - lse_ref.detach()).abs().max()), dq=rel(dq, gq), dk=rel(dk, gk), dv=rel(dv, gv))
    print(prec, (B, T, Cc), {k_: f"{v_:.1e}" for k_, v_ in errs.items()})
    assert errs["O"] < tf and errs["lse"] < tl
    assert errs["dq"] < tb and errs["dk"] < tb and errs["dv"] < tb


@pytest.mark.parametrize("B,T,Cc,splits", [(1, 2048, 256, 8), (2, 2048, 256, 4), (1, 1000, 256, 5), (3, 144, 64, 2), (1, 320, 128, 10), (2, 2048, 256, 0)])
def test_flash_attention_split_loops(lib, B, T, Cc, splits):
    """the split form of the fp32 attention kernels (sequential key / query loop spread over `splits` workgroups per row block, partials combined in
    fixed order) against the reference formulation in fp64 at the bounds of the unsplit kernels, and against the unsplit kernels themselves; ragged T,
    uneven last split, the count the network picks (splits = 0); invalid counts are refused."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T + Cc + 7)
    q, k, v, dO = (torch.randn(B, T, Cc, generator=g).cuda() for _ in range(4))
    q = q * 1.5
    scale = Cc ** -0.5
    if splits == 0:
        splits = lib.buddy_flash_attention_splits(B, T)
        assert splits == 8
    # a function of T alone (the summation order of a row must not depend on the batch it is in)
    assert lib.buddy_flash_attention_splits(1, 2048) == 8 and lib.buddy_flash_attention_splits(8, 2048) == 8 and lib.buddy_flash_attention_splits(4, 15040) == 1
    ws = torch.empty(lib.buddy_flash_attention_workspace(B, T, Cc, splits), device="cuda")
    out = {}
    for name, ns in (("split", splits), ("one", 1)):
        O = torch.empty_like(q); lse = torch.empty(B, T, device="cuda")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        delta = torch.empty(B, T, device="cuda")
        _lib.check(lib.buddy_flash_attention_fwd_split(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, ns, P(ws), S()))
        _lib.check(lib.buddy_flash_attention_bwd_split(P(q), P(k), P(v), P(O), P(dO), P(lse), P(delta), P(dq), P(dk), P(dv), B, T, Cc, scale, ns, P(ws), S()))
        torch.cuda.synchronize()
        out[name] = (O, lse, dq, dk, dv)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    sc = torch.einsum("bic,bjc->bij", qd, kd) * scale
    ref = torch.einsum("bij,bjc->bic", torch.softmax(sc, dim=-1), vd)
    gq, gk, gv = torch.autograd.grad(ref, (qd, kd, vd), dO.double())
    O, lse, dq, dk, dv = out["split"]
    errs = dict(O=rel(O, ref.detach()), lse=float((lse.double() - torch.logsumexp(sc, dim=-1).detach()).abs().max()), dq=rel(dq, gq), dk=rel(dk, gk), dv=rel(dv, gv))
    print((B, T, Cc, splits), {k_: f"{v_:.1e}" for k_, v_ in errs.items()})
    assert errs["O"] < 1e-5 and errs["lse"] < 1e-4 and errs["dq"] < 2e-5 and errs["dk"] < 2e-5 and errs["dv"] < 2e-5
    for a, b in zip(out["split"], out["one"]):
        assert rel(a, b.double()) < 2e-5
    # a split that would own no block, or a missing workspace, is an argument error
    assert lib.buddy_flash_attention_fwd_split(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, (T + 31) // 32 + 1, P(ws), S()) != 0
    assert lib.buddy_flash_attention_fwd_split(P(q), P(k), P(v), P(O), P(lse), B, T, Cc, scale, 2, None, S()) != 0


def test_fir_resample2_vs_reference(lib, golden):
    """upsample_2d / downsample_2d with the (1,3,3,1) kernel (reference up_or_down_sampling.py:195-257 -> upfirdn2d) and their transposes
    against vectors recorded from the reference's pure-PyTorch upfirdn2d; NHWC, C = 6 (scalar path) and via the adjoint identities."""
    from buddy_amd import _lib
    g = golden("fir_ops")
    nhwc = lambda a: torch.from_numpy(a).permute(0, 2, 3, 1).contiguous().cuda()
    x = nhwc(g["x"]); B, H, W, Cc = x.shape
    up = torch.empty(B, 2 * H, 2 * W, Cc, device="cuda"); dn = torch.empty(B, H // 2, W // 2, Cc, device="cuda")
    _lib.check(lib.buddy_fir_resample2(P(x), P(up), B, H, W, Cc, 1, 1.0, 0, S()))
    _lib.check(lib.buddy_fir_resample2(P(x), P(dn), B, H, W, Cc, 0, 1.0, 0, S()))
    assert rel(up, nhwc(g["up"])) < 1e-6 and rel(dn, nhwc(g["down"])) < 1e-6
    # VJPs: up^T = 4 down, down^T = up / 4; the second call accumulates on top of the first
    gu = torch.empty_like(x); cu, cd = nhwc(g["cot_up"]), nhwc(g["cot_down"])
    _lib.check(lib.buddy_fir_resample2(P(cu), P(gu), B, 2 * H, 2 * W, Cc, 0, 4.0, 0, S()))
    assert rel(gu, nhwc(g["vjp_up"])) < 1e-6
    _lib.check(lib.buddy_fir_resample2(P(cd), P(gu), B, H // 2, W // 2, Cc, 1, 0.25, 1, S()))
    assert rel(gu, nhwc(g["vjp_up"]) + nhwc(g["vjp_down"])) < 1e-6


@pytest.mark.parametrize("O,I", [(128, 128), (256, 384), (32, 64)])
@pytest.mark.parametrize("kind", [0, 2, 4, 6])
@pytest.mark.parametrize("dgrad", [0, 1])
def test_conv3_weight_prep_vs_host_restatement(lib, O, I, kind, dgrad):
    """buddy_conv3_weight_prep (csrc/wprep.hip: raw torch OIHW -> operand form of one kernel variant, on the device) against the host
    restatement the library exports for the unit tests (buddy_winograd{,4,6}_transform_weights on the tap-major packing): bit for bit --
    both evaluate G g G^T in un-contracted fp64 and round once."""
    from buddy_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(O * 7 + I + kind + dgrad)
    w = torch.randn(O, I, 3, 3, generator=g) * torch.logspace(-2, 1, O)[:, None, None, None]
    # tap-major packing of THIS convolution: forward [o][(dy*3+dx)*I + i] = w[o][i][ky=dx][kx=dy]; dgrad [i][(dy*3+dx)*O + o], taps flipped
    wt = w.permute(0, 3, 2, 1) if not dgrad else w.flip(2, 3).permute(1, 3, 2, 0)     # [co][dy][dx][ci]
    Co, Ci = (O, I) if not dgrad else (I, O)
    wt = wt.contiguous().reshape(Co, 9 * Ci)
    n = {0: 9, 2: 16, 4: 36, 6: 64}[kind] * O * I
    out = torch.full((n,), float("nan"), device="cuda")
    _lib.check(lib.buddy_conv3_weight_prep(P(w.cuda().contiguous()), O, I, dgrad, kind, P(out), S()))
    torch.cuda.synchronize()
    if kind == 0:
        ref = wt.reshape(-1)
    else:
        ref = torch.empty(n)
        fn = {2: lib.buddy_winograd_transform_weights, 4: lib.buddy_winograd4_transform_weights, 6: lib.buddy_winograd6_transform_weights}[kind]
        _lib.check(fn(wt.numpy().ctypes.data, Co, Ci, ref.numpy().ctypes.data))
    assert torch.equal(out.cpu(), ref)
