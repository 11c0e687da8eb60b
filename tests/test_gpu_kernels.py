"""-m gpu: the MFMA GEMM families on real gfx950 against float64 NumPy.  These pin the
MFMA lane layouts that the CPU-side hipsim interpreter only assumes."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from latex_ocr_amd import _abi
    return _abi.load()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _bf(a):
    return torch.from_numpy(a).to(torch.bfloat16)


@pytest.mark.parametrize("dt,a_f32,c_f32,small,M,N,K,act", [
    (0, 1, 1, 0, 300, 200, 96, 1), (0, 1, 1, 1, 64, 160, 128, 2),
    (1, 0, 0, 0, 257, 130, 160, 1), (1, 0, 1, 0, 129, 96, 64, 0),
    (1, 1, 1, 0, 200, 500, 512, 0), (1, 1, 1, 1, 64, 2048, 1024, 2),
])
def test_gemm_nt(dt, a_f32, c_f32, small, M, N, K, act):
    L = _lib()
    rng = np.random.default_rng(M + N + K)
    # asymmetric data: a transposed / permuted fragment cannot pass
    A = (rng.standard_normal((M, K)) + np.arange(K)[None, :] * 0.01).astype(np.float32)
    B = (rng.standard_normal((N, K)) + np.arange(N)[:, None] * 0.003).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    if dt == 1:
        Bd = _bf(B).cuda(); Bref = Bd.float().cpu().numpy()
        if a_f32:
            Ad = _dev(A); Aref = _bf(A).float().numpy()
        else:
            Ad = _bf(A).cuda(); Aref = Ad.float().cpu().numpy()
    else:
        Ad, Bd, Aref, Bref = _dev(A), _dev(B), A, B
    out_f32 = dt == 0 or c_f32
    C = torch.zeros(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.lxo_gemm_nt(dt, a_f32, c_f32, small, _p(Ad), _p(Bd), _p(C), M, N, K, K, K, N, _p(_dev(bias)), act,
                       ctypes.c_float(0.25), 0, st)
    assert rc == 0, L.lxo_last_error()
    torch.cuda.synchronize()
    ref = 0.25 * (Aref.astype(np.float64) @ Bref.astype(np.float64).T) + bias
    ref = np.maximum(ref, 0) if act == 1 else (np.tanh(ref) if act == 2 else ref)
    err = np.abs(C.float().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
    tol = 2e-5 if out_f32 else 8e-3
    assert err < tol, err


@pytest.mark.parametrize("dt,a_f32,b_f32,M,I,J,nsplit", [
    (0, 1, 1, 500, 136, 200, 3), (1, 0, 0, 1000, 576, 128, 4), (1, 1, 0, 333, 512, 504, 2),
    (1, 1, 1, 64, 1024, 2048, 1), (1, 0, 1, 640, 96, 2048, 2),
])
def test_gemm_tn(dt, a_f32, b_f32, M, I, J, nsplit):
    L = _lib()
    rng = np.random.default_rng(M + I + J)
    A = (rng.standard_normal((M, I)) + np.arange(I)[None, :] * 0.002).astype(np.float32)
    B = (rng.standard_normal((M, J)) + np.arange(M)[:, None] * 0.001).astype(np.float32)
    if dt == 1:
        Ad = _dev(A) if a_f32 else _bf(A).cuda()
        Bd = _dev(B) if b_f32 else _bf(B).cuda()
        Aref, Bref = _bf(A).float().numpy(), _bf(B).float().numpy()
    else:
        Ad, Bd, Aref, Bref = _dev(A), _dev(B), A, B
    C0 = rng.standard_normal((I, J)).astype(np.float32)
    C = _dev(C0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.lxo_gemm_tn(dt, a_f32, b_f32, _p(Ad), _p(Bd), _p(C), M, I, J, I, J, J, nsplit, 1, st)
    assert rc == 0, L.lxo_last_error()
    torch.cuda.synchronize()
    ref = C0 + Aref.astype(np.float64).T @ Bref.astype(np.float64)
    err = np.abs(C.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("B,H,W,Cin,Cout,valid", [(3, 10, 70, 64, 128, 0), (2, 16, 64, 128, 256, 1), (2, 7, 130, 256, 64, 0), (1, 32, 128, 64, 128, 0)])
def test_conv3x3_fwd_and_wgrad_bf16(B, H, W, Cin, Cout, valid):
    """implicit-GEMM conv (halo tiles, LDS-DMA) and the tap-reuse weight gradient (transposing LDS reads)
    against torch conv2d on the same bf16-rounded operands."""
    import torch.nn.functional as F
    L = _lib()
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(3, 3, Cin, Cout, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = (H - 2, W - 2) if valid else (H, W)
    pad = 0 if valid else 1
    wpk = w.reshape(9 * Cin, Cout).t().contiguous()                       # [Cout][9*Cin]
    xd, wd, bd = x.cuda(), wpk.cuda(), bias.cuda()
    y = torch.zeros(B, Ho, Wo, Cout, dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.lxo_conv3x3(1, _p(xd), _p(wd), _p(bd), _p(y), B, H, W, Cin, Ho, Wo, Cout, pad, 1, st) == 0, L.lxo_last_error()
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.float().permute(3, 2, 0, 1).double(), bias.double(), padding=pad))
    ref = ref.permute(0, 2, 3, 1).float()
    torch.cuda.synchronize()
    err = (y.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 6e-3, err
    dy = torch.randn(B, Ho, Wo, Cout, generator=g).to(torch.bfloat16)
    dw = torch.zeros(9 * Cin, Cout, device="cuda")
    assert L.lxo_conv3x3_wgrad(1, _p(xd), _p(dy.cuda()), _p(dw), B, H, W, Cin, Ho, Wo, Cout, pad, st) == 0, L.lxo_last_error()
    wr = w.float().permute(3, 2, 0, 1).double().requires_grad_(True)
    out = F.conv2d(x.float().permute(0, 3, 1, 2).double(), wr, None, padding=pad)
    (out * dy.float().permute(0, 3, 1, 2).double()).sum().backward()
    refw = wr.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout).float()       # HWIO flattened
    torch.cuda.synchronize()
    errw = (dw.cpu() - refw).abs().max().item() / refw.abs().max().item()
    assert errw < 2e-5, errw


@pytest.mark.parametrize("B,H,W,Cin,Cout,pad", [(2, 16, 64, 256, 256, 1), (3, 14, 62, 128, 512, 2), (2, 9, 37, 64, 256, 0), (2, 12, 70, 128, 128, 1), (2, 20, 66, 128, 64, 1)])
def test_conv3x3_ex_full_epilogue_bf16(B, H, W, Cin, Cout, pad):
    """fused epilogue of the conv kernels (bias, ReLU, pre-addend copy, f32 addend, ReLU mask, column sums) on the
    8x32x256 (Cout % 256 == 0) and 4x64x128 halo tiles; pad 2 = the dgrad geometry of the VALID layer."""
    import torch.nn.functional as F
    L = _lib()
    g = torch.Generator().manual_seed(Cin + Cout + H + pad)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(3, 3, Cin, Cout, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    addend = torch.randn(Ho * Wo, Cout, generator=g)
    mask = torch.randn(B, Ho, Wo, Cout, generator=g).to(torch.bfloat16)
    wpk = w.reshape(9 * Cin, Cout).t().contiguous()
    out = torch.zeros(B, Ho, Wo, Cout, dtype=torch.bfloat16, device="cuda"); pre = torch.zeros_like(out)
    cs = torch.zeros(Cout, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd, wd, bd, ad, md = x.cuda(), wpk.cuda(), bias.cuda(), addend.cuda(), mask.cuda()
    rc = L.lxo_conv3x3_ex(1, _p(xd), _p(wd), _p(bd), _p(out), B, H, W, Cin, Ho, Wo, Cout, pad, 1, _p(ad), Ho * Wo, _p(pre), _p(md), _p(cs), st)
    assert rc == 0, L.lxo_last_error()
    torch.cuda.synchronize()
    y = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.float().permute(3, 2, 0, 1).double(), bias.double(), padding=pad))
    y = y.permute(0, 2, 3, 1)
    rpre = y.clone()
    y = (y.reshape(B, Ho * Wo, Cout) + addend.double()[None]).reshape(B, Ho, Wo, Cout)
    y = torch.where(mask.double() > 0, y, torch.zeros_like(y))
    s = y.abs().max().item()
    assert (pre.float().cpu().double() - rpre).abs().max().item() / s < 6e-3
    assert (out.float().cpu().double() - y).abs().max().item() / s < 6e-3
    rcs = y.reshape(-1, Cout).sum(0)
    assert (cs.cpu().double() - rcs).abs().max().item() / rcs.abs().max().item() < 2e-4


@pytest.mark.parametrize("dt,nv,nimg,R,E,C", [(1, 64, 64, 868, 256, 512), (1, 6, 2, 84, 256, 512), (1, 3, 3, 37, 256, 512),
                                              (1, 4, 4, 100, 128, 256), (0, 5, 5, 84, 256, 512)])
def test_attention_fwd_vs_float64(dt, nv, nimg, R, E, C):
    """AttentionMechanism.context (attention_mechanism.py:46-94): scores, softmax over regions, context -- the LDS-DMA
    stream kernel (bf16, E=256, C=512), the register-staged kernel (other widths / f32) and the chunk combine."""
    L = _lib()
    g = torch.Generator().manual_seed(R + nv)
    beam = nv // nimg
    cdt = torch.bfloat16 if dt == 1 else torch.float32
    att_img = torch.randn(nimg, R, E, generator=g).to(cdt); img = torch.randn(nimg, R, C, generator=g).to(cdt)
    att_h = torch.randn(nv, E, generator=g); beta = torch.randn(E, generator=g) * 0.3
    Rp = (R + 7) // 8 * 8
    alpha = torch.zeros(nv, Rp, device="cuda"); part = torch.zeros(nv * 32 * (C + 2), device="cuda"); ctx = torch.zeros(nv, C, device="cuda")
    a_d, i_d, h_d, b_d = att_img.cuda(), img.cuda(), att_h.cuda(), beta.cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.lxo_attention_fwd(dt, _p(a_d), _p(i_d), _p(h_d), _p(b_d), _p(alpha), _p(part), _p(ctx), C, nv, R, E, C, beam, st)
    assert rc == 0, L.lxo_last_error()
    torch.cuda.synchronize()
    idx = torch.arange(nv) // beam
    e = (torch.tanh(att_img.double()[idx] + att_h.double()[:, None, :]) * beta.double()).sum(-1)
    a = torch.softmax(e, dim=-1)
    c = (a[:, :, None] * img.double()[idx]).sum(1)
    tol = 2e-5 if dt == 1 else 2e-6          # bf16 mode: fast tanh (1 - 2 rcp(exp(2x)+1)) on exact bf16 inputs
    assert (alpha[:, :R].cpu().double() - a).abs().max().item() < tol
    assert (ctx.cpu().double() - c).abs().max().item() < 50 * tol
