"""Helpers for the -m gpu parity tests (call librangeldm_hip through its C ABI)."""
import ctypes as C

import numpy as np
import torch

from rangeldm_amd import _lib


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def hip_conv(x0, weight, bias, x1=None, stride=1, pad_mode=0, upsample=False, gamma=None, beta=None, silu=False,
             eps=1e-5, temb=None, res=None):
    """rldm_test_conv: fp32 NCHW tensors in/out, the kernel under test in the middle."""
    dev = torch.device("cuda")
    d = _lib.ConvDescC()
    B, C0, W, H = x0.shape
    d.B, d.Cin0, d.Win, d.Hin = B, C0, W, H
    d.Cin1 = 0 if x1 is None else x1.shape[1]
    d.Cout, d.ksize = weight.shape[0], weight.shape[2]
    d.stride, d.pad_mode, d.upsample = stride, pad_mode, 1 if upsample else 0
    d.gn, d.silu, d.eps = (1 if gamma is not None else 0), (1 if silu else 0), eps
    up = 2 if upsample else 1
    Wo, Ho = W * up // stride, H * up // stride
    y = torch.empty((B, d.Cout, Wo, Ho), device=dev, dtype=torch.float32)

    def devp(t):
        if t is None:
            return None, None
        t = t.to(dev, torch.float32).contiguous()
        return t, C.c_void_p(t.data_ptr())

    def hostp(t):
        if t is None:
            return None, None
        a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        return a, a.ctypes.data_as(C.c_void_p)

    k0, p0 = devp(x0)
    k1, p1 = devp(x1)
    kr, pr = devp(res)
    hw, pw = hostp(weight)
    hb, pb = hostp(bias)
    hg, pg = hostp(gamma)
    hbt, pbt = hostp(beta)
    ht, pt = hostp(temb)
    _lib.check(_lib.lib().rldm_test_conv(C.byref(d), p0, p1, pw, pb, pg, pbt, pt, pr, C.c_void_p(y.data_ptr()),
                                         _lib.stream_ptr(dev)), "rldm_test_conv")
    torch.cuda.synchronize()
    return y.cpu()


def hip_attention(qkv, C_):
    dev = torch.device("cuda")
    B, L, _ = qkv.shape
    q = qkv.to(dev, torch.float32).contiguous()
    out = torch.empty((B, L, C_), device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().rldm_test_attention(C.c_void_p(q.data_ptr()), B, L, C_, C.c_void_p(out.data_ptr()),
                                              _lib.stream_ptr(dev)), "rldm_test_attention")
    torch.cuda.synchronize()
    return out.cpu()


def hip_conv_stats(x0, weight, bias, stride=1):
    """rldm_test_conv_stats: per-image per-channel (sum, sumsq) the conv epilogue emitted for its bf16 output."""
    dev = torch.device("cuda")
    d = _lib.ConvDescC()
    B, C0, W, H = x0.shape
    d.B, d.Cin0, d.Win, d.Hin, d.Cin1 = B, C0, W, H, 0
    d.Cout, d.ksize = weight.shape[0], weight.shape[2]
    d.stride, d.pad_mode, d.upsample, d.gn, d.silu, d.eps = stride, 0, 0, 0, 0, 1e-5
    xs = x0.to(dev, torch.float32).contiguous()
    hw = np.ascontiguousarray(weight.numpy(), dtype=np.float32)
    hb = np.ascontiguousarray(bias.numpy(), dtype=np.float32)
    st = torch.empty((B, d.Cout, 2), device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().rldm_test_conv_stats(C.byref(d), C.c_void_p(xs.data_ptr()), hw.ctypes.data_as(C.c_void_p),
                                               hb.ctypes.data_as(C.c_void_p), C.c_void_p(st.data_ptr()),
                                               _lib.stream_ptr(dev)), "rldm_test_conv_stats")
    torch.cuda.synchronize()
    return st.cpu()


def hip_attention_qkv(x, gamma, beta, wqkv, bqkv, groups=32, eps=1e-5):
    """rldm_test_attention_qkv: the fused GroupNorm -> q/k/v -> softmax.V launch of every UNet attention block.
    x (B, L, C) token-major fp32 -> (B, L, C) heads concatenated (before to_out)."""
    dev = torch.device("cuda")
    B, L, C_ = x.shape
    xs = x.to(dev, torch.float32).contiguous()
    out = torch.empty((B, L, C_), device=dev, dtype=torch.float32)

    def hostp(t):
        a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        return a, a.ctypes.data_as(C.c_void_p)

    hg, pg = hostp(gamma)
    hb, pb = hostp(beta)
    hw, pw = hostp(wqkv)
    hq, pq = hostp(bqkv)
    _lib.check(_lib.lib().rldm_test_attention_qkv(C.c_void_p(xs.data_ptr()), B, L, C_, groups, eps, pg, pb, pw, pq,
                                                  C.c_void_p(out.data_ptr()), _lib.stream_ptr(dev)),
               "rldm_test_attention_qkv")
    torch.cuda.synchronize()
    return out.cpu()
