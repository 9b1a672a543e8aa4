"""Per-op wrappers of the training-side C ABI entry points (K11/K12).  torch tensors are HBM containers only."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .net import Context, _stream_ptr


def _ctx(t):
    if not t.is_cuda:
        raise _lib.OdError("training ops need device tensors (MI355X); there is no CPU path")
    return Context.get(t.device)


def bn_stats(z, gamma, beta, eps=1e-3, run_mean=None, run_var=None, momentum=0.99):
    """z f16 [..., C] -> mean, rstd, scale, shift (f32 [C])"""
    ctx = _ctx(z)
    Cc = z.shape[-1]
    M = z.numel() // Cc
    out = [torch.empty(Cc, dtype=torch.float32, device=z.device) for _ in range(4)]
    wsb = ctx.lib.od_bn_workspace_bytes(M, Cc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
    _lib.check(ctx.lib.od_bn_stats(ctx.handle, z.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                   out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                   run_mean.data_ptr() if run_mean is not None else None,
                                   run_var.data_ptr() if run_var is not None else None, float(momentum),
                                   ws.data_ptr(), wsb, _stream_ptr()), "od_bn_stats")
    return out


def scale_act(z, scale, shift, act=None, alpha=0.0, res=None, res_mode="none"):
    ctx = _ctx(z)
    B, H, W, Cc = z.shape
    y = torch.empty_like(z)
    rm = {"none": 0, "same": 1, "up2": 2}[res_mode]
    _lib.check(ctx.lib.od_scale_act(ctx.handle, z.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                    res.data_ptr() if res is not None else None, rm, y.data_ptr(), B, H, W, Cc,
                                    _lib.ACT_ENUM[act], float(alpha), _stream_ptr()), "od_scale_act")
    return y


def bn_bwd(z, dy, scale, shift, mean, rstd, act=None, alpha=0.0, bn=True, dgamma=None, dbeta=None):
    ctx = _ctx(z)
    Cc = z.shape[-1]
    M = z.numel() // Cc
    if dgamma is None:
        dgamma = torch.zeros(Cc, dtype=torch.float32, device=z.device)
        dbeta = torch.zeros(Cc, dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z)
    wsb = ctx.lib.od_bn_workspace_bytes(M, Cc) + 2 * Cc * 4
    ws = torch.empty(wsb, dtype=torch.uint8, device=z.device)
    _lib.check(ctx.lib.od_bn_bwd(ctx.handle, z.data_ptr(), dy.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                 mean.data_ptr() if mean is not None else None,
                                 rstd.data_ptr() if rstd is not None else None, M, Cc, _lib.ACT_ENUM[act], float(alpha),
                                 int(bn), dgamma.data_ptr(), dbeta.data_ptr(), dz.data_ptr(), ws.data_ptr(), wsb,
                                 _stream_ptr()), "od_bn_bwd")
    return dz, dgamma, dbeta


def conv_bwd_weight(x, dz, Cin, Cout, ksize, stride, dw=None):
    """x f16 [B,H,W,Cin], dz f16 [B,Ho,Wo,Cout] -> dw f32 [Cout, k*k*Cin] dense (accumulated into `dw` when given)"""
    ctx = _ctx(x)
    B, H, W, _ = x.shape
    if dw is None:
        dw = torch.zeros((Cout, ksize * ksize * Cin), dtype=torch.float32, device=x.device)
    _lib.check(ctx.lib.od_conv2d_bwd_weight(ctx.handle, x.data_ptr(), dz.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout,
                                            ksize, stride, _stream_ptr()), "od_conv2d_bwd_weight")
    return dw


def pack_weights(w_master, Cout, Cin, ksize, want_bwd=True):
    """w_master f32 [Cout, k*k*Cin] (device) -> (w_fwd f16 [Cout_pad,Kpad], w_bwd f16 [Cin_pad,Kpad_t] or None)"""
    ctx = _ctx(w_master)
    cout_pad, kpad = _lib.conv_weight_dims(Cout, Cin, ksize)
    wf = torch.zeros((cout_pad, kpad), dtype=torch.float16, device=w_master.device)
    wb = None
    if want_bwd:
        cin_pad, kpad_t = _lib.conv_weight_dims(Cin, Cout, ksize)
        wb = torch.zeros((cin_pad, kpad_t), dtype=torch.float16, device=w_master.device)
    _lib.check(ctx.lib.od_pack_weights(ctx.handle, w_master.data_ptr(), wf.data_ptr(),
                                       wb.data_ptr() if wb is not None else None, Cout, Cin, ksize, _stream_ptr()),
               "od_pack_weights")
    return wf, wb


def conv_packed(x, w_packed, scale, bias, Cin, Cout, ksize, stride=1, act=None, alpha=0.0, res=None, res_mode="none",
                transposed=False, out=None, tile_cfg=-1):
    """od_conv2d_fwd on an already packed device weight tensor (forward, or backward-data with the w_bwd pack)."""
    ctx = _ctx(x)
    B, H, W, _ = x.shape
    if transposed:
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=x.device)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.bias, d.out = x.data_ptr(), w_packed.data_ptr(), scale.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.res = res.data_ptr() if res is not None else None
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, Cout
    d.ksize, d.stride = ksize, stride
    d.act, d.alpha = _lib.ACT_ENUM[act], float(alpha)
    d.res_mode = {"none": 0, "same": 1, "up2": 2}[res_mode]
    d.out_dtype = _lib.OD_DT_F16
    d.tile_cfg = tile_cfg
    d.transposed = int(transposed)
    _lib.check(ctx.lib.od_conv2d_fwd(ctx.handle, C.byref(d), _stream_ptr()), "od_conv2d_fwd")
    return out


def down2_sum_add(d, dup=None):
    ctx = _ctx(d)
    B, H, W, Cc = d.shape
    acc = dup is not None
    if dup is None:
        dup = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.float16, device=d.device)
    _lib.check(ctx.lib.od_down2_sum_add(ctx.handle, d.data_ptr(), dup.data_ptr(), B, H // 2, W // 2, Cc, int(acc),
                                        _stream_ptr()), "od_down2_sum_add")
    return dup


def sgd_step(w, m, g, lr, momentum=0.9, weight_decay=0.0, inv_loss_scale=1.0):
    ctx = _ctx(w)
    _lib.check(ctx.lib.od_sgd_step(ctx.handle, w.data_ptr(), m.data_ptr(), g.data_ptr(), w.numel(), float(lr),
                                   float(momentum), float(weight_decay), float(inv_loss_scale), _stream_ptr()),
               "od_sgd_step")
