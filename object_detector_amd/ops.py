"""Thin per-op wrappers over the C ABI (torch tensors in / out as HBM containers).  Used by the parity tests and
by anything that wants a single kernel rather than the whole plan.  No op here has a CPU implementation."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .net import Context, _stream_ptr, pack_conv_weight, pack_first_weight, pad_vec


def _ctx(t: torch.Tensor) -> Context:
    if not t.is_cuda:
        raise _lib.OdError("object_detector_amd ops need device tensors (MI355X); there is no CPU path")
    return Context.get(t.device)


def conv2d(x: torch.Tensor, w_ohwi: np.ndarray, scale: np.ndarray, bias: np.ndarray, stride=1, act=None, alpha=0.0,
           res: torch.Tensor | None = None, res_mode="none", out_f32=False, tile_cfg=-1, out=None,
           out_batch_stride=0, out_pix_stride=0, splitk=1, splitk_ws=None, next_pointwise=None):
    """x f16 [B,H,W,Cin] (device) ; w [Cout,k,k,Cin] host array -> out f16/f32 [B,Ho,Wo,Cout].
    next_pointwise = (w2 [Cout2,1,1,Cout], scale2, bias2, act2, alpha2): the 1x1 layer that consumes `out`
    (od_conv_desc.w2) -> returns (out, out2)."""
    ctx = _ctx(x)
    assert x.dtype == torch.float16 and x.is_contiguous()
    B, H, Wd, Cin = x.shape
    Cout, k = w_ohwi.shape[0], w_ohwi.shape[1]
    wp = torch.from_numpy(pack_conv_weight(w_ohwi)).to(x.device)
    npad = wp.shape[0]
    sc = torch.from_numpy(pad_vec(np.asarray(scale, np.float32), npad)).to(x.device)
    bi = torch.from_numpy(pad_vec(np.asarray(bias, np.float32), npad)).to(x.device)
    Ho, Wo = (H + stride - 1) // stride, (Wd + stride - 1) // stride
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.bias, d.out = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), out.data_ptr()
    d.res = res.data_ptr() if res is not None else None
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, Wd, Cin, Cout
    d.ksize, d.stride = k, stride
    d.act, d.alpha = _lib.ACT_ENUM[act], float(alpha)
    d.res_mode = {"none": _lib.OD_RES_NONE, "same": _lib.OD_RES_SAME, "up2": _lib.OD_RES_UP2}[res_mode]
    d.out_dtype = _lib.OD_DT_F32 if out_f32 else _lib.OD_DT_F16
    d.out_batch_stride, d.out_pix_stride = out_batch_stride, out_pix_stride
    d.tile_cfg = tile_cfg
    if splitk != 1:
        if splitk_ws is None:
            splitk_ws = torch.empty(32 * B * Ho * Wo * Cout, dtype=torch.float32, device=x.device)
        d.splitk, d.splitk_workspace, d.splitk_workspace_bytes = splitk, splitk_ws.data_ptr(), splitk_ws.numel() * 4
    out2 = None
    if next_pointwise is not None:
        w2, scale2, bias2, act2, alpha2 = next_pointwise
        assert w2.shape[1:] == (1, 1, Cout)
        w2p = torch.from_numpy(pack_conv_weight(w2)).to(x.device)
        sc2 = torch.from_numpy(pad_vec(np.asarray(scale2, np.float32), w2p.shape[0])).to(x.device)
        bi2 = torch.from_numpy(pad_vec(np.asarray(bias2, np.float32), w2p.shape[0])).to(x.device)
        out2 = torch.empty((B, Ho, Wo, w2.shape[0]), dtype=torch.float16, device=x.device)
        d.w2, d.scale2, d.bias2, d.out2 = w2p.data_ptr(), sc2.data_ptr(), bi2.data_ptr(), out2.data_ptr()
        d.Cout2, d.act2, d.alpha2 = w2.shape[0], _lib.ACT_ENUM[act2], float(alpha2)
    _lib.check(ctx.lib.od_conv2d_fwd(ctx.handle, C.byref(d), _stream_ptr()), "od_conv2d_fwd")
    if splitk != 1:
        out._splitk_ws = splitk_ws  # keep the workspace alive / inspectable
    if next_pointwise is not None:
        torch.cuda.current_stream().synchronize()  # the packed second-layer weights above are temporaries
        return out, out2
    return out


def conv2d_grouped(xs, w_ohwi: np.ndarray, scale: np.ndarray, bias: np.ndarray, act=None, alpha=0.0, out_f32=False,
                   tile_cfg=-1):
    """The same 3x3 stride-1 layer over several maps xs = [f16 [B,H_i,W_i,Cin]] in one call (od_conv_desc.nseg: the
    prediction module shared by the pyramid levels, reference docs/MODEL.md:8) -> [out_i [B,H_i,W_i,Cout]]."""
    ctx = _ctx(xs[0])
    B, Cin = xs[0].shape[0], xs[0].shape[3]
    Cout, k = w_ohwi.shape[0], w_ohwi.shape[1]
    wp = torch.from_numpy(pack_conv_weight(w_ohwi)).to(xs[0].device)
    sc = torch.from_numpy(pad_vec(np.asarray(scale, np.float32), wp.shape[0])).to(xs[0].device)
    bi = torch.from_numpy(pad_vec(np.asarray(bias, np.float32), wp.shape[0])).to(xs[0].device)
    outs = [torch.empty(tuple(x.shape[:3]) + (Cout,), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
            for x in xs]
    d = _lib.ConvDesc()
    d.w, d.scale, d.bias = wp.data_ptr(), sc.data_ptr(), bi.data_ptr()
    d.B, d.Cin, d.Cout, d.ksize, d.stride = B, Cin, Cout, k, 1
    d.act, d.alpha = _lib.ACT_ENUM[act], float(alpha)
    d.out_dtype = _lib.OD_DT_F32 if out_f32 else _lib.OD_DT_F16
    d.tile_cfg = tile_cfg
    d.nseg = len(xs)
    for i, (x, o) in enumerate(zip(xs, outs)):
        assert x.dtype == torch.float16 and x.is_contiguous() and x.shape[0] == B and x.shape[3] == Cin
        d.seg_x[i], d.seg_out[i], d.seg_H[i], d.seg_W[i] = x.data_ptr(), o.data_ptr(), x.shape[1], x.shape[2]
    _lib.check(ctx.lib.od_conv2d_fwd(ctx.handle, C.byref(d), _stream_ptr()), "od_conv2d_fwd(grouped)")
    torch.cuda.current_stream().synchronize()  # the packed weights above are temporaries
    return outs


def bottleneck(x: torch.Tensor, w1_ohwi: np.ndarray, scale1, bias1, w3_ohwi: np.ndarray, scale3, bias3, act="leaky",
               alpha=0.1):
    """Fused residual block x + act(bn3(conv3x3(act(bn1(conv1x1(x)))))): x f16 [B,H,W,C] -> f16 [B,H,W,C]."""
    ctx = _ctx(x)
    assert x.dtype == torch.float16 and x.is_contiguous()
    B, H, Wd, Cc = x.shape
    assert w1_ohwi.shape == (Cc // 2, 1, 1, Cc) and w3_ohwi.shape == (Cc, 3, 3, Cc // 2)
    keep = []

    def dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(x.device)
        keep.append(t)
        return t
    w1p, w3p = dev(pack_conv_weight(w1_ohwi)), dev(pack_conv_weight(w3_ohwi))
    out = torch.empty_like(x)
    d = _lib.BneckDesc()
    d.x, d.out, d.w1, d.w3 = x.data_ptr(), out.data_ptr(), w1p.data_ptr(), w3p.data_ptr()
    d.scale1 = dev(pad_vec(np.asarray(scale1, np.float32), w1p.shape[0])).data_ptr()
    d.bias1 = dev(pad_vec(np.asarray(bias1, np.float32), w1p.shape[0])).data_ptr()
    d.scale3 = dev(pad_vec(np.asarray(scale3, np.float32), w3p.shape[0])).data_ptr()
    d.bias3 = dev(pad_vec(np.asarray(bias3, np.float32), w3p.shape[0])).data_ptr()
    d.B, d.H, d.W, d.C = B, H, Wd, Cc
    d.act, d.alpha = _lib.ACT_ENUM[act], float(alpha)
    _lib.check(ctx.lib.od_bottleneck_fwd(ctx.handle, C.byref(d), _stream_ptr()), "od_bottleneck_fwd")
    torch.cuda.current_stream().synchronize()  # the packed weights above are temporaries
    return out


def stem(x_u8: torch.Tensor, w0_ohwi: np.ndarray, scale0, bias0, w3_ohwi: np.ndarray, scale3, bias3, act="leaky", alpha=0.1):
    """Fused first two layers: u8 [B,H,W,3] -> f16 [B,H/2,W/2,64] (od_stem_fwd)."""
    ctx = _ctx(x_u8)
    assert x_u8.dtype == torch.uint8 and x_u8.is_contiguous()
    B, H, Wd, _ = x_u8.shape
    assert w0_ohwi.shape == (32, 3, 3, 3) and w3_ohwi.shape == (64, 3, 3, 32)
    keep = []

    def dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(x_u8.device)
        keep.append(t)
        return t
    w0p, w3p = dev(pack_first_weight(w0_ohwi)), dev(pack_conv_weight(w3_ohwi))
    out = torch.empty((B, H // 2, Wd // 2, 64), dtype=torch.float16, device=x_u8.device)
    d = _lib.StemDesc()
    d.x, d.out, d.w0, d.w3 = x_u8.data_ptr(), out.data_ptr(), w0p.data_ptr(), w3p.data_ptr()
    d.scale0 = dev(np.asarray(scale0, np.float32)).data_ptr()
    d.bias0 = dev(np.asarray(bias0, np.float32)).data_ptr()
    d.scale3 = dev(pad_vec(np.asarray(scale3, np.float32), w3p.shape[0])).data_ptr()
    d.bias3 = dev(pad_vec(np.asarray(bias3, np.float32), w3p.shape[0])).data_ptr()
    d.B, d.H, d.W = B, H, Wd
    d.act, d.alpha = _lib.ACT_ENUM[act], float(alpha)
    _lib.check(ctx.lib.od_stem_fwd(ctx.handle, C.byref(d), _stream_ptr()), "od_stem_fwd")
    torch.cuda.current_stream().synchronize()
    return out


def conv_first(x_u8: torch.Tensor, w_ohwi: np.ndarray, scale, bias, act="leaky", alpha=0.1):
    ctx = _ctx(x_u8)
    assert x_u8.dtype == torch.uint8 and x_u8.is_contiguous()
    B, H, Wd, _ = x_u8.shape
    Cout = w_ohwi.shape[0]
    wp = torch.from_numpy(pack_first_weight(w_ohwi)).to(x_u8.device)
    sc = torch.from_numpy(np.asarray(scale, np.float32)).to(x_u8.device)
    bi = torch.from_numpy(np.asarray(bias, np.float32)).to(x_u8.device)
    out = torch.empty((B, H, Wd, Cout), dtype=torch.float16, device=x_u8.device)
    _lib.check(ctx.lib.od_conv_first_fwd(ctx.handle, x_u8.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(),
                                         out.data_ptr(), B, H, Wd, Cout, _lib.ACT_ENUM[act], float(alpha),
                                         _stream_ptr()), "od_conv_first_fwd")
    return out


def upsample2x_add(a: torch.Tensor, up: torch.Tensor):
    ctx = _ctx(a)
    B, H, Wd, Cc = a.shape
    assert tuple(up.shape) == (B, H // 2, Wd // 2, Cc)
    out = torch.empty_like(a)
    _lib.check(ctx.lib.od_upsample2x_add(ctx.handle, a.data_ptr(), up.data_ptr(), out.data_ptr(), B, H, Wd, Cc,
                                         _stream_ptr()), "od_upsample2x_add")
    return out


def decode_locs(locs: torch.Tensor, priors: torch.Tensor, loc_scale=0.1, clip=False):
    """locs f32 [N,P,4] or [P,4] -> boxes, device-side od.pb.decode_locs (reference check_assign.py:27)."""
    ctx = _ctx(locs)
    squeeze = locs.dim() == 2
    l3 = locs.unsqueeze(0) if squeeze else locs
    l3 = l3.contiguous().float()
    N, P, _ = l3.shape
    out = torch.empty_like(l3)
    _lib.check(ctx.lib.od_decode_locs(ctx.handle, l3.data_ptr(), priors.data_ptr(), out.data_ptr(), N, P,
                                      float(loc_scale), int(clip), _stream_ptr()), "od_decode_locs")
    return out[0] if squeeze else out


def loss_fwd_bwd(pred: torch.Tensor, y: torch.Tensor, num_classes=20, alpha=0.25, gamma=2.0, box_mode="smooth_l1",
                 weights=(1.0, 1.0, 1.0)):
    """pred, y f32 [B,P,2+NC+4] -> (losses f32 [4] = obj, cls, box, total ; grad f32 like pred)   (K10)"""
    ctx = _ctx(pred)
    assert pred.dtype == torch.float32 and y.dtype == torch.float32 and pred.is_contiguous() and y.is_contiguous()
    B, P, Cc = pred.shape
    assert Cc == num_classes + 6 and y.shape == pred.shape
    grad = torch.empty_like(pred)
    losses = torch.empty((4,), dtype=torch.float32, device=pred.device)
    wsb = ctx.lib.od_loss_workspace_bytes(B, P)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=pred.device)
    _lib.check(ctx.lib.od_loss_fwd_bwd(ctx.handle, pred.data_ptr(), y.data_ptr(), grad.data_ptr(), losses.data_ptr(),
                                       B, P, num_classes, float(alpha), float(gamma),
                                       {"smooth_l1": 0, "mse": 1}[box_mode], float(weights[0]), float(weights[1]),
                                       float(weights[2]), ws.data_ptr(), wsb, _stream_ptr()), "od_loss_fwd_bwd")
    return losses, grad
