"""Oracle for the training step: the same network in torch (CPU, float64, autograd), BatchNorm in training mode,
loss gradient from oracle/loss.py.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference docs/MODEL.md:5-21 (network), :33-52 (losses), :84-90 (per-layer learning rates)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import loss as oloss
from .network import BN_EPS, NUM_PRIORS, STAGES


def _act(a, act, mask=None):
    if act is None:
        return a
    if act[0] == "leaky":
        if mask is not None:  # slope chosen by a given sign pattern instead of a's own sign (see TorchDetector.slope_masks)
            return a * torch.where(mask, torch.ones((), dtype=a.dtype), torch.full((), act[1], dtype=a.dtype))
        return F.leaky_relu(a, act[1])
    return F.elu(a, act[1])


class TorchDetector:
    """slope_masks: {layer name: bool array [B,Ho,Wo,Cout]} = which side of the LeakyReLU kink every unit of that layer
    is to be treated as being on (True = positive).  A unit whose pre-activation is within the f16 pipeline's rounding
    noise of 0 lands on the other side of the kink on the device than here; its forward value changes by <= 0.9*|a| (tiny)
    but its slope by 10x.  Passing the device's pattern makes this oracle differentiate the SAME piecewise-linear
    function the device ran, so the remaining gradient error is arithmetic only; `flips` counts the units that moved."""

    def __init__(self, params, backbone_act=("leaky", 0.1), head_act=("elu", 1.0), dtype=torch.float64, slope_masks=None):
        self.slope_masks = slope_masks or {}
        self.flips, self.units = {}, {}
        self.batch_stats = {}  # layer -> (batch mean, biased batch variance) of its last call, float64 [Cout]
        self._run, self._mom = None, 0.99
        self.record_patterns = False  # True: forward() keeps this run's own sign pattern per leaky layer in .patterns
        self.patterns = {}
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=not (k.endswith(".mean") or k.endswith(".var")))
                  for k, v in params.items()}
        self.bact, self.hact, self.dtype = backbone_act, head_act, dtype
        self.tower = sum(1 for k in params if k.startswith("h.t") and k.endswith(".w"))

    def conv(self, x, name, stride=1, act=None, res=None, up2=False):
        w = self.p[name + ".w"].permute(0, 3, 1, 2)
        z = F.conv2d(x, w, stride=stride, padding=w.shape[-1] // 2)
        if name + ".gamma" in self.p:
            mu = z.mean((0, 2, 3), keepdim=True)
            var = z.var((0, 2, 3), unbiased=False, keepdim=True)
            self.batch_stats[name] = (mu.detach().reshape(-1).numpy().copy(), var.detach().reshape(-1).numpy().copy())
            if self._run is not None:  # a layer shared by the three levels updates its statistics three times per pass
                m, v = self._run[name]
                bm, bv = self.batch_stats[name]
                self._run[name] = (self._mom * m + (1 - self._mom) * bm, self._mom * v + (1 - self._mom) * bv)
            z = (z - mu) / torch.sqrt(var + BN_EPS)
            z = z * self.p[name + ".gamma"].view(1, -1, 1, 1) + self.p[name + ".beta"].view(1, -1, 1, 1)
        else:
            z = z + self.p[name + ".bias"].view(1, -1, 1, 1)
        mask = self.slope_masks.get(name)
        if mask is not None and act is not None and act[0] == "leaky":
            mask = torch.from_numpy(np.ascontiguousarray(mask)).permute(0, 3, 1, 2)
            self.flips[name] = int(((z.detach() > 0) != mask).sum())
            self.units[name] = mask.numel()
        else:
            mask = None
        if self.record_patterns and act is not None and act[0] == "leaky":
            self.patterns[name] = (z.detach() > 0).permute(0, 2, 3, 1).contiguous().numpy()
        y = _act(z, act, mask)
        if res is not None:
            y = y + (F.interpolate(res, scale_factor=2, mode="nearest") if up2 else res)
        return y

    def forward(self, x_u8, num_classes=20):
        x = torch.tensor(x_u8.astype(np.float64) / 255.0, dtype=self.dtype).permute(0, 3, 1, 2)
        a, hh = self.bact, self.hact
        x = self.conv(x, "b.conv0", act=a)
        taps = []
        for si, (n, _ch) in enumerate(STAGES, start=1):
            x = self.conv(x, f"b.down{si}", stride=2, act=a)
            for r in range(n):
                t = self.conv(x, f"b.s{si}.{r}.a", act=a)
                x = self.conv(t, f"b.s{si}.{r}.b", act=a, res=x)
            taps.append(x)
        c3, c4, c5 = taps[2], taps[3], taps[4]
        p5 = self.conv(c5, "n.lat5", act=hh)
        p4 = self.conv(self.conv(c4, "n.lat4", act=hh, res=p5, up2=True), "n.out4", act=hh)
        p3 = self.conv(self.conv(c3, "n.lat3", act=hh, res=p4, up2=True), "n.out3", act=hh)
        outs = []
        C = 2 + num_classes + 4
        for lv in (p3, p4, p5):
            t = lv
            for i in range(self.tower):
                t = self.conv(t, f"h.t{i}", act=hh)
            o = self.conv(t, "h.out").permute(0, 2, 3, 1)
            outs.append(o.reshape(o.shape[0], -1, C))
        return torch.cat(outs, 1)

    def running_stats_after(self, batches, momentum=0.99, num_classes=20):
        """BatchNorm running statistics after forward passes over `batches` (list of uint8 [B,S,S,3]) with the parameters held
        fixed: {layer: (run_mean, run_var)}.  Convention frozen for this build [BUILD-DEFINED, Keras' non-fused BatchNorm]:
        moving = momentum * moving + (1 - momentum) * batch, with the BIASED (population) batch variance, momentum 0.99,
        starting from the parameters' .mean / .var."""
        self._run = {k[:-6]: (self.p[k[:-6] + ".mean"].detach().numpy().astype(np.float64).copy(),
                              self.p[k[:-6] + ".var"].detach().numpy().astype(np.float64).copy())
                     for k in self.p if k.endswith(".gamma")}
        self._mom = momentum
        try:
            for x in batches:
                with torch.no_grad():
                    self.forward(x, num_classes)
            return self._run
        finally:
            self._run = None

    def loss_and_grads(self, x_u8, y_target, num_classes=20, box_mode="smooth_l1"):
        pred = self.forward(x_u8, num_classes)
        losses, g = oloss.loss_and_grad(pred.detach().numpy(), y_target, num_classes, box_mode=box_mode)
        pred.backward(torch.tensor(g, dtype=self.dtype))
        grads = {k: v.grad.numpy() for k, v in self.p.items() if v.grad is not None}
        return losses, grads, pred.detach().numpy()
