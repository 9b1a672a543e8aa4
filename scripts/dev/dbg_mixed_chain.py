"""Continue the mixed plan on the CPU from the DEVICE's own intermediate tensors (debugging aid)."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from object_detector_amd.detector import ObjectDetector  # noqa: E402
from oracle import network as onet  # noqa: E402

B, S = 2, 96
dev = torch.device("cuda:0")
x = onet.synthetic_images(B, S, seed=0)
od = ObjectDetector.synthetic(B, (S, S), seed=2, device=dev, use_multi_gpu=False, precision="mixed", n_inflight=1)
od.predict_batch_device(torch.from_numpy(x).to(dev))
torch.cuda.synchronize()
pred = od.net.pred.cpu().numpy()
mp = onet.MixedPlan(od.net.stream_stages, od.net.split)
r = mp.runner(od.params)
rms = lambda a: float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))  # noqa: E731
sc = np.abs(pred).max()
# A: the model's head on the device's p3 / p4 / p5 (hi + lo)
hl = {inf["name"]: inf for inf in od.net.op_info if inf["kind"] == "wide"}


def pair(name):
    t = hl[name]["hilo"].float().cpu().numpy()
    c = t.shape[-1] // 2
    return t[..., :c] + t[..., c:]


levels = [pair("n.out3.split"), pair("n.out4.split"), pair("n.lat5.split")]
a = r.head(levels)
print(f"A head(device p3,p4,p5) vs device logits: rms {rms(a - pred) / sc:.3e} of scale")
# B: the model's neck + head on the device's taps (f16 copies widened)
c3, c4, c5 = (t.float().cpu().numpy() for t in od.net.taps)
b = r.head(r.neck(c3, c4, c5))
print(f"B head(neck(device taps)) vs device logits: rms {rms(b - pred) / sc:.3e} of scale")
for nm, lv in zip(("n.out3", "n.out4", "n.lat5"), r.neck(c3, c4, c5)):
    print(f"   {nm}: model-from-device-taps vs device pair: rms {rms(lv - pair(nm + '.split')) / np.abs(lv).max():.3e}")
r.trace = {}
r.neck(c3, c4, c5)
for nm in ("n.lat5", "n.lat4", "n.out4", "n.lat3", "n.out3"):
    want = r.trace[nm][0]
    got = pair(nm + ".split")
    print(f"   {nm}: model vs device pair rms {rms(want - got) / np.abs(want).max():.3e}; max {np.abs(want - got).max() / np.abs(want).max():.3e}")
    if hl[nm + ".split"]["out16"] is not None:
        o16 = hl[nm + ".split"]["out16"].float().cpu().numpy()
        print(f"        out16 vs f16(model): differing elements {np.mean(o16 != want.astype(np.float16).astype(np.float32)):.3%}")
# n.lat4 again by hand: ELU(bn(conv(c4))) + up2(f16 p5)
p5_16 = hl["n.lat5.split"]["out16"].float().cpu().numpy()
r2 = onet.Runner(od.params, storage="f32")
m4 = r2.conv(c4, "n.lat4", act=("elu", 1.0), res=p5_16, res_up2=True, store=False)
print("   n.lat4 by hand vs device pair:", rms(m4 - pair("n.lat4.split")) / np.abs(m4).max(), " vs model trace:", rms(m4 - r.trace["n.lat4"][0]) / np.abs(m4).max())
