"""Layer-by-layer comparison of the device's mixed-precision plan with oracle.network.MixedPlan (debugging aid)."""
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
mp = onet.MixedPlan(od.net.stream_stages, od.net.split)
r = mp.runner(od.params)
r.trace = {}
r.forward(x)
r32 = onet.Runner(od.params, storage="f32")
r32.trace = {}
r32.forward(x)
seen = {}
for inf in od.net.op_info:
    if inf["kind"] != "conv" or inf.get("out") is None:
        continue
    nm = inf["name"]
    k = seen.get(nm, 0)
    seen[nm] = k + 1
    got = inf["out"].float().cpu().numpy()
    if nm not in r.trace:
        continue
    want = r.trace[nm][k]
    ref = r32.trace[nm][k]
    if got.shape != want.shape:
        print(nm, "shape", got.shape, want.shape)
        continue
    sc = np.abs(ref).max()
    print(f"{nm:12s} out {str(inf['out'].dtype):14s} dev-model rms {np.sqrt(((got - want) ** 2).mean()) / sc:.2e}  "
          f"model-fp32 rms {np.sqrt(((want - ref) ** 2).mean()) / sc:.2e}  dev-fp32 rms {np.sqrt(((got - ref) ** 2).mean()) / sc:.2e}")
print("taps (device f16 copy / f32 stream vs the model's f32 stream and the fp32 reference):")
rms = lambda a: float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))  # noqa: E731
last = {3: "b.s3.7.b", 4: "b.s4.7.b", 5: "b.s5.3.b"}
x32 = {inf["name"]: inf for inf in od.net.op_info if inf["kind"] == "wide"}
for k, tap in zip((3, 4, 5), od.net.taps):
    want, ref = r.trace[last[k]][0], r32.trace[last[k]][0]
    d16 = tap.float().cpu().numpy()
    d32 = x32[f"b.s{k}.{int(last[k].split('.')[2])}.add"]["out32"].cpu().numpy()
    sc = np.abs(ref).max()
    print(f"  stage {k}: f32 stream dev-fp32 {rms(d32 - ref) / sc:.3e} model-fp32 {rms(want - ref) / sc:.3e} | f16 copy dev-fp32 {rms(d16 - ref) / sc:.3e} "
          f"f16(model)-fp32 {rms(want.astype(np.float16).astype(np.float32) - ref) / sc:.3e}")
