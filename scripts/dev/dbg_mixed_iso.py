"""Every conv op of the mixed plan against the oracle on the DEVICE's own input of that op (debugging aid)."""
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
run = onet.Runner(od.params, storage="f32", precise=True)
names = od.net.time_ops()[1]
for inf, kn in zip(od.net.op_info, names):
    if inf["kind"] != "conv":
        continue
    nm = inf["name"]
    xin = inf["x"].cpu().numpy().astype(np.float32)
    if inf.get("split"):
        c = xin.shape[-1] // 2
        xin = xin[..., :c] + xin[..., c:]
    res = None if inf["res"] is None else inf["res"].cpu().numpy().astype(np.float32)
    r = run.conv(xin, nm, stride=inf["stride"], act=inf["act"], res=res, res_up2=inf["res_mode"] == 2, store=False)
    if inf["out"] is None:
        off, rows_n = inf["pred_rows"]
        d = od.net.pred[:, off:off + rows_n].cpu().numpy().reshape(r.shape)
    else:
        d = inf["out"].cpu().numpy().astype(np.float32)
    err = np.abs(d - r)
    sc = np.abs(r).max()
    print(f"{nm:10s} {str(inf['out'].dtype if inf['out'] is not None else 'pred'):14s} split={bool(inf.get('split'))!s:5s} "
          f"max {err.max() / sc:.2e} rms {np.sqrt((err ** 2).mean()) / sc:.2e} of scale {sc:.2f}   {kn[:60]}")
print("wide ops (outputs vs their own inputs; stream ops reuse y32 / x32, so only the LAST block of a stage is checkable):")
for inf in od.net.op_info:
    if inf["kind"] != "wide":
        continue
    y = inf["y"].cpu().numpy()
    msg = [inf["name"]]
    if inf["res"] is None:
        v = y
        if inf["out16"] is not None:
            msg.append(f"out16==f16(y): {np.array_equal(inf['out16'].cpu().numpy(), v.astype(np.float16))}")
        if inf["hilo"] is not None:
            hl = inf["hilo"].cpu().numpy().astype(np.float32)
            c = v.shape[-1]
            msg.append(f"hi==f16(y): {np.array_equal(hl[..., :c], v.astype(np.float16).astype(np.float32))} "
                       f"max|hi+lo-y|/max|y| {np.abs(hl[..., :c] + hl[..., c:] - v).max() / np.abs(v).max():.2e}")
    else:
        x32 = inf["out32"].cpu().numpy()
        msg.append(f"out16==f16(out32): {np.array_equal(inf['out16'].cpu().numpy(), x32.astype(np.float16))}")
    print("  ", *msg)
