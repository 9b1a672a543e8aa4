"""Device mixed plan vs oracle MixedPlan for several (stream, split) choices (debugging aid)."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from object_detector_amd import weights as W  # noqa: E402
from object_detector_amd.net import Net  # noqa: E402
from oracle import network as onet  # noqa: E402

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device("cuda:0")
x = onet.synthetic_images(B, S, seed=0)
params = W.random_init(2)
ref = onet.Runner(params, storage="f32").forward(x)
sc = np.abs(ref).max()
rms = lambda a: float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))  # noqa: E731
for st, sp in [((), ()), ((3, 4, 5), ()), ((3,), ()), ((4,), ()), ((5,), ()), ((), ("h.out",)), ((), ("h.t0", "h.out")), ((), ("n.out3", "n.out4")),
               ((), ("n.out3", "n.out4", "h.t0", "h.out")), ((3, 4, 5), ("n.out3", "n.out4", "h.t0", "h.out"))]:
    net = Net(params, B, (S, S), device=dev, precision="mixed", stream_stages=st, split=sp)
    got = net.forward(torch.from_numpy(x).to(dev)).cpu().numpy()
    model = onet.MixedPlan(st, sp, net.wide_fpn).runner(params).forward(x)
    print(f"stream {st} split {sp}: dev-fp32 rms {rms(got - ref) / sc:.3e}  model-fp32 rms {rms(model - ref) / sc:.3e}  "
          f"ratio {rms(got - ref) / rms(model - ref):.3f}  dev-model rms {rms(got - model) / sc:.3e}", flush=True)
