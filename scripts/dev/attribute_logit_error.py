"""which stored tensors carry the f16-storage logit error?  CPU only (oracle Runner), no GPU.
f16 storage is enabled for ONE group of layer outputs at a time (everything else stays f32) and the logit error vs the
all-f32 run is reported as a share of the all-f16 run's.  usage: python scripts/dev/attribute_logit_error.py [B] [S] [weights.npz]"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import network as onet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
if len(sys.argv) > 3:
    from object_detector_amd import weights as W
    params = onet.f16_weights(W.load(sys.argv[3])[0])
    tag = pathlib.Path(sys.argv[3]).name
else:
    params, tag = onet.init_weights(2), "random-init seed 2"
x = onet.synthetic_images(B, S, seed=0)


def stage_of(n):
    if n.startswith("b.conv0") or n.startswith("b.down1") or n.startswith("b.s1") or n.startswith("b.down2") or n.startswith("b.s2"):
        return "stem+s1-2"
    for k in (3, 4, 5):
        if n.startswith(f"b.down{k}") or n.startswith(f"b.s{k}"):
            return f"s{k}"
    return "neck+head"


groups = {
    "all (= the f16-storage oracle)": lambda n: True,
    "stem + stages 1-2 (160^2, 80^2 maps)": lambda n: stage_of(n) == "stem+s1-2",
    "stage 3 (40^2)": lambda n: stage_of(n) == "s3",
    "stage 4 (20^2)": lambda n: stage_of(n) == "s4",
    "stage 5 (10^2)": lambda n: stage_of(n) == "s5",
    "neck + prediction module": lambda n: stage_of(n) == "neck+head",
    "residual stream only (conv0, down*, *.b)": lambda n: n.startswith("b.") and not n.endswith(".a"),
    "block-internal tensors only (*.a)": lambda n: n.startswith("b.") and n.endswith(".a"),
    "residual stream of stages 3-5 only": lambda n: n.startswith("b.") and not n.endswith(".a") and stage_of(n) in ("s3", "s4", "s5"),
    "residual stream of stages 4-5 only": lambda n: n.startswith("b.") and not n.endswith(".a") and stage_of(n) in ("s4", "s5"),
    "everything EXCEPT the residual stream of stages 3-5 + neck/head": lambda n: not ((n.startswith("b.") and not n.endswith(".a") and stage_of(n) in ("s3", "s4", "s5")) or stage_of(n) == "neck+head"),
    "everything EXCEPT stages 4-5 + neck/head": lambda n: stage_of(n) in ("stem+s1-2", "s3"),
    "everything EXCEPT the residual stream (all stages) + neck/head": lambda n: n.startswith("b.") and n.endswith(".a"),
}
ref = onet.Runner(params, storage="f32").forward(x)
scale = float(np.abs(ref).max())
out = {"weights": tag, "batch": B, "size": S, "logit_scale": scale, "groups": {}}
base = None
for name, pred in groups.items():
    got = onet.Runner(params, storage=pred).forward(x)
    d = (got - ref).astype(np.float64)
    rms, mx = float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())
    if base is None:
        base = rms
    out["groups"][name] = dict(rms=rms, rms_rel_scale=rms / scale, max_rel_scale=mx / scale, share_of_variance=(rms / base) ** 2)
    print(f"{name:68s} rms {rms:.3e} = {rms / scale:.2e} x scale   max {mx / scale:.2e} x scale   variance share {(rms / base) ** 2:5.1%}",
          flush=True)
print(json.dumps(out))
