"""Which MIXED-precision plan reaches north_star's 1e-3 x scale?  Unlike attribute_logit_error.py (f16 storage switched on
for one group of tensors at a time) this models what the device can do -- oracle.network.MixedPlan: every convolution runs on
f16 MFMA operands unless it is a split-operand layer ((hi, lo) input pair = ~22 bits, 2x the MFMA work of that layer); what
can be wide cheaply are the ADD paths (the residual stream, the FPN sums).  CPU only.
usage: python scripts/dev/attribute_logit_error2.py [B] [S]"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import network as onet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
params = onet.init_weights(2)
x = onet.synthetic_images(B, S, seed=0)
HEAD4 = ("n.out3", "n.out4", "h.t0", "h.out")
ALL7 = ("n.lat3", "n.lat4", "n.lat5") + HEAD4
plans = [
    ("f16 everywhere (the default plan)", None),
    ("f32 FPN sums only", ((), (), True)),
    ("f32 stream s3-5, f16 FPN sums", ((3, 4, 5), (), False)),
    ("f32 stream s3-5 + f32 FPN sums", ((3, 4, 5), (), True)),
    ("  + split h.out", ((3, 4, 5), ("h.out",), True)),
    ("  + split h.t0, h.out", ((3, 4, 5), ("h.t0", "h.out"), True)),
    ("  + split n.out3/4, h.t0, h.out  (precision='mixed' default)", ((3, 4, 5), HEAD4, True)),
    ("  + split all 7 neck / head layers", ((3, 4, 5), ALL7, True)),
    ("f32 stream s4-5 + f32 FPN sums + split n.out3/4, h.t0, h.out", ((4, 5), HEAD4, True)),
    ("f32 stream s4-5 + f32 FPN sums + split n.lat4/5, n.out3/4, h.t0, h.out", ((4, 5), ("n.lat4", "n.lat5") + HEAD4, True)),
    ("f32 FPN sums + split n.out3/4, h.t0, h.out (f16 stream)", ((), HEAD4, True)),
]
ref = onet.Runner(params, storage="f32").forward(x)
scale = float(np.abs(ref).max())
base = None
print(f"random-init seed 2, {B} x {S}^2, logit scale {scale:.1f}; 'max 14 M' = 6.6 x rms (the measured max / rms over 14 M logits)")
for name, spec in plans:
    run = onet.Runner(params, storage="f16") if spec is None else onet.MixedPlan(*spec).runner(params)
    got = run.forward(x)
    d = (got - ref).astype(np.float64)
    rms, mx = float(np.sqrt(np.mean(d * d))), float(np.abs(d).max())
    base = base or rms
    print(f"{name:76s} rms {rms / scale:.2e}  max({d.size / 1e6:.1f} M) {mx / scale:.2e}  variance {(rms / base) ** 2:6.1%}  "
          f"max 14 M ~ {6.6 * rms / scale:.2e} x scale", flush=True)
