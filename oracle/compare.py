"""Size-independent comparison of device logits with the oracle's.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used
by tests/, __graft_entry__.smoke() and nothing in the product package.

north_star asks for logits "within 1e-3" of the reference's fp32 path.  With f16 activation storage (which north_star also
prescribes) that is not a per-element property: every stored activation carries a relative rounding error of up to 2^-11,
52 convolutions and 23 residual additions deep.  Measured on MI355X at the BASELINE sizes (gpurun_out/fullsize_parity.json,
DESIGN.md §5): rms |dlogit| = 2.2e-4 of the logit scale (5.9e-3 absolute at scale 26.6), identical for the device and for
the CPU oracle run with the same storage format; the maximum over n logits follows the Gaussian tail (0.95e-3 of the
scale over 78 k logits, 1.5e-3 over 14 M).
"""
from __future__ import annotations

import numpy as np


def logit_stats(got, ref16, ref32):
    """End-to-end logit comparison that does not depend on how many logits are compared.

    Both the device and the f16-storage oracle round every stored activation to f16 at the same points, but their f32
    accumulation orders differ, so a few roundings per layer fall the other way; after a handful of layers the two runs
    are independent realisations of the same f16 storage noise (per-layer exactness is pinned by tests/test_gpu_fullsize.py
    `_check_layers_in_isolation`).  The MAX difference over n logits therefore grows like sigma * sqrt(2 ln n) (78 k logits at 2 x 96^2, 14 M at
    32 x 320^2) while sigma itself does not; asserted are
      (a) sigma = rms(dev - f16 oracle) <= 3e-4 of the logit scale,
      (b) no outliers: max <= sigma * (sqrt(2 ln n) + 3)  (a wrong tile / race shows up here),
      (c) the device is as close to the plain fp32 oracle as the CPU f16-storage run is: rms ratio <= 1.15."""
    scale = max(1.0, float(np.abs(ref32).max()))
    d = (got - ref16).astype(np.float64)
    sigma = float(np.sqrt(np.mean(d * d)))
    mx = float(np.abs(d).max())
    n = d.size
    e_dev = float(np.sqrt(np.mean((got - ref32).astype(np.float64) ** 2)))
    e_cpu = float(np.sqrt(np.mean((ref16 - ref32).astype(np.float64) ** 2)))
    rec = dict(n_logits=n, logit_scale=scale, logit_rms=float(np.sqrt(np.mean(ref32.astype(np.float64) ** 2))),
               max_abs_dlogit=mx, rms_dlogit=sigma, max_rel_scale=mx / scale, rms_rel_scale=sigma / scale,
               max_over_sigma=mx / sigma, gaussian_max_over_sigma=float(np.sqrt(2 * np.log(n))),
               rms_dev_vs_fp32=e_dev, rms_f16oracle_vs_fp32=e_cpu,
               max_dev_vs_fp32=float(np.abs(got - ref32).max()), max_f16oracle_vs_fp32=float(np.abs(ref16 - ref32).max()))
    return rec


def assert_logits(rec, tag):
    print(f"[{tag}] |dlogit| vs f16-storage oracle: max {rec['max_abs_dlogit']:.3e} = {rec['max_rel_scale']:.2e} of scale "
          f"{rec['logit_scale']:.2f}, rms {rec['rms_dlogit']:.3e} = {rec['rms_rel_scale']:.2e} of scale, max/sigma "
          f"{rec['max_over_sigma']:.1f} (gaussian {rec['gaussian_max_over_sigma']:.1f}); vs fp32 oracle: rms dev "
          f"{rec['rms_dev_vs_fp32']:.3e} / cpu-f16 {rec['rms_f16oracle_vs_fp32']:.3e}")
    assert rec["rms_rel_scale"] <= 3e-4, rec
    assert rec["max_over_sigma"] <= rec["gaussian_max_over_sigma"] + 3.0, rec
    assert rec["rms_dev_vs_fp32"] <= 1.15 * rec["rms_f16oracle_vs_fp32"], rec
    # every logit of the DEFAULT (f16-storage) plan within 3e-3 x scale of the plain fp32 oracle (measured 1.4-1.5e-3 over
    # 14 / 28 M logits of a random-init network, 3.3e-4 with trained weights); precision="mixed" is held to north_star's
    # 1e-3 x scale on EVERY logit (tests/test_gpu_mixed_precision.py)
    assert rec["max_dev_vs_fp32"] <= 3e-3 * rec["logit_scale"], rec
