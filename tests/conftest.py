import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from object_detector_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


_ORACLE_LOGITS = {}


def oracle_logits(B, S, kind, seed_weights=2, seed_images=0):
    """Oracle forward of the bench workload (random-init weights `seed_weights`, synthetic images `seed_images`), cached for
    the test session: the fp32 pass at 16 x 640^2 costs ~30 s on the GPU box's host cores and several full-size tests want the
    same tensor.  kind: "f16" / "f32" storage, or ("mixed", stream_stages, split, wide_fpn)."""
    from oracle import network as onet
    key = (B, S, kind, seed_weights, seed_images)
    if key not in _ORACLE_LOGITS:
        params = onet.init_weights(seed_weights)
        x = onet.synthetic_images(B, S, seed=seed_images)
        if isinstance(kind, tuple):
            run = onet.MixedPlan(*kind[1:]).runner(params)
        else:
            run = onet.Runner(params, storage=kind)
        _ORACLE_LOGITS[key] = run.forward(x)
    return _ORACLE_LOGITS[key]
