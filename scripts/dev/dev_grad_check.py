import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent / "tests"))
import numpy as np, torch
from test_gpu_trainer import _setup, _rel
from object_detector_amd.trainer import Trainer
from oracle.train_ref import TorchDetector
cuda = torch.device("cuda:0")
B, S = 2, 96
ls = float(sys.argv[1]) if len(sys.argv) > 1 else 256.0
params, x, anns = _setup(cuda, B, S)
bact = ("elu", 1.0) if len(sys.argv) > 2 and sys.argv[2] == "elu" else ("leaky", 0.1)
tr = Trainer(params, B, (S, S), device=cuda, lr=0.0, loss_scale=ls, backbone_act=bact)
y, _n, _ = tr.pb.encode_batch(anns, return_device=True)
pred = tr.forward(torch.from_numpy(x).to(cuda)).clone()
tr.loss(y); g = tr.backward().cpu().numpy() / ls
rl, rg, rp = TorchDetector(params, backbone_act=bact).loss_and_grads(x, y.cpu().numpy())
print("pred err", np.abs(pred.cpu().numpy() - rp).max(), "scale", np.abs(rp).max())
for (name, kind), (o, n) in tr.seg.items():
    ref = rg[f"{name}.{kind}"].reshape(-1)
    print(f"{name:12s} {kind:6s} rel={_rel(g[o:o+n], ref):.4f} |ref|={np.linalg.norm(ref):.3e} |got|={np.linalg.norm(g[o:o+n]):.3e}")
