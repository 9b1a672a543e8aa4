import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent / "tests"))
import numpy as np, torch
from test_gpu_trainer import _setup
from object_detector_amd.trainer import Trainer
cuda = torch.device("cuda:0")
B, S = 2, 96
lr = float(sys.argv[1]); mom = float(sys.argv[2])
params, x, anns = _setup(cuda, B, S)
tr = Trainer(params, B, (S, S), device=cuda, lr=lr, momentum=mom, loss_scale=256.0)
xt = torch.from_numpy(x).to(cuda)
for i in range(12):
    l = tr.step(xt, anns).cpu().numpy()
    print(i, l)
