import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import oracle as O
from conftest import GOLDEN, rel_l2
from test_gpu_model import DEV, bending_fixture, build_model, copy_norms_frozen, graph_to, load_bending_norms
from test_gpu_training import run_loss_backward
d4 = np.load(f"{GOLDEN}/g4_loss_grads.npz")
d, cfg, sd = bending_fixture()
model = build_model(cfg, sd).train()
norms = {}
load_bending_norms(model, d, norms)
g = O.synthetic_graph(32, 16, seed=int(d4["graph_seed"]), lx=0.08, ly=0.01, with_target=True)
ls, grads = run_loss_backward(model, graph_to(g, DEV))
print(ls, d4["loss"])
sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
ol = O.loss(sdg, cfg, norms, g); ol[0].backward()
# fp64 oracle
sd64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
for name, l2 in zip(d4["grad_names"], d4["grad_l2"]):
    k = str(name)
    print(f"{k:40s} norm-rel {abs(float(grads[k].double().norm())-l2)/l2:.2e}  vs oracle32 {rel_l2(grads[k], sdg[k].grad):.2e}")
