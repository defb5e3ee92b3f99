#!/usr/bin/env python3
"""Diagnostic (GPU box): accuracy of the three matrix paths against an fp64 evaluation of the reference
formulation, outside the benign regime of the parity tests: scaled latents, large LayerNorm gains, scaled
edge-MLP weights, a 15-block residual stream.  Prints rel-L2 of the node and edge outputs per case and mode
(and of the fp32 CPU oracle, as the fp32 noise floor of the same case)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from gnn_amd import ops  # noqa: E402
from gnn_amd.meshgraphnet import EdgeNodeMessagePassing  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def layers(sd, n_layers):
    out = []
    for i in range(n_layers):
        m = EdgeNodeMessagePassing(128)
        m.load_state_dict({k[len(f"processors.{i}."):]: v for k, v in sd.items() if k.startswith(f"processors.{i}.")})
        out.append(m.to(dev))
    return out


def case(name, scale, gamma, wscale, n_layers, seed=1):
    cfg = O.OracleConfig(process_steps=n_layers)
    sd = O.init_state_dict(cfg, 5)
    gen = torch.Generator().manual_seed(seed)
    for k in list(sd):
        if ".4.weight" in k:
            sd[k] = gamma * (1.0 + 0.3 * torch.randn(128, generator=gen))
        elif ".4.bias" in k:
            sd[k] = 0.3 * gamma * torch.randn(128, generator=gen)
        elif "edge_mlp" in k and (".0." in k or ".2." in k):
            sd[k] = sd[k] * wscale
    g = O.synthetic_graph(40, 30, seed=1)
    n, ne = g["mesh_pos"].shape[0], g["edge_index"].shape[1]
    x, e = scale * torch.randn(n, 128, generator=gen), scale * torch.randn(ne, 128, generator=gen)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        x64, e64, x32, e32 = x.double(), e.double(), x, e
        for i in range(n_layers):
            x64, e64 = O.mp_layer(sd64, i, x64, g["edge_index"], e64)
            x32, e32 = O.mp_layer(sd, i, x32, g["edge_index"], e32)
    line = f"{name:44s} max|x_out| {float(x64.abs().max()):9.3g}  cpu-fp32 {rel(x32, x64):.2e}/{rel(e32, e64):.2e}"
    for mode in ("f16x2", "bf16x3", "f32"):
        prev = ops.set_matmul_mode(mode)
        ls = layers(sd, n_layers)
        xo, eo = x.to(dev), e.to(dev)
        ei = g["edge_index"].to(dev)
        for m in ls:
            xo, eo = m(xo, ei, eo)
        tripped = ops.range_check(reset=True)
        ops.set_matmul_mode(prev)
        line += f"  {mode} {rel(xo, x64):.2e}/{rel(eo, e64):.2e}{' RANGE' if tripped else ''}"
    print(line, flush=True)


for sc in (1e-3, 1.0, 1e2, 1e3, 1e4, 3e4, 1e5):
    case(f"1 block, latents x{sc:g}", sc, 1.0, 1.0, 1)
for gm in (3.0, 10.0):
    case(f"1 block, LN gain {gm:g}", 1.0, gm, 1.0, 1)
case("1 block, edge-MLP weights x8", 1.0, 1.0, 8.0, 1)
case("1 block, LN gain 10, edge weights x8, x1e2", 1e2, 10.0, 8.0, 1)
for gm in (1.0, 3.0, 10.0):
    case(f"15 blocks, LN gain {gm:g}", 1.0, gm, 1.0, 15)
case("15 blocks, LN gain 10, edge weights x8", 1.0, 10.0, 8.0, 15)
