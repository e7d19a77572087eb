"""GIN's eps gradient in bf16 against the reference golden (gen_gin36): how far off is it really, and where does the
difference come from?  python tools/exp_gin_eps.py  (on an MI355X)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_model_gpu import load_golden, probe, DEV, F32, BF16  # noqa: E402
from xggm_amd import synth  # noqa: E402


def run(dt, seed_shift=0):
    from xggm_amd.module.graph_generative_modeling import GINGenerator
    from xggm_amd.runtime import set_compute_dtype
    g = load_golden("gen_gin36")
    H, N, B, nl, seed = int(g["H"]), int(g["N"]), int(g["B"]), int(g["n_layers"]), int(g["seed"])
    gen = GINGenerator(hidden_dim=H, n_layers=nl)
    sd = {k: torch.from_numpy(synth.seeded_param("generator." + k, v.shape, seed)) for k, v in gen.state_dict().items()}
    gen.load_state_dict(sd)
    gen = set_compute_dtype(gen.to(DEV), dt).eval()
    xn, an = synth.generator_inputs("gen_gin36", "GIN", B, N, H, seed)
    x = torch.from_numpy(xn).to(DEV, dt).requires_grad_(True)
    adj = torch.from_numpy(an).to(DEV).requires_grad_(True)
    xo, ao = gen(x, adj)
    loss = (xo.float() * probe("xo", xo.shape, seed, device=DEV)).sum() + (ao * probe("ao", ao.shape, seed, device=DEV)).sum()
    loss.backward()
    G = {"generator." + k: p.grad.detach().double().cpu() for k, p in gen.named_parameters()}
    out = {}
    for n, rn in zip(g["grad_names"], g["grad_norms"]):
        if str(n).endswith("eps"):
            out[str(n)] = (float(G[str(n)].norm()), float(rn))
    return out


for dt in (F32, BF16):
    for n, (got, ref) in run(dt).items():
        print("%s %-60s got %.6f ref %.6f rel err %.3e" % ("f32 " if dt == F32 else "bf16", n, got, ref, abs(got - ref) / ref))
