"""GQA-OOD iteration (src/gqa/gqa_ood.py:165-292): GGM pass first (KL weight 12), plain
pass second.  Thin front-end over xggm_amd.vqa.vqacpv2."""
from ..vqa.vqacpv2 import (loss_func, compute_kl_loss, BCEWithLogitsLoss, plain_pass, ggm_pass, predict, evaluate,  # noqa: F401
                           train_iteration as _train_iteration, make_optimizer)  # noqa: F401


def train_iteration(model, optim, bce_loss, batch, delta=5, sigma=1.0, branch=None, clip=5.0):
    return _train_iteration(model, optim, bce_loss, batch, delta=delta, sigma=sigma, order="gqa", branch=branch,
                            clip=clip)
