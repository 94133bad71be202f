"""Oracle training step on CPU (test infrastructure; see oracle/__init__.py).

Restates the step the reference's Lightning loop performs around ``LSSD3D.training_step``
(``lesions3d/ssd3d.py:467-531``) and ``configure_optimizers`` (``ssd3d.py:704-722``):
forward -> MultiBox loss -> ``loss = conf + alpha * loc`` -> backward -> Adam(wd 5e-4, biases at 2*lr)
with the cosine schedule stepped once per training step INSIDE ``training_step`` (ssd3d.py:527-529), which
Lightning's automatic optimisation runs within the optimiser closure: update k uses the LR after k scheduler steps.
Used by the parity tests and as the timed CPU baseline of ``bench.py`` (kind "port").
"""
import torch

from .multibox import multibox_loss


def make_optimizer(model, lr, scheduler=True):
    """ssd3d.py:704-722.  ``rescale_factors`` is a parameter without gradient (unused in forward);
    torch's Adam skips parameters whose grad is None."""
    biases, others = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (biases if name.endswith(".bias") else others).append(p)
    opt = torch.optim.Adam([{"params": biases, "lr": 2 * lr}, {"params": others}], lr=lr, weight_decay=0.0005)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=40) if scheduler else None
    return opt, sch


def train_step(model, opt, sch, images, boxes, labels, threshold, alpha=1.0):
    """One optimisation step; returns (loss, conf_loss, loc_loss) as Python floats."""
    model.train()
    opt.zero_grad(set_to_none=True)
    locs, scores = model(images)
    conf, loc = multibox_loss(locs, scores, boxes, labels, model.priors_cxcycz, threshold)
    loss = conf + alpha * loc  # ssd3d.py:494
    loss.backward()
    if sch is not None:  # ssd3d.py:527-529 runs inside the closure, i.e. before the parameter update
        sch.step()
    opt.step()
    return float(loss.detach()), float(conf.detach()), float(loc.detach())
