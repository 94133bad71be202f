"""Optimiser + LR schedule of ``LSSD3D.configure_optimizers`` (reference ``lesions3d/ssd3d.py:704-722``):
Adam with L2 weight decay 5e-4, '.bias' parameters at twice the learning rate, cosine annealing (T_max = 40)
stepped once per training step (ssd3d.py:527-529).  One fused HIP launch over the flat parameter arena."""
import math

import torch

from . import _lib
from ._lib import ptr


class FusedAdam:
    """API subset of ``torch.optim.Adam`` (``step`` / ``zero_grad`` / ``param_groups`` / ``state_dict``) backed by
    ``msl_adam_step``.  Arithmetic order follows torch's single-tensor Adam (lerp, addcmul, addcdiv)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0005):
        self.model = model
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.param_groups = [{"name": "biases", "lr": 2 * lr, "initial_lr": 2 * lr},   # ssd3d.py:714
                             {"name": "not_biases", "lr": lr, "initial_lr": lr}]
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self.hp = None
        self._hp_ring = None  # pinned staging slots of the hyper-parameter vector: [host tensor, event of its last copy]

    def _ensure(self):
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":
            raise _lib.HipKernelError("FusedAdam runs on the HIP device only (no CPU fallback)")
        arena = self.model._engine.ensure_arena(dev)
        if self.exp_avg is None or self.exp_avg.numel() != arena.n_trainable or self.exp_avg.device != dev:
            self.exp_avg = torch.zeros(arena.n_trainable, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(arena.n_trainable, dtype=torch.float32, device=dev)
            self.hp = torch.zeros(8, dtype=torch.float32, device=dev)
        return arena

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def hyper(self, grad_scale=1.0):
        t = self.step_count
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        return [self.param_groups[0]["lr"] / bc1, self.param_groups[1]["lr"] / bc1, math.sqrt(bc2), b1, b2, self.eps,
                self.weight_decay, grad_scale]

    def step(self, grad_scale=1.0, gather_autograd_grads=True):
        """``gather_autograd_grads``: copy ``p.grad`` (autograd path) into the flat gradient arena first; the fused
        training step writes the arena directly and passes False."""
        arena = self._ensure()
        if gather_autograd_grads:
            for name, view in arena.grad_views.items():
                g = arena.params[name].grad
                if g is None:
                    view.zero_()
                elif g.data_ptr() != view.data_ptr():
                    view.copy_(g)
        self.prepare_step(grad_scale)
        _lib.call("msl_adam_step", ptr(arena.flat), ptr(arena.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.hp),
                  ptr(arena.is_bias), arena.n_trainable, torch.cuda.current_stream().cuda_stream)

    def prepare_step(self, grad_scale=1.0):
        """Host side of a step: advance the step counter and refresh the 8 hyper-parameter floats the kernel reads
        (the launch itself may then come from a recorded program)."""
        self._ensure()
        self.step_count += 1
        # Through a ring of PINNED staging buffers: a host-to-device copy from pageable memory holds the host until every
        # earlier command of the stream - the whole previous step - has finished, so the next step's launches could only be
        # enqueued into an idle GPU (~50 us per step).  A slot is reused only after the copy that last read it has run.
        if self._hp_ring is None:
            self._hp_ring = [[torch.empty(8, dtype=torch.float32).pin_memory(), None] for _ in range(8)]
        slot = self._hp_ring[self.step_count % len(self._hp_ring)]
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        else:
            slot[1].synchronize()
        slot[0].numpy()[:] = self.hyper(grad_scale)
        self.hp.copy_(slot[0], non_blocking=True)
        slot[1].record(torch.cuda.current_stream(self.hp.device))

    def state_dict(self):
        return {"step": self.step_count, "param_groups": [dict(g) for g in self.param_groups],
                "exp_avg": None if self.exp_avg is None else self.exp_avg.cpu(),
                "exp_avg_sq": None if self.exp_avg_sq is None else self.exp_avg_sq.cpu()}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]
        if sd.get("exp_avg") is not None:
            self._ensure()
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR (recursive form), eta_min = 0 by default."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.optimizer, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = 0

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self):
        self.last_epoch += 1
        e, T, m = self.last_epoch, self.T_max, self.eta_min
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            if (e - 1 - T) % (2 * T) == 0:
                g["lr"] = g["lr"] + (base - m) * (1 - math.cos(math.pi / T)) / 2
            else:
                g["lr"] = (1 + math.cos(math.pi * e / T)) / (1 + math.cos(math.pi * (e - 1) / T)) * (g["lr"] - m) + m

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs)}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self.base_lrs = list(sd["base_lrs"])
