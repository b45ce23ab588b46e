"""`pyro.optim.ClippedAdam` restated (pyro/optim/clipped_adam.py + PyroOptim: one optimiser state per
parameter tensor, lr *= lrd before every update, elementwise gradient clamp to +-clip_norm)."""
import math
import torch


class ClippedAdam:
    def __init__(self, optim_args):
        self.pt_optim_args = dict(optim_args)
        a = self.pt_optim_args
        self.lr0 = a.get("lr", 1e-3)
        self.betas = tuple(a.get("betas", (0.9, 0.999)))
        self.eps = a.get("eps", 1e-8)
        self.weight_decay = a.get("weight_decay", 0.0)
        self.clip_norm = a.get("clip_norm", 10.0)
        self.lrd = a.get("lrd", 1.0)
        self.state = {}

    def __call__(self, params):
        for p in params:
            st = self.state.setdefault(id(p), {"lr": self.lr0, "step": 0, "m": None, "v": None, "p": p})
            st["lr"] *= self.lrd
            if p.grad is None:
                continue
            g = p.grad.data
            g.clamp_(-self.clip_norm, self.clip_norm)
            if st["m"] is None:
                st["m"] = torch.zeros_like(g)
                st["v"] = torch.zeros_like(g)
            b1, b2 = self.betas
            st["step"] += 1
            if self.weight_decay != 0:
                g = g.add(p.data, alpha=self.weight_decay)
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = st["v"].sqrt().add_(self.eps)
            bc1 = 1 - b1 ** st["step"]
            bc2 = 1 - b2 ** st["step"]
            step_size = st["lr"] * math.sqrt(bc2) / bc1
            p.data.addcdiv_(st["m"], denom, value=-step_size)


class Adam:
    """`pyro.optim.Adam` = PyroOptim(torch.optim.Adam): one torch optimiser per parameter tensor (created when the tensor is
    first seen), stepped with the gradients SVI left on the parameters."""

    def __init__(self, optim_args):
        self.pt_optim_args = dict(optim_args)
        self.pt_optim_constructor = torch.optim.Adam
        self.state = {}

    def __call__(self, params):
        for p in params:
            o = self.state.get(id(p))
            if o is None:
                o = self.state[id(p)] = (torch.optim.Adam([p], **self.pt_optim_args), p)
            if p.grad is not None:
                o[0].step()
