"""`pyro.infer` subset: Trace_ELBO(num_particles=K), SVI, Predictive, config_enumerate (no-op)."""
import torch
from ..runtime import run_traced, _Wrapped
from . import autoguide  # noqa: F401


def config_enumerate(fn=None, **kw):
    if fn is None:
        return lambda f: f
    return fn


class Trace_ELBO:
    def __init__(self, num_particles=1, max_plate_nesting=float("inf"), **kw):
        self.num_particles = int(num_particles)
        self.max_plate_nesting = max_plate_nesting

    def _guess_max_plate_nesting(self, model, guide, args, kwargs):
        # pyro ELBO._guess_max_plate_nesting: one extra guide + replayed-model execution
        gt, _ = run_traced(guide, args, kwargs)
        run_traced(model, args, kwargs, replay=gt)
        self.max_plate_nesting = 5

    def _traces(self, model, guide, args, kwargs):
        if self.max_plate_nesting == float("inf"):
            with torch.no_grad():
                self._guess_max_plate_nesting(model, guide, args, kwargs)
        gt, gp = run_traced(guide, args, kwargs)
        mt, mp = run_traced(model, args, kwargs, replay=gt)
        return mt, gt, {**gp, **mp}

    def differentiable_loss_and_params(self, model, guide, *args, **kwargs):
        mt, gt, params = self._traces(model, guide, args, kwargs)
        elbo = mt.log_prob_sum() - gt.log_prob_sum()
        return -elbo, params, mt, gt

    def loss_and_grads(self, model, guide, *args, **kwargs):
        # pyro-ppl 1.8.6 Trace_ELBO.loss_and_grads (vectorize_particles=False): the particles are drawn one after the other
        # (num_particles sequential guide + replayed-model executions), loss = mean of the particles' losses, and every
        # particle's surrogate / num_particles is back-propagated, i.e. the gradients are the mean over particles
        loss, params = 0.0, {}
        for _ in range(self.num_particles):
            lp, pp, _, _ = self.differentiable_loss_and_params(model, guide, *args, **kwargs)
            (lp / self.num_particles).backward()
            loss += lp.item() / self.num_particles
            params.update(pp)
        return loss, params


TraceEnum_ELBO = Trace_ELBO


class SVI:
    def __init__(self, model, guide, optim, loss, **kw):
        self.model, self.guide, self.optim, self.loss = model, guide, optim, loss

    def step(self, *args, **kwargs):
        loss, params = self.loss.loss_and_grads(self.model, self.guide, *args, **kwargs)
        plist = list(params.values())
        self.optim(plist)
        for p in plist:                      # pyro.infer.util.zero_grads
            if p.grad is not None:
                p.grad = torch.zeros_like(p.grad)
        return loss


class Predictive:
    """Sequential (parallel=False) Predictive: n guide draws, then the model replayed on each."""

    def __init__(self, model, posterior_samples=None, guide=None, num_samples=None,
                 return_sites=(), parallel=False):
        self.model, self.guide, self.num_samples = model, guide, num_samples
        self.return_sites = return_sites

    def __call__(self, *args, **kwargs):
        with torch.no_grad():
            gts = []
            for _ in range(self.num_samples):
                gt, _ = run_traced(self.guide, args, kwargs)
                gts.append(gt)
            out = {}
            for gt in gts:
                samples = {k: n["value"] for k, n in gt.nodes.items() if n["type"] == "sample"}
                mt, _ = run_traced(_Wrapped(self.model, cond=samples), args, kwargs)
                for name, node in mt.nodes.items():
                    if self.return_sites:
                        if name not in self.return_sites:
                            continue
                    elif node["type"] == "sample" and node.get("is_observed") and name not in samples:
                        continue
                    out.setdefault(name, []).append(node["value"])
            return {k: torch.stack([v.detach() for v in vs]) for k, vs in out.items()}
