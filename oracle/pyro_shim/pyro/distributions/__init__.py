"""`pyro.distributions` subset built on torch.distributions (+ Delta, GammaPoisson restated)."""
import torch
import torch.distributions as td
from torch.distributions import constraints  # noqa: F401  (pyro.distributions.constraints)


class _Mixin:
    def to_event(self, n=None):
        if n is None:
            n = len(self.batch_shape)
        if n == 0:
            return self
        return Independent(self, n)


class Independent(_Mixin, td.Independent):
    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        base = self.base_dist.expand(batch_shape + self.event_shape[: self.reinterpreted_batch_ndims])
        return Independent(base, self.reinterpreted_batch_ndims)


class Normal(_Mixin, td.Normal):
    pass


class Gamma(_Mixin, td.Gamma):
    pass


class Poisson(_Mixin, td.Poisson):
    pass


class Uniform(_Mixin, td.Uniform):
    pass


class Beta(_Mixin, td.Beta):
    pass


class Bernoulli(_Mixin, td.Bernoulli):
    pass


class MultivariateNormal(_Mixin, td.MultivariateNormal):
    pass


class LowRankMultivariateNormal(_Mixin, td.LowRankMultivariateNormal):
    pass


class Delta(_Mixin, td.Distribution):
    """pyro.distributions.Delta: point mass; log_prob = log(x == v) + log_density; reparameterised."""
    has_rsample = True
    arg_constraints = {}
    support = constraints.real

    def __init__(self, v, log_density=0.0, event_dim=0, validate_args=None):
        self.v = v
        self.log_density = log_density
        bd = v.dim() - event_dim
        super().__init__(v.shape[:bd], v.shape[bd:], validate_args=False)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        return Delta(self.v.expand(batch_shape + self.event_shape), self.log_density,
                     len(self.event_shape))

    def rsample(self, sample_shape=torch.Size()):
        shape = torch.Size(sample_shape) + self.v.shape
        return self.v.expand(shape)

    sample = rsample

    def log_prob(self, x):
        v = self.v.expand(self.batch_shape + self.event_shape)
        lp = (x == v).type(x.dtype).log()
        if len(self.event_shape):
            lp = lp.sum(tuple(range(-len(self.event_shape), 0)))
        return lp + self.log_density


def _log_beta(x, y):
    return torch.lgamma(x) + torch.lgamma(y) - torch.lgamma(x + y)


class GammaPoisson(_Mixin, td.Distribution):
    """pyro.distributions.GammaPoisson(concentration, rate) (pyro/distributions/conjugate.py):
    log_prob(k) = -log_beta(c, k+1) - log(c+k) + c*log(rate) - (c+k)*log(1+rate)."""
    arg_constraints = {"concentration": constraints.positive, "rate": constraints.positive}
    support = constraints.nonnegative_integer

    def __init__(self, concentration, rate, validate_args=None):
        concentration, rate = td.utils.broadcast_all(concentration, rate)
        self.concentration, self.rate = concentration, rate
        super().__init__(concentration.shape, validate_args=False)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        return GammaPoisson(self.concentration.expand(batch_shape), self.rate.expand(batch_shape))

    def sample(self, sample_shape=torch.Size()):
        rate = td.Gamma(self.concentration, self.rate).sample(sample_shape)
        return torch.poisson(rate)

    def log_prob(self, value):
        post = self.concentration + value
        return (-_log_beta(self.concentration, value + 1) - post.log()
                + self.concentration * self.rate.log() - post * (1 + self.rate).log())
