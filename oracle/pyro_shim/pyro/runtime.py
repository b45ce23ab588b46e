"""Effect-handling core of the shim (trace / replay / condition / block / plate)."""
import contextlib
import torch
from torch.distributions import constraints, transform_to


class ParamStore:
    def __init__(self):
        self._uncon = {}
        self._constraint = {}

    def clear(self):
        self._uncon.clear()
        self._constraint.clear()

    def __contains__(self, name):
        return name in self._uncon

    def keys(self):
        return self._uncon.keys()

    def setdefault(self, name, init, constraint):
        if name not in self._uncon:
            with torch.no_grad():
                init = init() if callable(init) else init
                u = transform_to(constraint).inv(init.detach().clone())
                u = u.contiguous().clone()
            u.requires_grad_(True)
            self._uncon[name] = u
            self._constraint[name] = constraint
        return self.get(name)

    def get(self, name):
        u = self._uncon[name]
        c = self._constraint[name]
        if c is constraints.real:
            v = u
        else:
            v = transform_to(c)(u)
        return v

    def unconstrained(self, name):
        return self._uncon[name]

    def named_parameters(self):
        """pyro-ppl 1.8.6 params/param_store.py ParamStoreDict.named_parameters: (name, UNCONSTRAINED leaf) pairs -- the one
        accessor the fixture generators use, so that they read the real library's store the same way (oracle/ref_loader.py)."""
        return self._uncon.items()

    def __getitem__(self, name):
        return self.get(name)

    def get_state(self):
        return {"params": {k: v.detach().clone() for k, v in self._uncon.items()},
                "constraints": dict(self._constraint)}

    def set_state(self, state):
        self.clear()
        for k, v in state["params"].items():
            self._uncon[k] = v.detach().clone().requires_grad_(True)
            self._constraint[k] = state["constraints"][k]


_PARAM_STORE = ParamStore()


def get_param_store():
    return _PARAM_STORE


def clear_param_store():
    _PARAM_STORE.clear()


def set_rng_seed(seed):
    torch.manual_seed(seed)


class Trace:
    def __init__(self):
        self.nodes = {}

    def add(self, name, **kw):
        if name in self.nodes:
            raise RuntimeError(f"site {name!r} appears twice in trace")
        self.nodes[name] = kw

    def log_prob_sum(self):
        tot = 0.0
        for n in self.nodes.values():
            if n["type"] == "sample":
                tot = tot + n["log_prob"].sum()
        return tot


class _Ctx:
    """One running program (a guide or a model) being traced."""

    def __init__(self, trace=None, replay=None):
        self.trace = trace
        self.replay = replay
        self.cond = {}
        self.hide = set()
        self.plates = []
        self.params = {}


_CTX_STACK = []


def _ctx():
    return _CTX_STACK[-1] if _CTX_STACK else None


@contextlib.contextmanager
def running(ctx):
    _CTX_STACK.append(ctx)
    try:
        yield ctx
    finally:
        _CTX_STACK.pop()


class plate:
    """`pyro.plate(name, size, dim=...)`: marks a batch dim as conditionally independent and
    broadcasts distributions sampled inside it to `size` along `dim` (pyro BroadcastMessenger)."""

    def __init__(self, name, size, subsample_size=None, dim=None, device=None, **kw):
        if dim is None or dim >= 0:
            raise ValueError("shim plate needs an explicit negative dim")
        self.name, self.size, self.dim = name, int(size), dim

    def __enter__(self):
        c = _ctx()
        if c is not None:
            if any(p.dim == self.dim for p in c.plates):
                raise ValueError(f"plate dim {self.dim} collides")
            c.plates.append(self)
        return torch.arange(self.size)

    def __exit__(self, *exc):
        c = _ctx()
        if c is not None:
            c.plates.remove(self)
        return False


def _broadcast_to_plates(fn, plates):
    if not plates:
        return fn
    bs = list(fn.batch_shape)
    for p in plates:
        need = -p.dim
        while len(bs) < need:
            bs.insert(0, 1)
        if bs[p.dim] == 1:
            bs[p.dim] = p.size
        elif bs[p.dim] != p.size:
            raise ValueError(f"Shape mismatch inside plate({p.name!r}) at dim {p.dim}: "
                             f"{bs[p.dim]} vs {p.size}")
    bs = torch.Size(bs)
    if bs != fn.batch_shape:
        fn = fn.expand(bs)
    return fn


def sample(name, fn, obs=None, infer=None, **kw):
    c = _ctx()
    if c is None:
        return fn.rsample() if getattr(fn, "has_rsample", False) else fn.sample()
    fn = _broadcast_to_plates(fn, c.plates)
    observed = False
    if name in c.cond:                       # poutine.condition
        value, observed = c.cond[name], True
    elif obs is not None:
        value, observed = obs, True
    elif c.replay is not None and name in c.replay.nodes and c.replay.nodes[name]["type"] == "sample":
        value = c.replay.nodes[name]["value"]
    else:
        value = fn.rsample() if getattr(fn, "has_rsample", False) else fn.sample()
    if c.trace is not None and name not in c.hide:   # poutine.block hides from the tracer only
        c.trace.add(name, type="sample", value=value, fn=fn, is_observed=observed,
                    log_prob=fn.log_prob(value))
    return value


def param(name, init_tensor=None, constraint=constraints.real, event_dim=None):
    if init_tensor is None and name not in _PARAM_STORE:
        raise KeyError(name)
    value = _PARAM_STORE.setdefault(name, init_tensor, constraint) if init_tensor is not None \
        else _PARAM_STORE.get(name)
    c = _ctx()
    if c is not None:
        c.params[name] = _PARAM_STORE.unconstrained(name)
    return value


def deterministic(name, value, event_dim=None):
    c = _ctx()
    if c is not None and c.trace is not None and name not in c.hide:
        c.trace.add(name, type="deterministic", value=value)
    return value


class _Wrapped:
    """Result of poutine.condition / poutine.block: a callable that installs extra state."""

    def __init__(self, fn, cond=None, hide=None):
        self.fn, self.cond, self.hide = fn, dict(cond or {}), set(hide or ())

    def __call__(self, *a, **k):
        c = _ctx()
        if c is None:
            with running(_Ctx()):
                return self(*a, **k)
        old_cond, old_hide = c.cond, c.hide
        c.cond = {**old_cond, **self.cond}
        c.hide = old_hide | self.hide
        try:
            return self.fn(*a, **k)
        finally:
            c.cond, c.hide = old_cond, old_hide


def run_traced(fn, args, kwargs=None, replay=None):
    ctx = _Ctx(trace=Trace(), replay=replay)
    with running(ctx):
        fn(*args, **(kwargs or {}))
    return ctx.trace, ctx.params
