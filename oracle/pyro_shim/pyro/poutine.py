"""`pyro.poutine` subset: condition, block, trace(...).get_trace, replay."""
from .runtime import _Wrapped, run_traced


def condition(fn, data):
    return _Wrapped(fn, cond=data)


def block(fn=None, hide=None, **kw):
    if fn is None:                      # used as context manager `with poutine.block():`
        import contextlib
        return contextlib.nullcontext()
    return _Wrapped(fn, hide=hide or ())


class _Tracer:
    def __init__(self, fn):
        self.fn = fn

    def get_trace(self, *a, **k):
        tr, _ = run_traced(self.fn, a, k)
        return tr


def trace(fn):
    return _Tracer(fn)


def replay(fn, trace=None):
    def _run(*a, **k):
        tr, _ = run_traced(fn, a, k, replay=trace)
        return tr
    return _run
