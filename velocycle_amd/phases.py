from .containers import Phases  # noqa: F401
