from .containers import Cycle, reorder  # noqa: F401
