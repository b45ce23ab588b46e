from .containers import AngularSpeed  # noqa: F401
