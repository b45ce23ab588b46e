"""`velocycle_amd.optim.ClippedAdam` -- constructed like `pyro.optim.ClippedAdam({...})`."""
from .svi import ClippedAdam  # noqa: F401
