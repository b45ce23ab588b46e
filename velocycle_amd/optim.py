"""`velocycle_amd.optim.ClippedAdam` / `Adam` -- constructed like `pyro.optim.ClippedAdam({...})` / `pyro.optim.Adam({...})`."""
from .svi import Adam, ClippedAdam  # noqa: F401
