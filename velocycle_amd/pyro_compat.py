"""The few `pyro` module-level names the tutorials use around fit(): a parameter store holding the
fitted variational parameters under Pyro's names and shapes, `clear_param_store()`, `param(name)`."""
from __future__ import annotations

from typing import Dict

import torch

_STORE: Dict[str, torch.Tensor] = {}


def clear_param_store():
    _STORE.clear()


def get_param_store():
    return _STORE


def param(name):
    return _STORE[name]
