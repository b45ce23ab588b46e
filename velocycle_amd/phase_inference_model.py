"""Module name kept for drop-in imports: `from velocycle_amd import phase_inference_model`."""
from .fit_models import PhaseFitModel  # noqa: F401
from .preprocessing import phase_latent_variable_guide, phase_latent_variable_model  # noqa: F401
