"""Module name kept for drop-in imports: `from velocycle_amd import velocity_inference_model`."""
from .fit_models import VelocityFitModel  # noqa: F401
from .preprocessing import (velocity_latent_variable_guide, velocity_latent_variable_guide_LRMN,  # noqa: F401
                            velocity_latent_variable_model, velocity_latent_variable_model_LRMN)
