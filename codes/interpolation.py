"""Alias of ladder_latent_data_distribution_modelling_amd/codes/interpolation.py (shortest-likely-path interpolation)."""
from ladder_latent_data_distribution_modelling_amd.codes.interpolation import *  # noqa: F401,F403
from ladder_latent_data_distribution_modelling_amd.codes.interpolation import SLPInterpolator, path_terms  # noqa: F401
