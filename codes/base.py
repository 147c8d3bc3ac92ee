"""Drop-in alias of the reference's codes/base.py surface (implementation: ladder_latent_data_distribution_modelling_amd/codes/base.py)."""
from ladder_latent_data_distribution_modelling_amd.codes.base import *  # noqa: F401,F403
