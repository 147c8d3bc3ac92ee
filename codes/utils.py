"""Drop-in alias of the reference's codes/utils.py surface (implementation: ladder_latent_data_distribution_modelling_amd/codes/utils.py)."""
from ladder_latent_data_distribution_modelling_amd.codes.utils import *  # noqa: F401,F403
