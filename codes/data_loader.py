"""Drop-in alias of the reference's codes/data_loader.py surface (implementation: ladder_latent_data_distribution_modelling_amd/codes/data_loader.py)."""
from ladder_latent_data_distribution_modelling_amd.codes.data_loader import *  # noqa: F401,F403
