"""Drop-in alias of the reference's codes/models.py surface (implementation: ladder_latent_data_distribution_modelling_amd/codes/models.py)."""
from ladder_latent_data_distribution_modelling_amd.codes.models import *  # noqa: F401,F403
