"""Drop-in alias of the reference's codes/trainers.py surface (implementation: ladder_latent_data_distribution_modelling_amd/codes/trainers.py)."""
from ladder_latent_data_distribution_modelling_amd.codes.trainers import *  # noqa: F401,F403
