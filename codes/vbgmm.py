"""Alias of ladder_latent_data_distribution_modelling_amd/codes/vbgmm.py (device-resident Bayesian GMM fit)."""
from ladder_latent_data_distribution_modelling_amd.codes.vbgmm import *  # noqa: F401,F403
from ladder_latent_data_distribution_modelling_amd.codes.vbgmm import DeviceBayesianGaussianMixture  # noqa: F401
