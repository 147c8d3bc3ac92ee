"""Alias of ladder_latent_data_distribution_modelling_amd/codes/tf_bundle.py (TF checkpoint-v2 bundle reader/writer)."""
from ladder_latent_data_distribution_modelling_amd.codes.tf_bundle import *  # noqa: F401,F403
from ladder_latent_data_distribution_modelling_amd.codes.tf_bundle import save_checkpoint, load_checkpoint, read_index, write_index, BundleError  # noqa: F401
