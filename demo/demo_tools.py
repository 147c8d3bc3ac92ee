"""Reference import path `from demo.demo_tools import ...` -> the package's implementation."""
from ladder_latent_data_distribution_modelling_amd.demo.demo_tools import *  # noqa: F401,F403
from ladder_latent_data_distribution_modelling_amd.demo.demo_tools import (  # noqa: F401
    get_embeddings_from_val_set, define_prior_distribution, generate_prior_embeddings, plot_images_and_its_reconstruction)
