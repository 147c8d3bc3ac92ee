"""Alias of ladder_latent_data_distribution_modelling_amd/codes/session.py (`sess.run(fetches, feed_dict)` facade)."""
from ladder_latent_data_distribution_modelling_amd.codes.session import *  # noqa: F401,F403
from ladder_latent_data_distribution_modelling_amd.codes.session import Session, Handle, attach_handles  # noqa: F401
