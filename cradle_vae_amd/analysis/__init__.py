"""Mirror of ``cradle_vae.analysis``."""
