"""Mirror of ``cradle_vae.models.utils``: the same module names re-exporting this build's classes."""
