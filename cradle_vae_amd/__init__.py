"""cradle_vae_amd: MI355X-native ELBO step and ATE predictor of CRADLE-VAE.

The package directory is ``cradle_vae_amd`` (underscore): a hyphen is not a valid Python
identifier, so ``cradle-vae_amd`` could not be imported.

Nothing here imports the HIP library eagerly; ``cradle_vae_amd.hip`` loads
``csrc/libcradle_hip.so`` on first use and raises if it is missing.
"""

__version__ = "0.1.0"
