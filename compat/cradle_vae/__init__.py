"""Drop-in alias: with ``<repo>/compat`` (and ``<repo>``) on PYTHONPATH, ``import cradle_vae``
resolves to the MI355X build, so code written against the reference's package paths
(``cradle_vae.models.*``, ``cradle_vae.models.utils.perturbation_lightning_module``,
``cradle_vae.models.utils.predictor``, ``cradle_vae.analysis.average_treatment_effects``,
``cradle_vae.data.utils.perturbation_dataset`` ...) runs on the HIP path unchanged.

Dataset-specific data modules (``cradle_vae.data.{norman,adamson,dixit,replogle}``: scanpy /
rapids pipelines over downloaded .h5ad files) are not part of this build; ``cradle_vae.data``
exposes the in-memory dataset classes and ``ObservationNormalizationStatistics`` only.
"""
import importlib
import sys
import types

_ALIASES = {
    "cradle_vae.models": "cradle_vae_amd.models",
    "cradle_vae.models.cradle_vae": "cradle_vae_amd.models",
    "cradle_vae.models.cradle_vae.model": "cradle_vae_amd.model",
    "cradle_vae.models.cradle_vae.loss_module": "cradle_vae_amd.loss_module",
    "cradle_vae.models.cradle_vae.predictor": "cradle_vae_amd.predictor",
    "cradle_vae.models.cradle_vae.guides.correlated_normal_guide": "cradle_vae_amd.guide",
    "cradle_vae.models.utils": "cradle_vae_amd.models.utils",
    "cradle_vae.models.utils.perturbation_lightning_module": "cradle_vae_amd.models.utils.perturbation_lightning_module",
    "cradle_vae.models.utils.predictor": "cradle_vae_amd.models.utils.predictor",
    "cradle_vae.models.utils.loss_modules": "cradle_vae_amd.models.utils.loss_modules",
    "cradle_vae.models.utils.mlp": "cradle_vae_amd.models.utils.mlp",
    "cradle_vae.models.utils.normalization": "cradle_vae_amd.models.utils.normalization",
    "cradle_vae.models.utils.gumbel_softmax_bernoulli": "cradle_vae_amd.models.utils.gumbel_softmax_bernoulli",
    "cradle_vae.analysis": "cradle_vae_amd.analysis",
    "cradle_vae.analysis.average_treatment_effects": "cradle_vae_amd.analysis.average_treatment_effects",
    "cradle_vae.data.utils.perturbation_dataset": "cradle_vae_amd.data",
    "cradle_vae.data.utils.batch_statistics": "cradle_vae_amd.data",
    "cradle_vae.data.utils.anndata": "cradle_vae_amd.analysis.metrics",
}

for _alias, _target in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_target)

# packages without a one-to-one counterpart
_guides = types.ModuleType("cradle_vae.models.cradle_vae.guides")
_guides.correlated_normal_guide = sys.modules["cradle_vae.models.cradle_vae.guides.correlated_normal_guide"]
sys.modules["cradle_vae.models.cradle_vae.guides"] = _guides

_data = types.ModuleType("cradle_vae.data")
_data_utils = types.ModuleType("cradle_vae.data.utils")
_datamodule = types.ModuleType("cradle_vae.data.utils.perturbation_datamodule")
_datamodule.ObservationNormalizationStatistics = importlib.import_module(
    "cradle_vae_amd.guide").ObservationNormalizationStatistics
_data_utils.perturbation_dataset = sys.modules["cradle_vae.data.utils.perturbation_dataset"]
_data_utils.perturbation_datamodule = _datamodule
_data_utils.batch_statistics = sys.modules["cradle_vae.data.utils.batch_statistics"]
_data_utils.anndata = sys.modules["cradle_vae.data.utils.anndata"]
_data.utils = _data_utils
sys.modules["cradle_vae.data"] = _data
sys.modules["cradle_vae.data.utils"] = _data_utils
sys.modules["cradle_vae.data.utils.perturbation_datamodule"] = _datamodule

models = sys.modules["cradle_vae.models"]
analysis = sys.modules["cradle_vae.analysis"]
data = _data
