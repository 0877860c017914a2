"""Class namespace the config resolves names in -- mirror of ``cradle_vae.models``
(cradle_vae/models/__init__.py:1-8)."""
from ..guide import cradle_vae_CorrelatedNormalGuide  # noqa: F401
from ..loss_module import (  # noqa: F401
    cradle_vae_CustomReweightedELBOLossModule,
    cradle_vae_ELBOLossModule,
    cradle_vae_IWELBOLossModule,
)
from ..model import cradle_vae_Model  # noqa: F401
from ..predictor import cradle_vae_Predictor  # noqa: F401
