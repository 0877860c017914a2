from ...lightning_module import (  # noqa: F401
    PerturbationLightningModule,
    TrainConfigPerturbationLightningModule,
)
