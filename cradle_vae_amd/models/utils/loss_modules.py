from ...loss_module import PerturbationPlatedELBOLossModule  # noqa: F401
