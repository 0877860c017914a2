from ...predictor import PerturbationPlatedPredictor  # noqa: F401
