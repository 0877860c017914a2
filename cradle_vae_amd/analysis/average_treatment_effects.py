from ..ate import estimate_model_average_treatment_effect  # noqa: F401
