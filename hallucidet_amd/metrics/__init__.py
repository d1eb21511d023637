from .metrics import Detection, MeanAveragePrecision, MAP  # noqa: F401
