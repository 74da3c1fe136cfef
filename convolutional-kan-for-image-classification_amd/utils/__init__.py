from .regularization import L1, L2, WeightDecay   # noqa: F401
