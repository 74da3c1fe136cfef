from .dp import BucketedGradReducer   # noqa: F401
