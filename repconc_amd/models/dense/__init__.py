from .modeling_dense import AutoDense, BertDense, RobertaDense, DistilBertDense  # noqa: F401
