from .modeling_repconc import RepCONC, QuantizeOutput, sinkhorn_algorithm, decode  # noqa: F401
