from .finetune_jpq import JPQ, jpq_step_end  # noqa: F401
