"""Dense text encoders (L1 of SURVEY.md §1) — OUT of the hot-path scope: they run unchanged on
PyTorch-ROCm.  Only what `RepCONC.from_pretrained` needs is provided, with the reference's class
names, config switches (`pooling`, `similarity_metric`) and state_dict prefixes (`bert.`,
`roberta.`, `distilbert.`) so reference checkpoints load.  Semantics: models/dense/modeling_dense.py:14-135.
"""
import torch
import torch.nn.functional as F
from transformers import AutoConfig, BertModel, DistilBertModel, RobertaModel
from transformers.models.bert.modeling_bert import BertPreTrainedModel
from transformers.models.distilbert.modeling_distilbert import DistilBertPreTrainedModel
from transformers.models.roberta.modeling_roberta import RobertaPreTrainedModel


def _pool(hidden, attention_mask, how):
    if how == "mean":
        w = attention_mask.unsqueeze(-1).to(torch.float32)
        return (hidden * w).sum(1) / w.sum(1).clamp(min=1e-9)
    if how == "cls":
        return hidden[:, 0]
    raise NotImplementedError(how)


class _DenseMixin:
    backbone_attr = None

    def forward(self, input_ids, attention_mask, return_dict=False):
        out = getattr(self, self.backbone_attr)(input_ids, attention_mask, return_dict=True)
        emb = _pool(out.last_hidden_state, attention_mask, getattr(self.config, "pooling", "cls"))
        if getattr(self.config, "similarity_metric", None) == "METRIC_COS":
            emb = F.normalize(emb, p=2, dim=-1)
        if return_dict:
            out.embedding = emb
            return out
        return emb

    @property
    def language_model(self):
        return getattr(self, self.backbone_attr)


class BertDense(_DenseMixin, BertPreTrainedModel):
    backbone_attr = "bert"

    def __init__(self, config):
        BertPreTrainedModel.__init__(self, config)
        self.bert = BertModel(config, add_pooling_layer=False)


class RobertaDense(_DenseMixin, RobertaPreTrainedModel):
    backbone_attr = "roberta"

    def __init__(self, config):
        RobertaPreTrainedModel.__init__(self, config)
        self.roberta = RobertaModel(config, add_pooling_layer=False)


class DistilBertDense(_DenseMixin, DistilBertPreTrainedModel):
    backbone_attr = "distilbert"

    def __init__(self, config):
        DistilBertPreTrainedModel.__init__(self, config)
        self.distilbert = DistilBertModel(config)


class AutoDense:
    _BY_TYPE = {"bert": BertDense, "roberta": RobertaDense, "distilbert": DistilBertDense}

    @staticmethod
    def from_pretrained(model_name_or_path, config=None):
        config = config or AutoConfig.from_pretrained(model_name_or_path)
        try:
            cls = AutoDense._BY_TYPE[config.model_type]
        except KeyError:
            raise NotImplementedError(config.model_type)
        return cls.from_pretrained(model_name_or_path, config=config)
