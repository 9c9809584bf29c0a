"""`repconc` import surface served by repconc_amd — opt in by putting this directory on the path:

    PYTHONPATH=/path/to/repo/compat:/path/to/repo  python evaluate/run_repconc_eval.py ...

Every `repconc.X` module resolves to THE SAME module object as `repconc_amd.X` (a meta-path alias, no second copy of
any state), so the reference's callers keep their imports:

    from repconc.models.repconc import RepCONC                                   (finetune_repconc.py, run_warmup.py)
    from repconc.models.repconc.evaluate_repconc import ModelArguments, EvalArguments, RepCONCEvaluater, \\
        initialize_index, add_docs, from_pq_to_ivfpq, load_index_to_gpu, encode_corpus, encode_query, \\
        search, batch_search                                                     (run_repconc_eval.py:16-24)
    from repconc.models.dense import AutoDense;  from repconc.train.run_warmup import warmup_from_embeds
    from repconc.utils.eval_utils import load_corpus, load_queries, TextDataset, get_collator_func

Not on the path by default: a checkout of the reference next to this repo keeps importing its own `repconc`.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import repconc_amd as _real

__version__ = getattr(_real, "__version__", "0")


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module                      # the real module object: state is shared, not duplicated

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    _prefix = __name__ + "."

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self._prefix):
            return None
        try:
            module = importlib.import_module("repconc_amd." + fullname[len(self._prefix):])
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(module), is_package=hasattr(module, "__path__"))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
