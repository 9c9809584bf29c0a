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
    _KEEP = ("__spec__", "__loader__", "__package__", "__name__", "__file__", "__path__", "__cached__")

    def __init__(self, module):
        self._module = module
        self._saved = {}

    def create_module(self, spec):
        # the real module object: state is shared, not duplicated.  importlib's module_from_spec() now overwrites its
        # import attributes with the ALIAS spec (name repconc.X, this loader); they are put back in exec_module so that
        # relative imports inside repconc_amd.X, importlib.reload and spec-based tooling keep seeing the real module
        self._saved = {k: getattr(self._module, k) for k in self._KEEP if hasattr(self._module, k)}
        return self._module

    def exec_module(self, module):
        for k, v in self._saved.items():
            try:
                setattr(module, k, v)
            except (AttributeError, TypeError):
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
