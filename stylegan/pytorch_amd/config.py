"""Configuration tree of the training driver without yacs (not installed on the MI355X image).

``cfg`` mirrors the reference's global defaults (reference config.py:12-92) node for node, so a reference YAML
(``configs/*.yaml``) merges into it unchanged and ``train.py``'s accesses (``opt.model.gen``, ``opt.sched.epochs``,
``**opt.model.g_optim`` ...) work as they do with yacs: a node is a ``dict`` with attribute access, ``merge_from_file``,
``merge_from_list`` and ``freeze``.  Like yacs, merging rejects keys the defaults do not have and values of a different type.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError(f"attempted to set {name} on a frozen CfgNode")
        self[name] = value

    def freeze(self, flag=True):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)

    def defrost(self):
        self.freeze(False)

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        return out

    @staticmethod
    def _coerce(old, new, path):
        """yacs' type rule: same type, or one of the casts it allows (tuple<->list, int->float, str tuple literals)."""
        if isinstance(new, str) and not isinstance(old, str):
            try:
                new = ast.literal_eval(new)
            except (ValueError, SyntaxError):
                pass
        if isinstance(old, str) and not isinstance(new, str):
            # e.g. ``device_id: ('3')`` in the reference YAMLs is the plain string "('3')" to PyYAML; yacs literal_evals
            # it to '3'.  Anything that is not a string for a string default is an error, as in yacs.
            raise ValueError(f"type mismatch for {path}: {type(old).__name__} vs {type(new).__name__}")
        if isinstance(old, str) and isinstance(new, str):
            try:
                lit = ast.literal_eval(new)
                if isinstance(lit, str):
                    new = lit
            except (ValueError, SyntaxError):
                pass
            return new
        if type(old) is type(new) or old is None:
            return new
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        raise ValueError(f"type mismatch for {path}: {type(old).__name__} vs {type(new).__name__}")

    def _merge(self, other, path=""):
        for k, v in other.items():
            full = f"{path}.{k}" if path else k
            if k not in self:
                raise KeyError(f"non-existent config key: {full}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{full} is a node")
                self[k]._merge(v, full)
            else:
                dict.__setitem__(self, k, CfgNode._coerce(self[k], v, full))

    def merge_from_file(self, path):
        if self.is_frozen():
            raise AttributeError("merge_from_file on a frozen CfgNode")
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_other_cfg(self, other):
        self._merge(other)

    def merge_from_list(self, items):
        """['sched.epochs', [1, 1], 'output_dir', '/tmp/x', ...] -- the non-YAML override of SURVEY.md 8b."""
        assert len(items) % 2 == 0
        for key, val in zip(items[0::2], items[1::2]):
            node, parts = self, key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"non-existent config key: {key}")
            dict.__setitem__(node, parts[-1], CfgNode._coerce(node[parts[-1]], val, key))


def default_cfg():
    """The reference's defaults, reference config.py:12-92."""
    return CfgNode({
        "output_dir": "", "device": "cuda", "device_id": "0",                                     # :14-16
        "structure": "fixed", "conditional": False, "n_classes": 0, "loss": "logistic", "drift": 0.001,   # :18-22
        "d_repeats": 1, "use_ema": True, "ema_decay": 0.999,                                      # :23-25
        "num_works": 4, "num_samples": 36, "feedback_factor": 10, "checkpoint_factor": 10,        # :27-30
        "sched": {                                                                                # :35-42 (depth 9 example)
            "epochs": [4, 4, 4, 4, 8, 16, 32, 64, 64],
            "batch_sizes": [128, 128, 128, 64, 32, 16, 8, 4, 2],
            "fade_in_percentage": [50, 50, 50, 50, 50, 50, 50, 50, 50],
        },
        "dataset": {"img_dir": "", "folder": True, "resolution": 128, "channels": 3},            # :51-55
        "model": {
            "gen": {"latent_size": 512, "mapping_layers": 4, "blur_filter": [1, 2, 1], "truncation_psi": 0.7,
                    "truncation_cutoff": 8},                                                      # :61-67
            "dis": {"use_wscale": True, "blur_filter": [1, 2, 1]},                                # :72-74
            "g_optim": {"learning_rate": 0.003, "beta_1": 0, "beta_2": 0.99, "eps": 1e-8},        # :79-83
            "d_optim": {"learning_rate": 0.003, "beta_1": 0, "beta_2": 0.99, "eps": 1e-8},        # :88-92
        },
    })


cfg = default_cfg()
