"""The registry surface the reference's configs and plugin code rely on, without mmcv/mmdet:
registry objects under the names the reference imports (mmcv.cnn.bricks.registry,
mmdet.models, mmdet.core.bbox.builder), `build_from_cfg`, and a `Config.fromfile` that executes
an mmcv-style python config (projects/configs/*.py load unchanged)."""
import copy
import os


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args or None)

    def __contains__(self, key):
        return key in self.module_dict

    def __repr__(self):
        return f"Registry({self.name}, {sorted(self.module_dict)})"


def build_from_cfg(cfg, registry, default_args=None):
    """mmcv.utils.build_from_cfg: cfg is a dict with a 'type' key naming a registered class."""
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
    args = copy.copy(dict(cfg))
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**args)


ATTENTION = Registry("attention")
PLUGIN_LAYERS = Registry("plugin layer")
POSITIONAL_ENCODING = Registry("position encoding")
FEEDFORWARD_NETWORK = Registry("feed-forward network")
NORM_LAYERS = Registry("norm layer")
TRANSFORMER_LAYER = Registry("transformerLayer")
TRANSFORMER_LAYER_SEQUENCE = Registry("transformer-layers sequence")
DETECTORS = Registry("detector")
HEADS = Registry("head")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
LOSSES = Registry("loss")
BBOX_SAMPLERS = Registry("bbox_sampler")
BBOX_CODERS = Registry("bbox_coder")
BBOX_ASSIGNERS = Registry("bbox_assigner")


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


class Config(ConfigDict):
    """Minimal mmcv.Config: `Config.fromfile(path)` executes the python file and keeps its
    public top-level names; `merge_from_dict({'model.head.x': v})` mirrors --cfg-options
    (tools/test.py:77-87)."""

    @staticmethod
    def fromfile(path):
        scope = {"__file__": os.path.abspath(path)}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), scope)
        import types
        cfg = Config()
        for k, v in scope.items():
            if k.startswith("__") or isinstance(v, types.ModuleType) or callable(v):
                continue
            cfg[k] = _wrap(v)
        cfg["filename"] = path
        return cfg

    def merge_from_dict(self, options):
        for key, val in options.items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = _wrap(val)
        return self
