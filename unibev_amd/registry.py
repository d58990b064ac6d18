"""mmcv-style registries and config loading for the plugin boundary.

The reference selects every component by a registry ``type`` string plus constructor kwargs
(SURVEY.md section 8(b)); its shipped config files must build unchanged.  This module provides
the registries under the names the reference imports (``ATTENTION``, ``TRANSFORMER_LAYER``,
``TRANSFORMER_LAYER_SEQUENCE`` from mmcv.cnn.bricks.registry, ``TRANSFORMER`` from
mmdet.models.utils.builder, ``HEADS`` / ``DETECTORS`` from mmdet) and a loader for mmcv
python-file configs (``_base_`` inheritance included).
"""
import copy
import os


class Registry:
    """name -> class table with the ``@REG.register_module()`` decorator protocol of mmcv."""

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return key in self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        if name is not None and not isinstance(name, (str, list, tuple)):
            raise TypeError(f'name must be a str or a sequence of str, got {type(name)}')

        def _register(cls):
            names = [cls.__name__] if name is None else ([name] if isinstance(name, str) else name)
            for n in names:
                if not force and n in self._module_dict:
                    raise KeyError(f'{n} is already registered in {self._name}')
                self._module_dict[n] = cls
            return cls

        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate ``registry[cfg['type']](**rest)``; same error behaviour as mmcv."""
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    args = copy.copy(dict(cfg))
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif isinstance(obj_type, type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    return obj_cls(**args)


ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('feed-forward Network')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
TRANSFORMER = Registry('Transformer')
POSITIONAL_ENCODING = Registry('position encoding')
NORM_LAYERS = Registry('norm layer')
HEADS = Registry('head')
DETECTORS = Registry('detector')
VOXEL_ENCODERS = Registry('voxel_encoder')
MIDDLE_ENCODERS = Registry('middle_encoder')
BACKBONES = Registry('backbone')
NECKS = Registry('neck')


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


def build_positional_encoding(cfg, default_args=None):
    return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)


# ----------------------------------------------------------------------------------------------- configs
class ConfigDict(dict):
    """dict with attribute access (mmcv.ConfigDict subset)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _to_cfg(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cfg(v) for v in obj)
    return obj


def _merge(base, child):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v
    return out


def load_config(path):
    """Execute an mmcv python-file config and return its public variables as a ConfigDict.
    ``_base_`` (str or list of str, relative to the file) is merged first, child keys win."""
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    scope = {'__file__': path}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), scope)
    public = {k: v for k, v in scope.items()
              if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = public.pop('_base_', None)
    merged = {}
    if bases:
        for b in ([bases] if isinstance(bases, str) else bases):
            merged = _merge(merged, load_config(os.path.join(os.path.dirname(path), b)))
    return _to_cfg(_merge(merged, public))
