"""Tiny stand-in for the mmcv registries the reference builds the hot path through
(mmdet3d/models/registry.py, mmdet3d/models/builder.py:56-58, mmcv.cnn build_conv_layer /
build_norm_layer).  mmcv/mmdet are not installed on the build or GPU boxes; the same `type=`
strings and dict configs work here, and if mmcv is importable the classes are registered there
too (force=True, as mmdet3d/ops/spconv/overwrite_spconv/write_spconv2.py:20-36 does)."""
import inspect

from torch import nn


class Registry:

    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, module=None, force=False):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        t = cfg.pop('type')
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f'{t} is not in the {self.name} registry')
        return cls(**cfg)


CONV_LAYERS = Registry('conv layer')
NORM_LAYERS = Registry('norm layer')
MIDDLE_ENCODERS = Registry('middle_encoder')
VOXEL_ENCODERS = Registry('voxel_encoder')

NORM_LAYERS.register_module('BN1d', module=nn.BatchNorm1d)
NORM_LAYERS.register_module('BN', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('BN2d', module=nn.BatchNorm2d)


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer: cfg = dict(type='SubMConv3d', indice_key=...)."""
    cfg = dict(cfg)
    cls = CONV_LAYERS.get(cfg.pop('type'))
    if cls is None:
        raise KeyError('unknown conv layer type')
    return cls(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=''):
    """mmcv.cnn.build_norm_layer -> (name, layer); abbreviation 'bn' as mmcv infers for BN*."""
    cfg = dict(cfg)
    cls = NORM_LAYERS.get(cfg.pop('type'))
    if cls is None:
        raise KeyError('unknown norm layer type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return 'bn' + str(postfix), layer


def _populate():
    """Import the modules whose classes register themselves (mmdet3d does this in
    mmdet3d/models/__init__.py), so a `type=` string resolves whatever was imported before."""
    from . import (pillar_scatter, second, sparse_encoder, sparse_unet, spconv,  # noqa: F401
                   voxel_encoder)


def build_middle_encoder(cfg):
    _populate()
    return MIDDLE_ENCODERS.build(cfg)


def build_voxel_encoder(cfg):
    _populate()
    return VOXEL_ENCODERS.build(cfg)


def build_backbone(cfg):
    _populate()
    from .second import BACKBONES
    return BACKBONES.build(cfg)


def build_neck(cfg):
    _populate()
    from .second import NECKS
    return NECKS.build(cfg)


def register_with_mmcv():
    """Mirror the registrations into mmcv/mmdet3d registries when those packages exist."""
    try:
        from mmcv.cnn import CONV_LAYERS as MMCV_CONV  # noqa: WPS433
    except Exception:
        return False
    for key, cls in CONV_LAYERS._modules.items():
        if inspect.isclass(cls):
            MMCV_CONV.register_module(key, force=True, module=cls)
    return True
