"""unibev_amd — MI355X-native (gfx950) implementation of UniBEV's BEV-encoder hot path.

Importing the package registers the reference's registry keys (``UniBEVTransformer``,
``ImgEncoder``/``PtsEncoder``, ``ImgLayer``/``PtsLayer``, ``SpatialCrossAttentionImg``/``Pts``,
``MSDeformableAttention3DImg``/``Pts``, ``MultiScaleDeformableAttention``, ...) in
``unibev_amd.registry``.  Compute runs only through ``libunibev_hip.so`` (``unibev_amd._lib``);
there is no CPU fallback.
"""
from . import registry  # noqa: F401
from . import modules  # noqa: F401
from .registry import (ATTENTION, TRANSFORMER, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE,  # noqa
                       build_attention, build_from_cfg, build_transformer,
                       build_transformer_layer_sequence, load_config)

__version__ = '0.1.0'
