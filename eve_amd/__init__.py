"""eve_amd -- MI355X-native implementation of EVE's modality-aware MoE layer.

Public names mirror the reference (models/layer.py, models/sharded_moe.py,
models/experts.py, models/multiway_transformer.py:44-55):

    from eve_amd import MoE, MOELayer, TopKGate, Experts, MultiwayTransformerMLP

All computation happens in hand-written HIP kernels (eve_amd/csrc, gfx950) reached through
the C ABI of libeve_moe.so (include/eve_moe.h).
"""
from .encoder import (Attention, MultiwayTransformerEncoder, MultiwayTransformerEncoderLayer,
                      MultiwayTransformerEnocderLayer, eve_base_encoder)
from .experts import ACT2FN, Experts, MultiwayTransformerMLP, QuickGELU
from .graph import GraphedTrainStep
from .layer import MoE
from .multiway import ModalitySplitFFN
from . import retrieval
from .sharded_moe import DeviceNoise, MOELayer, TopKGate, gumbel_rsample, top2gating

__all__ = ["Attention", "DeviceNoise", "MultiwayTransformerEncoder", "MultiwayTransformerEncoderLayer", "MultiwayTransformerEnocderLayer",
           "eve_base_encoder", "MoE", "GraphedTrainStep", "ModalitySplitFFN", "MOELayer", "TopKGate", "Experts", "MultiwayTransformerMLP", "QuickGELU", "ACT2FN",
           "gumbel_rsample", "top2gating", "retrieval"]
