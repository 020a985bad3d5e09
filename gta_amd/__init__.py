"""gta_amd -- MI355X-native geometric-transform attention (GTA).

Drop-in for the one hot path of autonomousvision/gta: ``source/utils/gta.py`` +
the ``Attention``/``Transformer`` blocks of ``source/layers.py`` + the rep builders of
``source/encoder.py`` / ``source/decoder.py``.  Host code is PyTorch-ROCm; the operator itself
is hand-written HIP for gfx950 behind the C ABI in ``include/gta_hip.h``.
"""
from .gta import (multihead_geometric_transform_attention, make_2dcoord, make_SO2mats,  # noqa: F401
                  pack_reps, gta_attention, multihead_vecrep_attention, attention_map)
from .layers import Attention, Transformer, PreNorm, FeedForward, JaxLinear, ViTLinear  # noqa: F401
from .reps import pre_compute_reps_encoder, pre_compute_reps_decoder  # noqa: F401
from .srt import ImprovedSRTEncoder, ImprovedSRTDecoder, TransformingSRT  # noqa: F401
from .image2d import GTA2DTransformer  # noqa: F401

__version__ = "0.1.0"
