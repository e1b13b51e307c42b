"""causal_eeg_vae_amd — MI355X-native hot path of the event-graph VGAE + VQ training step.

Python host API over libceeg.so (hand-written gfx950 HIP kernels, C-ABI in include/ceeg.h).
"""
from ._lib import CeegError, LIB_PATH, lib  # noqa: F401
from .quantizer import VectorQuantizer  # noqa: F401
from .models import (  # noqa: F401
    MAX_LOGSTD, CsrPair, EventGraph, GCNConv, InnerProductDecoder, VariationalGCNEncoder, VGAE, VGAEVQ,
    ConditionalVGAEVQ, GATConv, negative_sampling, sigma_params,
)
from .conditioner import GraphConditioner  # noqa: F401
from .collate import CollatedBatch, MinibatchStream, segment_ids  # noqa: F401
