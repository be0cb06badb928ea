"""npvp_amd - MI355X-native (gfx950) implementation of NPVP's Stage-2 predictor hot path.

`from npvp_amd import Predictor, VidHRFormerEncoder, VidHRformerDecoderNAR, ...` mirrors
`from models import ...` of the reference.  All arithmetic on the path runs in the hand-written HIP
kernels of libnpvp_hip.so (npvp_amd/csrc, C ABI in include/npvp_hip.h); importing the package works
anywhere, but calling an op without the built library or without a GPU raises - there is no fallback.
"""
import os as _os

# HIP-graph replays (trainer.GraphedTrainStep) and the ROCm runtime's packet-capture path (round 6, profiles/r06_graph_alloc_hazard.txt).
# By default the runtime replays a graph from AQL packets it prepared at instantiation - the fast path (c2: 214 ms per replayed step
# against 236 without it).  On ROCm 7.2 that path computes ONE WRONG STEP, silently and independently of what the bytes are, when
# device memory that was free when the graph was instantiated is allocated and written by any kernel between two replays (a caller's
# `loss.clone()`, a new batch tensor, a metrics buffer: tools/graph_alloc_hazard.py reproduces it with a 4-byte fill); with
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 every variant of the reproducer is exact.  A library cannot know what its caller allocates, so
# the package takes the safe runtime mode unless told otherwise - the variable has to be in the environment before the HIP runtime
# initialises, i.e. import npvp_amd before the first CUDA call.  NPVP_GRAPH_PACKET_CAPTURE=1 keeps the fast path for loops that
# allocate nothing between replays (bench.py's timed loops do not, and say so in their record).
if _os.environ.get("NPVP_GRAPH_PACKET_CAPTURE", "0") != "1":
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def graph_packet_capture():
    """True when graph replays go through the runtime's packet-capture path (fast; see the hazard above)"""
    return _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1") != "0"


from .models import (Predictor, VidHRFormerEncoder, VidHRformerDecoderNAR, VidHRFormerBlockEnc, VidHRFormerBlockDecNAR,
                     SpatialLocalMultiheadAttention, MlpDWBN, MultiheadAttention, CoorGenerator, NRMLP, PosFeatFuser,
                     EventEncoder, L1Loss, Div_KL, DropPath, ResnetEncoder, ResnetDecoder, build_frozen_autoencoder, to_device_layout)
from .trainer import (FlatAdamW, predictor_train_step, full_train_step, predictor_val_step, full_val_step, cosine_warm_restarts_lr, build_predictor_from_cfg,
                      context_lists, rand_context_collate, rand_context_batch_process, vfi_batch_process,
                      save_lightning_checkpoint, load_lightning_checkpoint, GraphedTrainStep)
from . import ops
from . import metrics, data

__all__ = ["Predictor", "VidHRFormerEncoder", "VidHRformerDecoderNAR", "VidHRFormerBlockEnc", "VidHRFormerBlockDecNAR",
           "SpatialLocalMultiheadAttention", "MlpDWBN", "MultiheadAttention", "CoorGenerator", "NRMLP", "PosFeatFuser",
           "EventEncoder", "L1Loss", "Div_KL", "DropPath", "ResnetEncoder", "ResnetDecoder", "build_frozen_autoencoder", "to_device_layout", "FlatAdamW", "predictor_train_step", "full_train_step", "predictor_val_step", "full_val_step", "context_lists",
           "rand_context_collate", "rand_context_batch_process", "vfi_batch_process",
           "save_lightning_checkpoint", "load_lightning_checkpoint", "GraphedTrainStep",
           "cosine_warm_restarts_lr", "build_predictor_from_cfg", "ops", "metrics", "data"]
