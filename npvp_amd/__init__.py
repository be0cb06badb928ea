"""npvp_amd - MI355X-native (gfx950) implementation of NPVP's Stage-2 predictor hot path.

`from npvp_amd import Predictor, VidHRFormerEncoder, VidHRformerDecoderNAR, ...` mirrors
`from models import ...` of the reference.  All arithmetic on the path runs in the hand-written HIP
kernels of libnpvp_hip.so (npvp_amd/csrc, C ABI in include/npvp_hip.h); importing the package works
anywhere, but calling an op without the built library or without a GPU raises - there is no fallback.
"""
import os as _os

# HIP-graph replays (trainer.GraphedTrainStep) and the ROCm runtime's two replay modes (round 6, profiles/r06_graph_alloc_hazard.txt).
# By default the runtime replays a graph from AQL packets it prepared at instantiation (0.3 - 1.5 ms of host per replay of ~1 000
# kernels instead of 2.5 - 8 ms).  On ROCm 7.2 that mode does NOT replay a graph's MEMSET NODES reliably: once the process has issued
# other work after the instantiation - an eager hipMemsetAsync of 1 MiB is enough, tools/graph_memset_node_repro.py shows it with
# torch alone - a memset node fills part of its buffer with a stale pattern or does nothing, silently.  The step had two of them
# (the amax table npvp_split_weights_f16 cleared with hipMemsetAsync, the 4-byte semaphore of a multi-block torch reduction such as
# mean()): wrong operand scales and wrong parameters from the first, a loss scalar that is never written from the second.  With DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 the runtime marshals every node at
# launch and is exact.  The package's own step has no memset node any more (zero fills are kernels, the losses are the library's
# fixed-order sums; GraphedTrainStep counts the nodes of every capture and refuses a step with memset nodes in the prepared-packet
# mode), plain or data-parallel, and it is equal to the bit in both modes - but a caller may capture ops of its own (the frozen decoder's MIOpen
# convolutions, torch reductions), so the package still takes the mode that is exact with ANY graph unless told otherwise.  The
# variable has to be in the environment before the HIP runtime initialises, i.e. import npvp_amd before the first CUDA call.
# NPVP_GRAPH_PACKET_CAPTURE=1 keeps the runtime's default (and the refusal above).
if _os.environ.get("NPVP_GRAPH_PACKET_CAPTURE", "0") != "1":
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def graph_packet_capture():
    """True when graph replays go through the runtime's prepared-packet path (less host time; see above for what it does to memset nodes)"""
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
