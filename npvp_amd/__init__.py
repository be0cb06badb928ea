"""npvp_amd - MI355X-native (gfx950) implementation of NPVP's Stage-2 predictor hot path.

`from npvp_amd import Predictor, VidHRFormerEncoder, VidHRformerDecoderNAR, ...` mirrors
`from models import ...` of the reference.  All arithmetic on the path runs in the hand-written HIP
kernels of libnpvp_hip.so (npvp_amd/csrc, C ABI in include/npvp_hip.h); importing the package works
anywhere, but calling an op without the built library or without a GPU raises - there is no fallback.
"""
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
