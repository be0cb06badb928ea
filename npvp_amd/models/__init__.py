"""Mirror of the reference's `models` package for the Stage-2 predictor path (ref/models/__init__.py)."""
from .criterion import L1Loss, Div_KL
from .VidHRFormer import (VidHRformerDecoderNAR, VidHRFormerEncoder, VidHRFormerBlockEnc, VidHRFormerBlockDecNAR,
                          SpatialLocalMultiheadAttention, MlpDWBN, MultiheadAttention, DropPath)
from .submodules import CoorGenerator, NRMLP, PosFeatFuser, EventEncoder
from .Predictor import Predictor
from .ResNetAutoEncoder import ResnetEncoder, ResnetDecoder, ResnetBlock, Factorized3DConvAttn, NonLocalAttenion2D, build_frozen_autoencoder, to_device_layout
