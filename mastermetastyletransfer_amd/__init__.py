"""MI355X-native stylization path of Master (Meta Style Transformer): a drop-in for the reference's
``codes.full_model.MasterStyleTransferModel`` whose forward runs hand-written gfx950 HIP kernels."""
from .graph import GraphedStylizer, GraphedTrainStep
from .imageio import Preprocess, postprocess
from .loss import VGG19_custom, custom_loss
from .model import MasterStyleTransferModel
from .meta import ReptileTrainer, allreduce_gradients, set_fast_adaptation
from .optim import FusedAdam, reptile_update
from .modules import (Decoder, ShiftedWindowAttention, ShiftedWindowAttention_for_decoder_last_MHA, StyleDecoder,
                      StyleEncoder, StyleSwinTransformerBlock, StyleTransformer, SwinEncoder)

__all__ = ["FusedAdam", "reptile_update", "ReptileTrainer", "set_fast_adaptation", "allreduce_gradients", "MasterStyleTransferModel", "GraphedStylizer", "GraphedTrainStep", "Preprocess", "postprocess", "custom_loss", "VGG19_custom", "StyleTransformer", "StyleEncoder", "StyleDecoder", "StyleSwinTransformerBlock",
           "ShiftedWindowAttention", "ShiftedWindowAttention_for_decoder_last_MHA", "Decoder", "SwinEncoder"]
