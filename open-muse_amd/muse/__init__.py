"""muse — MI355X-native drop-in for the hot path of huggingface/open-muse.

Same import surface as the reference package for the classes on the path (reference muse/__init__.py:18-25):
MaskGitTransformer, MaskGiTUViT, MaskGitVQGAN, VQGANModel (the taming tokenizer of the text-to-image configs),
PipelineMuse, PipelineMuseInpainting, EMAModel (the weight average train_muse.py advances behind every optimizer step), get_mask_chedule; everything computes
through libmuse_hip.so (hand-written HIP kernels for gfx950).
The MoVQ / Paella VQ models are not part of this build: their names import (the training scripts import them unconditionally) and
refuse to construct.  `muse.lr_schedulers` / `muse.training_utils` carry the host-side helpers those scripts import.
"""
__version__ = "0.0.1"

from .ema import EMAModel
from .modeling_maskgit_vqgan import MaskGitVQGAN
from .modeling_taming_vqgan import VQGANModel
from .modeling_transformer import MaskGitTransformer
from .modeling_transformer_v2 import MaskGiTUViT, MaskGiTUViT_v2
from . import pre_encode
from .pipeline_muse import PipelineMuse, PipelineMuseInpainting
from .sampling import get_mask_chedule
from .unbuilt import MOVQ, PaellaVQModel
from . import lr_schedulers, training_utils
from .training import (FusedAdamW, GradReducer, TrainStep, cond_dropout, grouped_parameters, mask_or_random_replace_tokens,
                       prepare_inputs_and_labels)

__all__ = ["MOVQ", "PaellaVQModel", "EMAModel", "MaskGitVQGAN", "VQGANModel", "MaskGitTransformer", "MaskGiTUViT", "MaskGiTUViT_v2", "PipelineMuse", "PipelineMuseInpainting", "get_mask_chedule", "FusedAdamW", "GradReducer",
           "TrainStep", "prepare_inputs_and_labels", "mask_or_random_replace_tokens", "cond_dropout", "grouped_parameters"]
