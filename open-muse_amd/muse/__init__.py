"""muse — MI355X-native drop-in for the hot path of huggingface/open-muse.

Same import surface as the reference package for the classes on the path (reference muse/__init__.py:18-25):
MaskGitTransformer, MaskGitVQGAN, PipelineMuse, get_mask_chedule; everything computes through libmuse_hip.so
(hand-written HIP kernels for gfx950).  Components the hot path does not touch (taming/MoVQ/Paella VQ models, EMA,
U-ViT) are not part of this build.
"""
__version__ = "0.0.1"

from .modeling_maskgit_vqgan import MaskGitVQGAN
from .modeling_transformer import MaskGitTransformer
from .pipeline_muse import PipelineMuse
from .sampling import get_mask_chedule
from .training import FusedAdamW, GradReducer, TrainStep, prepare_inputs_and_labels

__all__ = ["MaskGitVQGAN", "MaskGitTransformer", "PipelineMuse", "get_mask_chedule", "FusedAdamW", "GradReducer",
           "TrainStep", "prepare_inputs_and_labels"]
