"""Names the reference's training scripts import from `muse` but whose models are outside this build's hot path (SURVEY.md section 2:
"OUT OF SCOPE - not named in north_star"): the MoVQ and Paella tokenizers (muse/modeling_movq.py, muse/modeling_paella_vq.py).  The
scripts import them unconditionally (training/train_muse.py:51-60, training/train_maskgit_imagenet.py:38) and pick one by the config's
`model.vq_model.type`; the two built tokenizers are `maskgit_vqgan` (muse.MaskGitVQGAN) and `vqgan` (muse.VQGANModel).  Importing the
names works; building one of these models says what is missing instead of computing something else."""
from __future__ import annotations

from ._hip import MuseHipError


class _NotBuilt:
    _what = "this model"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} ({self._what}) is not part of the MI355X hot-path build: the tokenizers built on the "
                                  "HIP kernels are muse.MaskGitVQGAN (`maskgit_vqgan`) and muse.VQGANModel (`vqgan`)")

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        cls()

    @classmethod
    def from_config(cls, *args, **kwargs):
        cls()


class MOVQ(_NotBuilt):
    _what = "the MoVQ tokenizer, reference muse/modeling_movq.py"


class PaellaVQModel(_NotBuilt):
    _what = "the Paella VQ tokenizer, reference muse/modeling_paella_vq.py"


__all__ = ["MOVQ", "PaellaVQModel", "MuseHipError"]
