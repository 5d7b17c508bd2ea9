"""Model runtime: config registration and `from_pretrained` / `save_pretrained`.

Keeps the on-disk formats and the class surface of the reference's muse/modeling_utils.py (ConfigMixin :804-1125,
ModelMixin :228-766, register_to_config :1128-1170): `config.json` (sorted JSON of the registered init kwargs plus
`_class_name` / `_version`) and `pytorch_model.bin` (torch.save of the state_dict), so checkpoints move freely between
the reference and this package.  Pure host-side Python: no kernel work happens here.
"""
from __future__ import annotations

import functools
import inspect
import json
import os
from collections import OrderedDict
from pathlib import PosixPath
from typing import Any, Callable, Dict, Optional, Union

import numpy as np
import torch

__version__ = "0.0.1"  # written to config.json as "_version", same value as the reference (muse/__init__.py:16)

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "pytorch_model.safetensors"


class FrozenDict(OrderedDict):
    """Config container with attribute access.  Like the reference's (modeling_utils.py:772-801) it accepts new
    attributes after construction (the VQGAN constructor stores derived values on it) but refuses dict mutation."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in self.items():
            object.__setattr__(self, k, v)
        object.__setattr__(self, "_sealed", True)

    def _blocked(self, *a, **k):
        raise Exception(f"You cannot mutate a {self.__class__.__name__} instance in place.")

    __delitem__ = setdefault = pop = update = _blocked

    def __setitem__(self, k, v):
        if getattr(self, "_sealed", False):
            self._blocked()
        super().__setitem__(k, v)

    # copy.deepcopy(model) (the reference's training_utils.EMA keeps a deep copy of the model, :61-80) and pickling go through
    # __reduce_ex__: rebuild from the items, then restore the attributes that are not items (the constructors' derived values)
    def __reduce__(self):
        extra = {k: v for k, v in vars(self).items() if k not in self and k != "_sealed"}
        return (self.__class__, (dict(self),), extra)

    def __setstate__(self, state):
        for k, v in (state or {}).items():
            object.__setattr__(self, k, v)


class ConfigMixin:
    config_name = CONFIG_NAME
    ignore_for_config = []

    def register_to_config(self, **kwargs):
        kwargs.pop("kwargs", None)
        for k, v in kwargs.items():
            try:
                setattr(self, k, v)  # reference :834-839 — config keys are also attributes of the module
            except AttributeError:
                pass
        old = getattr(self, "_internal_dict", None)
        merged = {**dict(old or {}), **kwargs}
        self._internal_dict = FrozenDict(merged)
        # derived values a constructor hung on the config as plain attributes (the VQGANs' num_resolutions / latent_size, reference
        # modeling_maskgit_vqgan.py:370-372) survive a later registration (from_pretrained adds `_name_or_path`): the reference's
        # submodules keep reading them from the old object, ours read the model's current one
        for k, v in (vars(old).items() if old is not None else ()):
            if k not in merged and not k.startswith("_"):
                object.__setattr__(self._internal_dict, k, v)

    @property
    def config(self) -> FrozenDict:
        return self._internal_dict

    # ---- serialisation -----------------------------------------------------------------------------------------
    def to_json_string(self) -> str:
        d = dict(getattr(self, "_internal_dict", {}))
        d["_class_name"] = self.__class__.__name__
        d["_version"] = __version__

        def saveable(v):
            if isinstance(v, np.ndarray):
                return v.tolist()
            if isinstance(v, PosixPath):
                return str(v)
            return v

        return json.dumps({k: saveable(v) for k, v in d.items()}, indent=2, sort_keys=True) + "\n"

    def to_json_file(self, path):
        with open(path, "w", encoding="utf-8") as f:
            f.write(self.to_json_string())

    def save_config(self, save_directory, push_to_hub: bool = False, **kwargs):
        if os.path.isfile(save_directory):
            raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        self.to_json_file(os.path.join(save_directory, self.config_name))

    def __repr__(self):
        return f"{self.__class__.__name__} {self.to_json_string()}"

    # ---- loading -----------------------------------------------------------------------------------------------
    @classmethod
    def _get_init_keys(cls):
        return set(inspect.signature(cls.__init__).parameters.keys())

    @classmethod
    def load_config(cls, pretrained_model_name_or_path, return_unused_kwargs: bool = False, **kwargs):
        subfolder = kwargs.pop("subfolder", None)
        path = str(pretrained_model_name_or_path)
        if os.path.isfile(path):
            config_file = path
        elif os.path.isdir(path):
            cand = os.path.join(path, subfolder, cls.config_name) if subfolder else os.path.join(path, cls.config_name)
            if not os.path.isfile(cand):
                raise EnvironmentError(f"Error no file named {cls.config_name} found in directory {path}.")
            config_file = cand
        else:
            config_file = _hub_file(path, cls.config_name, subfolder, kwargs)
        with open(config_file, "r", encoding="utf-8") as f:
            cfg = json.load(f)
        return (cfg, kwargs) if return_unused_kwargs else cfg

    @classmethod
    def from_config(cls, config: Union[FrozenDict, Dict[str, Any]] = None, return_unused_kwargs: bool = False, **kwargs):
        if config is None:
            raise ValueError("Please make sure to provide a config as the first positional argument.")
        config = dict(config)
        init_keys = cls._get_init_keys() - {"self", "kwargs"}
        kwargs_only = not init_keys   # `def __init__(self, **kwargs)` (MaskGiTUViT_v2): the reference passes the whole config (:908)
        init_dict, unused = {}, {}
        for k, v in {**config, **kwargs}.items():
            if k.startswith("_"):
                continue
            if kwargs_only or k in init_keys:
                init_dict[k] = v
            else:
                unused[k] = v
        model = cls(**init_dict)
        hidden = {k: v for k, v in config.items() if k.startswith("_") and k not in ("_class_name", "_version")}
        if hidden:
            model.register_to_config(**hidden)
        return (model, unused) if return_unused_kwargs else model


def _hub_file(repo_id, filename, subfolder, kwargs):
    try:
        from huggingface_hub import hf_hub_download

        return hf_hub_download(repo_id, filename=filename, subfolder=subfolder, cache_dir=kwargs.get("cache_dir"),
                               revision=kwargs.get("revision"), local_files_only=kwargs.get("local_files_only", False))
    except Exception as e:  # no network in the build / GPU containers
        raise EnvironmentError(
            f"Can't load {filename} for '{repo_id}': not a local path and the Hub is not reachable ({e}).") from e


def register_to_config(init):
    """Decorator for __init__: records every init argument (positional or keyword, with defaults) in `self.config`;
    arguments starting with '_' are recorded but not passed to the constructor (reference :1128-1170)."""

    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        if not isinstance(self, ConfigMixin):
            raise RuntimeError(f"`@register_to_config` applied to {self.__class__.__name__}, which is not a ConfigMixin")
        public = {k: v for k, v in kwargs.items() if not k.startswith("_")}
        private = {k: v for k, v in kwargs.items() if k.startswith("_")}
        ignore = getattr(self, "ignore_for_config", [])
        params = [(n, p.default) for i, (n, p) in enumerate(inspect.signature(init).parameters.items())
                  if i > 0 and n not in ignore]
        cfg = {}
        for a, (n, _) in zip(args, params):
            cfg[n] = a
        for n, default in params:
            if n not in cfg:
                cfg[n] = public.get(n, default)
        self.register_to_config(**{**private, **cfg})
        init(self, *args, **public)

    return inner


class ModelMixin(torch.nn.Module):
    config_name = CONFIG_NAME
    _supports_gradient_checkpointing = False

    def __init__(self):
        super().__init__()

    # ---- knobs the training scripts touch ----------------------------------------------------------------------
    @property
    def is_gradient_checkpointing(self) -> bool:
        return bool(getattr(self, "gradient_checkpointing", False))

    def enable_gradient_checkpointing(self):
        if not self._supports_gradient_checkpointing:
            raise ValueError(f"{self.__class__.__name__} does not support gradient checkpointing.")
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def enable_xformers_memory_efficient_attention(self, attention_op: Optional[Callable] = None):
        """No-op: attention here always runs on the library's own HIP kernels (training scripts call this at
        train_maskgit_imagenet.py:229-230 / train_muse.py:392-393)."""
        return None

    def disable_xformers_memory_efficient_attention(self):
        return None

    def set_use_memory_efficient_attention_xformers(self, valid: bool, attention_op: Optional[Callable] = None):
        return None

    # ---- save / load -------------------------------------------------------------------------------------------
    def save_pretrained(self, save_directory, is_main_process: bool = True, save_function: Callable = None,
                        state_dict: Optional[Dict[str, torch.Tensor]] = None):
        if os.path.isfile(save_directory):
            raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        if is_main_process:
            self.save_config(save_directory)
        if state_dict is None:
            state_dict = self.state_dict()
        state_dict = {k: v.detach().to("cpu").contiguous().clone() for k, v in state_dict.items()}
        for f in os.listdir(save_directory):  # drop stale weight files, like the reference does
            if f.startswith(WEIGHTS_NAME[:-4]) and os.path.isfile(os.path.join(save_directory, f)) and is_main_process:
                os.remove(os.path.join(save_directory, f))
        (save_function or torch.save)(state_dict, os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        cache_dir = kwargs.pop("cache_dir", None)
        torch_dtype = kwargs.pop("torch_dtype", None)
        subfolder = kwargs.pop("subfolder", None)
        output_loading_info = kwargs.pop("output_loading_info", False)
        ignore_mismatched_sizes = kwargs.pop("ignore_mismatched_sizes", False)
        for k in ("low_cpu_mem_usage", "device_map", "force_download", "resume_download", "proxies", "local_files_only",
                  "use_auth_token", "revision", "from_flax", "use_safetensors"):
            kwargs.pop(k, None)
        if torch_dtype is not None and not isinstance(torch_dtype, torch.dtype):
            raise ValueError(f"{torch_dtype} needs to be of type `torch.dtype`, e.g. `torch.float16`.")

        path = str(pretrained_model_name_or_path)
        config, unused = cls.load_config(path, return_unused_kwargs=True, subfolder=subfolder, cache_dir=cache_dir, **kwargs)
        model_file = _model_file(path, subfolder, cache_dir)
        model = cls.from_config(config, **unused)

        state_dict = torch.load(model_file, map_location="cpu", weights_only=True)
        own = model.state_dict()
        missing = [k for k in own if k not in state_dict]
        if missing:
            raise ValueError(f"Cannot load {cls} from {path} because the following keys are missing: \n {', '.join(missing)}.")
        unexpected = [k for k in state_dict if k not in own]
        mismatched = []
        for k in list(state_dict):
            if k in own and tuple(own[k].shape) != tuple(state_dict[k].shape):
                if not ignore_mismatched_sizes:
                    raise ValueError(f"Cannot load {path}: {k} has shape {tuple(state_dict[k].shape)} in the checkpoint, "
                                     f"{tuple(own[k].shape)} in the model.")
                mismatched.append(k)
                del state_dict[k]
        model.load_state_dict({k: v for k, v in state_dict.items() if k in own}, strict=False)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        model.register_to_config(_name_or_path=path)
        model.eval()
        if output_loading_info:
            return model, dict(missing_keys=missing, unexpected_keys=unexpected, mismatched_keys=mismatched, error_msgs=[])
        return model

    # ---- precision casts -----------------------------------------------------------------------------------------------------------
    # The reference's scripts cast whole models (`model.half()`, scripts/benchmark_models.py:33-34; `.to(device, dtype=dtype)`,
    # pipeline_muse.py:55-64; `from_pretrained(torch_dtype=...)`).  Here the master parameters stay float32 (what FusedAdamW steps, what
    # save_pretrained writes) and a floating-point cast selects the transformers' COMPUTE mode instead: half / bfloat16 -> the bf16 MFMA
    # path (the kernels have no f16 variant; bf16 has the wider exponent and the same MFMA rate), float32 / float64 -> exact f32.
    # The tokenizers ("keep vae in fp32", pipeline_muse.py:62) keep their own mode.
    _cast_selects_compute_mode = False          # the transformer classes set this

    def _compute_mode_for(self, dtype):
        if dtype is not None and dtype.is_floating_point and self._cast_selects_compute_mode:
            self.set_compute_dtype(torch.bfloat16 if dtype in (torch.float16, torch.bfloat16) else torch.float32)
            return True
        return False

    def half(self):
        return self if self._compute_mode_for(torch.float16) else super().half()

    def bfloat16(self):
        return self if self._compute_mode_for(torch.bfloat16) else super().bfloat16()

    def float(self):
        return self if self._compute_mode_for(torch.float32) else super().float()

    def double(self):
        return self if self._compute_mode_for(torch.float64) else super().double()

    def to(self, *args, **kwargs):
        if not self._cast_selects_compute_mode:
            return super().to(*args, **kwargs)
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        self._compute_mode_for(dtype)
        return super().to(device=device, non_blocking=non_blocking) if device is not None else self

    # ---- introspection -----------------------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")

    @property
    def dtype(self) -> torch.dtype:
        for p in self.parameters():
            return p.dtype
        return torch.float32

    def num_parameters(self, only_trainable: bool = False, exclude_embeddings: bool = False) -> int:
        skip = set()
        if exclude_embeddings:
            skip = {id(p) for n, p in self.named_parameters() if "embeddings" in n or n.endswith("embedding.weight")}
        return sum(p.numel() for p in self.parameters()
                   if (p.requires_grad or not only_trainable) and id(p) not in skip)


def _model_file(path, subfolder, cache_dir):
    if os.path.isfile(path):
        return path
    if os.path.isdir(path):
        cands = [os.path.join(path, subfolder, WEIGHTS_NAME)] if subfolder else []
        cands.append(os.path.join(path, WEIGHTS_NAME))
        for c in cands:
            if os.path.isfile(c):
                return c
        raise EnvironmentError(f"Error no file named {WEIGHTS_NAME} found in directory {path}.")
    return _hub_file(path, WEIGHTS_NAME, subfolder, {"cache_dir": cache_dir})
