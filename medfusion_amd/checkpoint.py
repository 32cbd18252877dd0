"""Checkpoint loading without pytorch_lightning (reference: model_base.py:63-85,
diffusion_pipeline.py:57-60, Lightning `load_from_checkpoint`).

A Lightning .ckpt is a torch-pickle dict {'state_dict', 'hyper_parameters', ...} whose hyper-parameters hold
CLASS objects of the reference package (`medical_diffusion.models...UNet`).  A restricted unpickler maps those
references onto this package's classes, so trained Medfusion weights load into the HIP path with
`DiffusionPipeline.load_from_checkpoint(path)` and no reference / Lightning import.
"""
from __future__ import annotations

import io
import pickle
from pathlib import Path

import torch

_CLASS_MAP = {
    "UNet": ("medfusion_amd.unet", "UNet"),
    "TimeEmbbeding": ("medfusion_amd.unet", "TimeEmbbeding"),
    "SinusoidalPosEmb": ("medfusion_amd.unet", "SinusoidalPosEmb"),
    "LabelEmbedder": ("medfusion_amd.unet", "LabelEmbedder"),
    "GaussianNoiseScheduler": ("medfusion_amd.scheduler", "GaussianNoiseScheduler"),
    "VAE": ("medfusion_amd.vae", "VAE"),
    "DiffusionPipeline": ("medfusion_amd.pipeline", "DiffusionPipeline"),
}


class _Placeholder:
    """Stands in for training-only classes referenced by hyper-parameters (losses, LPIPS, ...)."""

    def __init__(self, *a, **k):
        pass


class _RemapUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("medical_diffusion"):
            if name in _CLASS_MAP:
                import importlib

                mod, attr = _CLASS_MAP[name]
                return getattr(importlib.import_module(mod), attr)
            return _Placeholder
        if module.startswith(("pytorch_lightning", "lightning", "lpips", "pytorch_msssim", "torchmetrics")):
            return _Placeholder
        return super().find_class(module, name)


class _RemapPickle:
    """pickle_module shim for torch.load"""
    __name__ = "medfusion_amd_remap_pickle"
    Unpickler = _RemapUnpickler

    @staticmethod
    def load(f, **kw):
        return _RemapUnpickler(f, **kw).load()


def read_checkpoint(path, map_location="cpu") -> dict:
    return torch.load(str(path), map_location=map_location, pickle_module=_RemapPickle, weights_only=False)


def _strip(sd: dict, prefix: str) -> dict:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def load_module_from_checkpoint(cls, path, map_location="cpu", **overrides):
    """`cls.load_from_checkpoint(path)` for a leaf module (the VAE referenced by diffusion_pipeline.py:57-58)."""
    ck = read_checkpoint(path, map_location)
    hp = dict(ck.get("hyper_parameters", {}))
    hp.update(overrides)
    hp = {k: v for k, v in hp.items() if not isinstance(v, _Placeholder) and v is not _Placeholder}
    model = cls(**hp)
    model.load_state_dict(ck["state_dict"], strict=False)
    return model.eval()


def load_pipeline_from_checkpoint(cls, path, map_location="cpu", **overrides):
    ck = read_checkpoint(path, map_location)
    hp = dict(ck.get("hyper_parameters", {}))
    hp.update(overrides)
    hp = {k: v for k, v in hp.items() if v is not _Placeholder and not isinstance(v, _Placeholder)}
    sd = ck["state_dict"]
    has_vae_weights = any(k.startswith("latent_embedder.") for k in sd)
    ckpt_path = hp.get("latent_embedder_checkpoint", "")
    if hp.get("latent_embedder") is not None and has_vae_weights and not (ckpt_path and Path(ckpt_path).exists()):
        # the nested VAE checkpoint path baked into the hparams usually does not exist on the sampling box;
        # its weights are in this state_dict anyway -- rebuild the VAE from the tensor shapes' hyper-parameters if given
        vae_kwargs = overrides.get("latent_embedder_kwargs")
        if vae_kwargs is None:
            raise RuntimeError("latent_embedder_checkpoint is not readable; pass latent_embedder_kwargs=... to rebuild the VAE")
        hp["latent_embedder"] = hp["latent_embedder"](**vae_kwargs)
    hp.pop("latent_embedder_kwargs", None)
    pipe = cls(**hp)
    missing, unexpected = pipe.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not k.endswith("num_batches_tracked")]
    if bad:
        raise RuntimeError(f"checkpoint is missing {len(bad)} tensors, e.g. {bad[:4]}")
    return pipe.eval()
