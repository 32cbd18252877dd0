"""Checkpoint loading without pytorch_lightning (reference: model_base.py:63-85,
diffusion_pipeline.py:57-60, Lightning `load_from_checkpoint`).

A Lightning .ckpt is a torch-pickle dict {'state_dict', 'hyper_parameters', ...} whose hyper-parameters hold
CLASS objects of the reference package (`medical_diffusion.models...UNet`).  An allow-listing unpickler maps those
references onto this package's classes, so trained Medfusion weights load into the HIP path with
`DiffusionPipeline.load_from_checkpoint(path)` and no reference / Lightning import.

The nested VAE (diffusion_pipeline.py:57-58 calls `latent_embedder.load_from_checkpoint(latent_embedder_checkpoint)` with a path
baked in at training time) is resolved in this order: the `latent_embedder_checkpoint=` override; the baked path (as given, then
relative to the pipeline checkpoint's directory); `latent_embedder_kwargs=`; and finally the VAE's hyper-parameters INFERRED from the
shapes of the `latent_embedder.*` tensors the pipeline checkpoint itself carries (a LightningModule saves its sub-modules' weights).
"""
from __future__ import annotations

import pickle
import re
from pathlib import Path

import torch

_CLASS_MAP = {
    "UNet": ("medfusion_amd.unet", "UNet"),
    "TimeEmbbeding": ("medfusion_amd.unet", "TimeEmbbeding"),
    "SinusoidalPosEmb": ("medfusion_amd.unet", "SinusoidalPosEmb"),
    "LearnedSinusoidalPosEmb": ("medfusion_amd.unet", "LearnedSinusoidalPosEmb"),
    "LabelEmbedder": ("medfusion_amd.unet", "LabelEmbedder"),
    "GaussianNoiseScheduler": ("medfusion_amd.scheduler", "GaussianNoiseScheduler"),
    "VAE": ("medfusion_amd.vae", "VAE"),
    "DiffusionPipeline": ("medfusion_amd.pipeline", "DiffusionPipeline"),
}
# training-only objects a checkpoint may reference: replaced by an inert placeholder
_PLACEHOLDER_PREFIXES = ("medical_diffusion", "pytorch_lightning", "lightning", "lightning_fabric", "lpips", "pytorch_msssim", "torchmetrics",
                         "torch.optim", "torch.nn.modules.loss", "torchvision")
# everything else must come from here (a checkpoint is data: no other global may be resolved, so unpickling cannot run foreign code)
_ALLOWED = {
    "collections": {"OrderedDict", "defaultdict"},
    # (no `getattr`, no `object`: getattr(object, "__subclasses__")() reaches every class of the interpreter)
    "builtins": {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "complex", "slice", "range", "bytearray"},
    "pathlib": {"Path", "PosixPath", "PurePosixPath", "WindowsPath", "PureWindowsPath"},
    "numpy": {"ndarray", "dtype"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"},
    "numpy._core.multiarray": {"_reconstruct", "scalar"},
    "torch": {"Size", "device", "dtype", "FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
              "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "Tensor", "float32", "float64", "float16", "bfloat16", "int64",
              "int32", "int16", "int8", "uint8", "bool"},
    "torch._utils": {"_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_tensor", "_rebuild_parameter_with_state"},
    # (no `_load_from_bytes`: it is torch.load(BytesIO(b), weights_only=False), an unrestricted nested unpickle; only the legacy
    # non-zip format needs it)
    "torch.storage": {"UntypedStorage", "TypedStorage"},
    "torch.nn.parameter": {"Parameter"},
    "torch.serialization": {"_get_layout"},
}


class _Placeholder:
    """Stands in for training-only classes / objects referenced by hyper-parameters (losses, LPIPS, AttributeDict, ...).
    Tolerates every way pickle may build or fill an instance."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def __call__(self, *a, **k):
        return self

    def append(self, v):
        pass

    def extend(self, v):
        pass

    def update(self, *a, **k):
        pass


class _RemapUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("medical_diffusion") and name in _CLASS_MAP:
            import importlib

            mod, attr = _CLASS_MAP[name]
            return getattr(importlib.import_module(mod), attr)
        if name == "AttributeDict":   # Lightning's hparams container (a dict subclass)
            return dict
        if module.startswith(_PLACEHOLDER_PREFIXES):
            return _Placeholder
        if name in _ALLOWED.get(module, ()):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint references {module}.{name}, which is not on the allow-list of medfusion_amd.checkpoint")


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


def _clean_hparams(hp: dict) -> dict:
    def bad(v):
        return v is _Placeholder or isinstance(v, _Placeholder)

    out = {}
    for k, v in hp.items():
        if bad(v):
            continue
        if isinstance(v, dict):
            v = {kk: vv for kk, vv in v.items() if not bad(vv)}
        out[k] = v
    return out


_TRAINING_ONLY = re.compile(r"^(perceiver|loss|loss_fct|discriminator|vqvae|ssim_fct)\b|\.num_batches_tracked$")


def _check_missing(missing, what):
    bad = [k for k in missing if not _TRAINING_ONLY.search(k)]
    if bad:
        raise RuntimeError(f"{what}: the checkpoint is missing {len(bad)} tensors of the model, e.g. {bad[:4]} -- different architecture or key names")


def infer_vae_kwargs(sd: dict) -> dict:
    """Hyper-parameters of a `VAE` (latent_embedders.py:620-749) from the SHAPES of its state-dict tensors (keys without the
    `latent_embedder.` prefix).  GroupNorm group counts leave no trace in the shapes: the reference default (8 groups) is assumed."""
    def shape(key):
        if key not in sd:
            raise RuntimeError(f"cannot infer the VAE architecture: tensor '{key}' is not in the checkpoint")
        return tuple(sd[key].shape)

    if any(".attention." in k for k in sd):
        raise RuntimeError("the VAE in this checkpoint uses attention blocks: pass latent_embedder_kwargs=... explicitly")
    res = any(".basic_block." in k for k in sd)
    conv0 = "inc.block_seq.0.basic_block.conv.weight" if res else "inc.block_seq.0.conv.weight"
    w0 = shape(conv0)
    hid, ks, strides = [w0[0]], [w0[2]], [1]
    n_enc = 1 + max([int(m.group(1)) for k in sd for m in [re.match(r"encoders\.(\d+)\.", k)] if m], default=-1)
    for i in range(n_enc):
        blk = f"encoders.{i}.conv_block.block_seq.0." + ("basic_block.conv.weight" if res else "conv.weight")
        wb = shape(blk)
        hid.append(wb[0])
        ks.append(wb[2])
        strides.append(2 if f"encoders.{i}.down_op.down_op.weight" in sd else 1)
    emb = shape("inc_dec.block_seq.0." + ("basic_block.conv.weight" if res else "conv.weight"))[1]
    deep = 1 + max([int(m.group(1)) for k in sd for m in [re.match(r"outc_ver\.(\d+)\.", k)] if m], default=-1)
    return dict(in_channels=w0[1], out_channels=shape("outc.conv.weight")[0], spatial_dims=2, emb_channels=emb, hid_chs=hid, kernel_sizes=ks,
                strides=strides, use_res_block=res, deep_supervision=deep, use_attention="none")


def load_module_from_checkpoint(cls, path, map_location="cpu", **overrides):
    """`cls.load_from_checkpoint(path)` for a leaf module (the VAE referenced by diffusion_pipeline.py:57-58)."""
    ck = read_checkpoint(path, map_location)
    hp = _clean_hparams(dict(ck.get("hyper_parameters", {})))
    hp.update(overrides)
    model = cls(**hp)
    missing, _unexpected = model.load_state_dict(ck["state_dict"], strict=False)
    _check_missing(missing, f"{cls.__name__}.load_from_checkpoint({path})")
    return model.eval()


def load_pipeline_from_checkpoint(cls, path, map_location="cpu", **overrides):
    ck = read_checkpoint(path, map_location)
    hp = _clean_hparams(dict(ck.get("hyper_parameters", {})))
    vae_kwargs = overrides.pop("latent_embedder_kwargs", None)
    hp.update(overrides)
    sd = ck["state_dict"]
    vae_cls = hp.get("latent_embedder")
    if isinstance(vae_cls, type):
        baked = hp.get("latent_embedder_checkpoint", "") or ""
        here = Path(path).resolve().parent
        # as given (relative to the cwd, like the reference), then relative to the checkpoint's directory and its parents (a `runs/` tree
        # that moved as a whole), then a file of that name next to the pipeline checkpoint
        cands = ([Path(baked)] + [q / baked for q in [here, *list(here.parents)[:4]]] + [here / Path(baked).name]) if baked else []
        found = next((c for c in cands if c.is_file()), None)
        if found is not None:
            hp["latent_embedder_checkpoint"] = str(found)
        else:
            vsd = _strip(sd, "latent_embedder.")
            if not vsd:
                raise RuntimeError(f"latent_embedder_checkpoint '{baked}' is not readable and the pipeline checkpoint carries no latent_embedder.* "
                                   f"tensors: pass latent_embedder_checkpoint=<path of the VAE checkpoint>")
            hp["latent_embedder"] = vae_cls(**(vae_kwargs if vae_kwargs is not None else infer_vae_kwargs(vsd)))  # weights: from `sd` below
    pipe = cls(**hp)
    missing, _unexpected = pipe.load_state_dict(sd, strict=False)
    _check_missing(missing, f"DiffusionPipeline.load_from_checkpoint({path})")
    return pipe.eval()
