"""Stand-in for the slice of pytorch_lightning (1.6-1.9 API) the reference touches.  Build container only (oracle/shims/README.md).

`save_hyperparameters()` and `load_from_checkpoint()` follow Lightning's documented behaviour closely enough that a checkpoint written
through this shim has the LAYOUT of a real Medfusion checkpoint: {'state_dict', 'hyper_parameters' (the merged __init__ arguments of
the whole class hierarchy, class objects pickled by reference), 'pytorch-lightning_version', 'epoch', 'global_step'}.
"""
import inspect

import torch
import torch.nn as nn


class AttributeDict(dict):
    """pytorch_lightning.utilities.parsing.AttributeDict: the container of `self.hparams`"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


AttributeDict.__module__ = "pytorch_lightning.utilities.parsing"


class LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        """Collect the arguments of every `__init__` frame of this object's construction (innermost first, so that the outermost --
        the concrete class -- wins), like Lightning's `collect_init_args`."""
        hp = AttributeDict()
        frame = inspect.currentframe().f_back
        frames = []
        while frame is not None:
            if frame.f_code.co_name == "__init__" and frame.f_locals.get("self") is self:
                frames.append(frame)
            frame = frame.f_back
        for fr in frames:  # innermost -> outermost
            info = inspect.getargvalues(fr)
            for name in info.args:
                if name != "self":
                    hp[name] = fr.f_locals[name]
            if info.keywords:
                hp.update(fr.f_locals[info.keywords])
        object.__setattr__(self, "_hparams", hp)

    @property
    def hparams(self):
        return getattr(self, "_hparams", AttributeDict())

    def log(self, *a, **k):
        pass

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
        ck = torch.load(str(checkpoint_path), map_location=map_location or "cpu", weights_only=False)
        hp = dict(ck.get("hyper_parameters", {}))
        hp.update(kwargs)
        accepted = inspect.signature(cls.__init__).parameters
        if not any(p.kind == inspect.Parameter.VAR_KEYWORD for p in accepted.values()):
            hp = {k: v for k, v in hp.items() if k in accepted}
        model = cls(**hp)
        model.load_state_dict(ck["state_dict"], strict=strict)
        return model

    def checkpoint_dict(self, epoch=0, global_step=0):
        """what Trainer.save_checkpoint writes for this module (the parts a loader reads)"""
        return {"epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "1.9.4", "state_dict": self.state_dict(),
                "hyper_parameters": self.hparams, "optimizer_states": [], "lr_schedulers": [], "callbacks": {}, "loops": {}}


class LightningDataModule:
    pass
