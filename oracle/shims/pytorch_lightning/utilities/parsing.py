from pytorch_lightning import AttributeDict  # noqa: F401  (pickled by reference in checkpoints: pytorch_lightning.utilities.parsing.AttributeDict)
