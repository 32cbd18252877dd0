import contextlib


@contextlib.contextmanager
def pl_legacy_patch():
    yield
