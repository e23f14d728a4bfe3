"""Stub of `jaxtyping` (absent offline) so the REFERENCE modules import unmodified
(SURVEY.md Appendix C).  Annotation objects only; no runtime checking."""
from contextlib import contextmanager


class _Meta(type):
    def __getitem__(cls, item):
        return cls


class _Ann(metaclass=_Meta):
    pass


Float = Int = Int64 = Int32 = Bool = Shaped = UInt8 = _Ann


@contextmanager
def install_import_hook(*args, **kwargs):
    yield
