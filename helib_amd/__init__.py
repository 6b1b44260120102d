"""helib_amd -- MI355X-native DoubleCRT polynomial-arithmetic engine behind HElib's
Cmodulus / DoubleCRT / key-switching interface (include/helib_amd.h)."""
from . import capi  # noqa: F401

__all__ = ["capi"]
