"""mispmm -- MI355X-native SpMM engine: Python plumbing (tests / bench) over the
C-ABI in include/mispmm.h.  The drop-in C++ host API lives in ../host/."""
from . import formats, synth  # noqa: F401
