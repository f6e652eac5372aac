"""pumi-pic_amd: MI355X-native PUMI-PIC particle hot loop (push -> adjacency search -> scatter ->
rebuild/migrate) behind a C-ABI (include/pumipic_hip.h).  Python here is plumbing for tests and
bench.py only; the product is pumi-pic_amd/csrc (HIP) + pumi-pic_amd/include (C++ host API)."""
from . import synth  # noqa: F401
from . import ptlio  # noqa: F401
from . import ppmio  # noqa: F401
