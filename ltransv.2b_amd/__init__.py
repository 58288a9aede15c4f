"""MI355X-native LTRANS particle-advection engine (host side).

The directory name follows the reference repository (``LTRANSv.2b``) and is not
a valid Python identifier; import it through the root-level alias module::

    import ltrans_amd            # /root/repo/ltrans_amd.py registers this package

Only what the hot path needs lives here:

``csrc/``      hand-written HIP kernels for gfx950 + the C-ABI (include/ltrans_hip.h)
``fortran/``   ISO_C_BINDING interface module + Fortran host harness
``abi.py``     ctypes mirror of include/ltrans_hip.h
``engine.py``  Engine: thin Python mirror of the reference's interface for the path
``driver.py``  external/internal time-step loops (LTRANS.f90:141-161,548-612)
``namelist.py`` LTRANS.data reader ($group ... $end namelists)
``roms_synth.py`` synthetic ROMS grid / history / particle generator + the
               init-time geometry the reference builds on the host
"""
from . import abi  # noqa: F401
from .abi import LtConfig, LtError  # noqa: F401
from .engine import Engine, create_bounds, load_hip_library  # noqa: F401
from .namelist import read_namelists, config_from_namelists  # noqa: F401
