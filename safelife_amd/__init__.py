"""
safelife_amd -- the SafeLife per-step hot path on AMD Instinct MI355X (gfx950).

This package accelerates exactly one path of the reference (PartnershipOnAI/safelife): the
cellular-automaton step of ``safelife/speedups_src`` and the integer glue of
``SafeLifeEnv.step()/reset()`` around it.  Layout:

* ``speedups``    functions with the reference's ``safelife.speedups`` signatures, on the GPU
* ``vector_env``  ``SafeLifeVectorEnv``: B device-resident envs, one fused launch per step
* ``game`` / ``env``  ``SafeLifeGame`` / ``SafeLifeEnv`` look-alikes (one env at a time) built on
  ``speedups`` so reference-style wrappers and training loops run unchanged
* ``levels``      ``.npz`` level loader and the device level pool
* ``sharding``    one process per GPU: env partition + reward/done gather over RCCL
* ``csrc/``       hand-written HIP kernels and the C-ABI (include/safelife_hip.h)

Importing the package needs neither torch nor a GPU; calling any compute entry point without
``libsafelife_hip.so`` or without a HIP device raises (there is no CPU fallback).
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: the fused step kernel takes a
# ~600-byte argument struct and starts ~5 us earlier per launch this way (measured on MI355X).  Must
# be set before the HIP runtime initialises, i.e. before torch is imported.
import sys as _sys

if "torch" in _sys.modules and _os.environ.get("HIP_FORCE_DEV_KERNARG") != "1":
    # the HIP runtime is already up (torch was imported first) and reads the variable only when it loads
    import warnings as _warnings
    _warnings.warn("safelife_amd: torch was imported before safelife_amd and HIP_FORCE_DEV_KERNARG is not set; "
                   "the fused step kernel will start ~3 us later per launch (11.7 instead of 8.7 us per C3 step). "
                   "Import safelife_amd first or export HIP_FORCE_DEV_KERNARG=1.", RuntimeWarning, stacklevel=2)
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .cell_types import CellTypes, DEFAULT_POINTS_TABLE  # noqa: F401,E402

__version__ = "0.1.0"

__all__ = ["CellTypes", "DEFAULT_POINTS_TABLE", "speedups", "levels", "vector_env", "game", "env",
           "sharding"]
