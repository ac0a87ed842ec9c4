"""Python host side of the MI355X direct N-body engine.

This package is plumbing around the C ABI in ``include/nbody3d_hip.h`` (the
product is ``csrc/libnbody3d_hip.so`` and the Node wrapper in ``js/``): a ctypes
binding used by the tests and ``bench.py``, initial-condition generators, and
the one-process-per-GPU shard driver that runs the position all-gather through
``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests).

There is NO CPU compute path in here: every force/integrate call goes through
the HIP library and fails loudly when it (or a GPU) is missing.
"""
from .capi import (  # noqa: F401
    MultiSimulation,
    NBodyError,
    Simulation,
    abi_version,
    device_count,
    library_path,
    load_library,
)
from . import ic  # noqa: F401
from .shard import ShardPlan  # noqa: F401

EPS2 = 1e-4  # reference softening, /root/reference nbody3d.js:234
