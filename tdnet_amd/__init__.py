"""tdnet_amd -- the MI355X-native TDNet per-frame inference hot path (see README.md / DESIGN.md).

Importing the package sets ONE process-wide default, before the HIP runtime starts:

    GPU_MAX_HW_QUEUES = 2     (only if the variable is not set already)

Why.  HIP maps a process's streams onto a pool of hardware queues per priority class, up to GPU_MAX_HW_QUEUES (default 4) each.  A handle
runs a frame on three queues at once (the caller's stream, the second row-parity chain, the cache-only attention chain).  Once the process
owns about six hardware queues -- which happens as soon as anything else creates a few streams first: torch's stream pools (32 streams per
priority, created when torch.distributed's NCCL backend asks for one), RCCL itself, other libraries -- the queues are no longer all resident
and the frame runs at 0.66x: measured 275 -> 185 frames/s on td4-psp18 @1024x2048 for a handle created AFTER
`init_process_group("nccl")`, i.e. exactly the order of a multi-GPU run (tools/rccl_streams_probe.py, profiles/r04l_*).  With 2 (or 3) queues
per class every scenario measured runs at the full rate: RCCL first, 5 extra streams, idle handles, fp16 mode.  The variable is read when the
HIP runtime initialises, so it has to be in the environment before the first GPU call of the process; `hw_queue_note()` says whether it was.
"""
import os

_PRESET = os.environ.get("GPU_MAX_HW_QUEUES")
# Opt-out for an embedding application that manages its own environment: TDNET_NO_ENV_DEFAULTS=1 leaves os.environ untouched (the C library
# still says once on stderr when the variable is unset or > 3; README.md "Process-wide side effect").  bench.py, tests/conftest.py and
# __graft_entry__.py export the variable themselves, before any import.
if not os.environ.get("TDNET_NO_ENV_DEFAULTS"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")


def hw_queue_note():
    """One line for logs / the bench line: the value in force and whether this package could still set it."""
    late = False
    try:
        import torch
        late = _PRESET is None and torch.cuda.is_initialized() and not _SET_BEFORE_INIT
    except Exception:
        pass
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if late:
        return "GPU_MAX_HW_QUEUES=%s set AFTER the HIP runtime started (no effect: import tdnet_amd before the first GPU call, or export it)" % v
    if v is None:
        return "GPU_MAX_HW_QUEUES not set (TDNET_NO_ENV_DEFAULTS: HIP's default of 4 hardware queues per priority class applies)"
    return "GPU_MAX_HW_QUEUES=%s (%s)" % (v, "from the environment" if _PRESET is not None else "set by tdnet_amd at import")


def _cuda_initialised():
    try:
        import sys
        torch = sys.modules.get("torch")
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


_SET_BEFORE_INIT = not _cuda_initialised()
