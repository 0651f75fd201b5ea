"""Helper streams that REALLY run beside the streams they are meant to overlap with.

The schedule of a pass (sa_stack.py: the FPS producer beside its chunked consumers; pointnet2_modules.py: the scales of
a layer beside each other) needs kernels of different HIP streams to execute concurrently.  HIP maps its streams onto a
few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and streams that share a queue are serialised; which ones share
depends on every other stream the process has created -- with RCCL initialised the producer and its consumers landed on one
queue and a pass took FPS + everything else (3.2 ms) instead of their maximum (2.3 ms) on MI355X.  No reference counterpart:
the reference launches everything on the legacy default stream.

new_stream_beside() therefore takes streams from torch's pool until one passes the device-side probe of
sps_streams_run_concurrently against every stream in `beside`; streams that fail stay referenced (they keep their queue
slot, so the next one maps elsewhere).  If none passes it warns once and returns the last one: results are unaffected,
only the overlap is lost.
"""
import ctypes
import os
import warnings
from typing import Sequence

import torch

_TRIES = int(os.environ.get("SPS_STREAM_TRIES", "12"))
_LIMIT_US = 2000
_REJECTED = []          # streams that shared a queue with somebody: kept alive on purpose
_SCRATCH = {}
_WARNED = False
stats = {"probes": 0, "rejected": 0}


def run_concurrently(a: torch.cuda.Stream, b: torch.cuda.Stream) -> bool:
    """True if a kernel on `b` ran while a kernel on `a` was still running (both streams are synchronised by the probe)."""
    from . import _lib
    L = _lib.load()
    dev = a.device
    if dev.index not in _SCRATCH:
        _SCRATCH[dev.index] = torch.zeros(2, dtype=torch.int32, device=dev)
    out = ctypes.c_int(0)
    with torch.cuda.device(dev):
        _lib.check(L.sps_streams_run_concurrently(ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream),
                                                  ctypes.c_void_p(_SCRATCH[dev.index].data_ptr()), _LIMIT_US, ctypes.byref(out)),
                   "streams_run_concurrently")
    stats["probes"] += 1
    return bool(out.value)


def new_stream_beside(device, beside: Sequence[torch.cuda.Stream], priority: int = 0) -> torch.cuda.Stream:
    """A stream of `device` whose kernels run concurrently with those of every stream in `beside`."""
    global _WARNED
    device = torch.device(device)
    if os.environ.get("SPS_STREAM_PROBE", "1") == "0" or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=device, priority=priority)
    st = None
    for _ in range(max(1, _TRIES)):
        st = torch.cuda.Stream(device=device, priority=priority)
        if all(s.cuda_stream != st.cuda_stream and run_concurrently(s, st) for s in beside):
            return st
        _REJECTED.append(st)
        stats["rejected"] += 1
    if not _WARNED:
        _WARNED = True
        warnings.warn("spsnet_amd: no helper stream runs concurrently with the current stream (all of them share its hardware "
                      "queue); results are unaffected but the pass loses its overlap -- set GPU_MAX_HW_QUEUES=8 (or higher) "
                      "before the process touches the GPU", RuntimeWarning)
    return st


_HELPERS = {}   # (device index, root stream handle) -> {tag: (stream, exclusive)}
_ROOT = {}      # handle of a stream handed out by helper() -> (root stream handle, root stream): helpers of helpers join the pass


def helper(device, main: torch.cuda.Stream, tag: str, exclusive: bool = False) -> torch.cuda.Stream:
    """The helper stream `tag` of the pass that runs on `main` (one per (device, main stream, tag); asked for while one of the
    pass's own helpers is current, it joins the same pass).  Every helper runs beside the pass's stream and beside the pass's
    EXCLUSIVE helpers; an exclusive helper (the FPS producer: one kernel that occupies its queue for most of a pass) runs
    beside all of the pass's streams -- anything queued behind it would wait for the whole FPS.  Passes on different main
    streams are not ordered against each other (with four hardware queues they could not be)."""
    device = torch.device(device)
    root_handle, root = _ROOT.get((device.index, main.cuda_stream), (main.cuda_stream, main))
    reg = _HELPERS.setdefault((device.index, root_handle), {})
    if tag not in reg:
        beside = [root] + [s for s, ex in reg.values() if ex or exclusive]
        if main.cuda_stream != root_handle:
            beside.append(main)
        uniq = {}
        for s_ in beside:
            uniq.setdefault(s_.cuda_stream, s_)
        st = new_stream_beside(device, list(uniq.values()))
        reg[tag] = (st, exclusive)
        _ROOT.setdefault((device.index, st.cuda_stream), (root_handle, root))
    return reg[tag][0]


def forget(device, main: torch.cuda.Stream):
    """Drop the helpers registered for the pass on `main`."""
    _HELPERS.pop((torch.device(device).index, main.cuda_stream), None)
