"""Helper streams that REALLY run beside the streams they are meant to overlap with.

The schedule of a pass (sa_stack.py: the FPS producer beside its chunked consumers; pointnet2_modules.py: the scales of
a layer beside each other) needs kernels of different HIP streams to execute concurrently.  HIP maps its streams onto a
few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and streams that share a queue are serialised; which ones share
depends on every other stream the process has created -- with RCCL initialised the producer and its consumers landed on one
queue and a pass took FPS + everything else (3.2 ms) instead of their maximum (2.3 ms) on MI355X.  No reference counterpart:
the reference launches everything on the legacy default stream.

Placement rules (a "pass" = everything issued from one main stream, its ROOT):
  * the pass's EXCLUSIVE helper -- the FPS producer, one kernel that holds its queue for most of a pass -- is placed FIRST,
    the moment the pass asks for its first helper of any kind: it must run beside the root, and every later helper beside
    it.  (Placed on demand, it would have to run beside every helper that already exists -- root + producer + two scale
    streams are the four default queues; a fifth role, or a producer asked for after them, could not be placed at all.)
  * every other helper runs beside the root and beside the exclusive helpers; among themselves they may share a queue.
  * a candidate stream is only handed out after the device-side probe (sps_streams_run_concurrently: a one-lane kernel on
    one stream waits, bounded by wall clock, for a word that a kernel launched on the other stream right behind it sets)
    has shown it running beside each of them; candidates that fail stay referenced (they keep their queue slot, so the
    next one maps elsewhere).
A lost overlap is LOUD: a helper that could not be placed (no candidate passed, or the per-device probe budget ran out) is
returned all the same -- results never depend on placement -- but it is recorded in `stats["unplaced"]`, `unplaced()` lists
it, `overlap_verified()` turns False and a RuntimeWarning is raised once; bench.py prints all three on its line.
The first helper() call of a pass synchronises the streams it probes (a set-up step: up to ~2 ms per rejected candidate).
"""
import ctypes
import os
import warnings
from typing import Sequence

import torch

_TRIES = int(os.environ.get("SPS_STREAM_TRIES", "12"))
_MAX_PROBES_PER_DEVICE = int(os.environ.get("SPS_STREAM_MAX_PROBES", "160"))
_LIMIT_US = 2000
_REJECTED = []          # streams that shared a queue with somebody: kept alive on purpose (bounded by the probe budget)
_SCRATCH = {}
_PROBES = {}            # device index -> probes spent
_WARNED = False
EXCLUSIVE_TAG = "producer"
stats = {"probes": 0, "rejected": 0, "unplaced": []}


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
    _PROBES[dev.index] = _PROBES.get(dev.index, 0) + 1
    return bool(out.value)


def _probing_off():
    return os.environ.get("SPS_STREAM_PROBE", "1") == "0" or torch.cuda.is_current_stream_capturing()


def new_stream_beside(device, beside: Sequence[torch.cuda.Stream], priority: int = 0, what: str = "helper") -> torch.cuda.Stream:
    """A stream of `device` whose kernels run concurrently with those of every stream in `beside`; a stream that could not be
    shown to is returned with `_sps_unplaced = True` and recorded in stats["unplaced"]."""
    global _WARNED
    device = torch.device(device)
    if _probing_off():
        st = torch.cuda.Stream(device=device, priority=priority)
        st._sps_unprobed = True       # handed out without a probe (set-up under capture / SPS_STREAM_PROBE=0): verifies nothing
        return st
    st = None
    for _ in range(max(1, _TRIES)):
        if _PROBES.get(device.index, 0) + len(beside) > _MAX_PROBES_PER_DEVICE:
            break                       # a long-lived process with many main streams: stop probing, say so below
        st = torch.cuda.Stream(device=device, priority=priority)
        if all(s.cuda_stream != st.cuda_stream and run_concurrently(s, st) for s in beside):
            return st
        _REJECTED.append(st)
        stats["rejected"] += 1
    if st is None:
        st = torch.cuda.Stream(device=device, priority=priority)
    st._sps_unplaced = True
    stats["unplaced"].append(f"{what}@cuda:{device.index}")
    if not _WARNED:
        _WARNED = True
        warnings.warn(f"spsnet_amd: helper stream '{what}' could not be placed on a hardware queue of its own (no candidate ran "
                      "concurrently with the pass's stream / its FPS producer, or the probe budget is spent); results are "
                      "unaffected but the pass LOSES ITS OVERLAP (FPS + everything else instead of their maximum) -- set "
                      "GPU_MAX_HW_QUEUES=8 (or higher) before the process touches the GPU", RuntimeWarning)
    return st


_HELPERS = {}   # (device index, root stream handle) -> {tag: (stream, exclusive)}
_ROOT = {}      # (device index, handle of a stream handed out by helper()) -> (root handle, root stream): helpers of helpers join the pass
_ON_FORGET = []  # callbacks(device index, root handle, [helper handles]): the callers' own caches of helper streams
# ADVICE r4 suggested not to reserve (and probe) a "producer" queue for passes that bring their own producer stream
# (sa_stack.CuFence: a CU-masked stream).  Built and measured in round 5 (profiles/round5/r5s_producer_queue_modes.txt, bench.py
# --pipelined, two CU-fenced passes in flight, same box): no reservation 1.92-2.03 ms per pass, the fence's own stream adopted as the
# exclusive helper 1.97-2.02, the reservation as it is 1.65-1.71 -- the reserved queue is what keeps the modules' scale streams off the
# hardware queue the CU-masked producer lands on.  The reservation stays; SPS_RESERVE_PRODUCER=0 switches it off (A/B).


def on_forget(callback):
    """Register a cache of helper streams: callback(device_index, root_handle, helper_handles) is called by forget()."""
    _ON_FORGET.append(callback)


def helper(device, main: torch.cuda.Stream, tag: str, exclusive: bool = False) -> torch.cuda.Stream:
    """The helper stream `tag` of the pass that runs on `main` (one per (device, main stream, tag); asked for while one of the
    pass's own helpers is current, it joins the same pass).  See the module docstring for the placement rules."""
    device = torch.device(device)
    root_handle, root = _ROOT.get((device.index, main.cuda_stream), (main.cuda_stream, main))
    key = (device.index, root_handle)
    reg = _HELPERS.get(key)
    if reg is None:
        reg = _HELPERS[key] = {}
        if not _probing_off() and os.environ.get("SPS_RESERVE_PRODUCER", "1") != "0":
            # the exclusive helper first: its queue is reserved before any other role can take the last free one
            st = new_stream_beside(device, [root], what=EXCLUSIVE_TAG)
            reg[EXCLUSIVE_TAG] = (st, True)
            _ROOT.setdefault((device.index, st.cuda_stream), (root_handle, root))
    if tag not in reg:
        beside = [root] + [s for s, ex in reg.values() if ex or exclusive]
        if main.cuda_stream != root_handle:
            beside.append(main)
        uniq = {}
        for s_ in beside:
            uniq.setdefault(s_.cuda_stream, s_)
        st = new_stream_beside(device, list(uniq.values()), what=tag)
        reg[tag] = (st, exclusive or tag == EXCLUSIVE_TAG)
        _ROOT.setdefault((device.index, st.cuda_stream), (root_handle, root))
    return reg[tag][0]


def unplaced(device=None, main: torch.cuda.Stream = None):
    """Tags of the helpers that could not be placed -- of the pass on `main`, or (no arguments) of every pass so far."""
    if main is None:
        return list(stats["unplaced"])
    device = torch.device(device)
    root_handle, _ = _ROOT.get((device.index, main.cuda_stream), (main.cuda_stream, main))
    reg = _HELPERS.get((device.index, root_handle), {})
    return [tag for tag, (st, _) in reg.items() if getattr(st, "_sps_unplaced", False)]


def overlap_verified(device, main: torch.cuda.Stream) -> bool:
    """True if every helper stream the pass on `main` has asked for so far was SHOWN to run beside the pass's stream and its FPS
    producer (probing disabled -- SPS_STREAM_PROBE=0, or set-up under stream capture -- verifies nothing: False)."""
    device = torch.device(device)
    root_handle, _ = _ROOT.get((device.index, main.cuda_stream), (main.cuda_stream, main))
    reg = _HELPERS.get((device.index, root_handle))
    if not reg or os.environ.get("SPS_STREAM_PROBE", "1") == "0":
        return False
    return not any(getattr(st, "_sps_unplaced", False) or getattr(st, "_sps_unprobed", False) for st, _ in reg.values())


def forget(device, main: torch.cuda.Stream):
    """Drop the helpers registered for the pass on `main`: the registry, the helper -> root links and, through on_forget(), the
    callers' caches of those streams (sa_stack / pointnet2_modules / backbones keep one per role).  The next helper() call of
    that main stream places a fresh set (and pays for its probes)."""
    idx = torch.device(device).index
    root_handle, _ = _ROOT.get((idx, main.cuda_stream), (main.cuda_stream, main))
    reg = _HELPERS.pop((idx, root_handle), None) or {}
    handles = [st.cuda_stream for st, _ in reg.values()]
    for h in handles:
        _ROOT.pop((idx, h), None)
    for cb in _ON_FORGET:
        cb(idx, root_handle, handles)
