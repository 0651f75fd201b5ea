"""spsnet_amd -- MI355X (gfx950) implementation of SPSNet / IA-SSD's point-sampling and
set-abstraction hot path, behind the reference's `pcdet.ops.pointnet2.pointnet2_batch`
surface (reference: pcdet/ops/pointnet2/pointnet2_batch/).

Layout:
    csrc/                     hand-written HIP kernels + the C ABI (include/spsnet_sa.h)
    lib/libspsnet_sa.so       built by `make -C spsnet_amd/csrc` (or __graft_entry__.build())
    pointnet2_batch_cuda.py   the 11-function extension module the reference imports
    pointnet2_utils.py        autograd ops + QueryAndGroup / GroupAll (reference pointnet2_utils.py)
    pointnet2_modules.py      SA modules (reference pointnet2_modules.py)
    sa_stack.py               IA-SSD SA-stack driver used by bench.py / tests
    scenes.py                 deterministic synthetic KITTI-shaped clouds

There is no CPU or PyTorch fallback: importing an op module without the built library raises.
"""
__version__ = "0.1.0"


def init(device=None):
    """Optional one-time set-up of the library's per-device state (sps_init: the flag pool of the FPS sorting pre-pass).  The op
    wrappers call it themselves in front of their first FPS launch on a device; call it by hand BEFORE capturing a fresh
    process's first pass into a HIP graph (the call allocates and synchronises, which a capture must not)."""
    import torch
    from . import _lib
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return _lib.ensure_init(dev)
