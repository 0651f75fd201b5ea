"""Which physical compute units does a CU-masked HIP stream reach?  For a few masks, launch a spinning grid on a stream
made by sps_stream_create_cu_mask and print the set of (XCD, SE, CU) the workgroups report.  GPU box only."""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spsnet_amd import _lib

L = _lib.load()


def masked_stream(bits, words=8):
    mask = (ctypes.c_uint * words)()
    for b in bits:
        mask[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    _lib.check(L.sps_stream_create_cu_mask(words, ctypes.cast(mask, ctypes.c_void_p), ctypes.byref(st)), "cu mask")
    return st


def where(stream_ptr, blocks=2048, threads=256, spin=200):
    out = torch.zeros((blocks, 2), dtype=torch.int32, device="cuda")
    _lib.check(L.sps_debug_where(blocks, threads, spin, out.data_ptr(), stream_ptr), "where")
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.uint32)
    xcc = o[:, 0] & 0xF
    cu, sh, se = (o[:, 1] >> 8) & 0xF, (o[:, 1] >> 12) & 1, (o[:, 1] >> 13) & 0x7
    return sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    full = where(None)
    print("unmasked: %d distinct (xcc, se, sh, cu); per xcc:" % len(full), np.bincount([w[0] for w in full]))
    for name, bits in (("bit 0", [0]), ("bit 1", [1]), ("bit 8", [8]), ("bit 32", [32]), ("bits 0-7", range(8)),
                       ("bits 0-15", range(16)), ("bits 0-31", range(32)), ("bits 32-63", range(32, 64)),
                       ("bits 16-255", range(16, 256)), ("bits 0-255", range(256))):
        st = masked_stream(list(bits))
        w = where(st)
        print(f"{name:12s}: {len(w):3d} CUs; per xcc {np.bincount([x[0] for x in w], minlength=8).tolist()}; first {w[:6]}")
        L.sps_stream_destroy(st)
