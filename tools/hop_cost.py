"""Cost of stream hops and tiny launches on this chip/runtime (HIP-event timed, medians): what the tail of a pass pays per
kernel boundary and per cross-stream dependency.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device("cuda:0")
a = torch.zeros(64, device=dev)
big = torch.zeros(1 << 24, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def tiny(): a.add_(1.0)
def busy(): big.add_(1.0)          # ~25 us
def med(fn, reps=200):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        out.append(fn())
    return float(np.median(out)) * 1e3
def ev(): return torch.cuda.Event(enable_timing=True)
def chain(n):
    def f():
        with torch.cuda.stream(s1):
            busy(); e0 = ev(); e0.record()
            for _ in range(n): tiny()
            e1 = ev(); e1.record()
        torch.cuda.synchronize(); return e0.elapsed_time(e1)
    return f
print("same stream, 1 tiny kernel after a busy one:   %.1f us" % med(chain(1)))
print("same stream, 10 tiny kernels:                  %.1f us (%.1f each)" % (med(chain(10)), med(chain(10)) / 10))
def hop_waiting():
    with torch.cuda.stream(s1):
        busy(); e0 = ev(); e0.record(); done = torch.cuda.Event(); done.record()
    with torch.cuda.stream(s2):
        s2.wait_event(done); tiny(); e1 = ev(); e1.record()
    torch.cuda.synchronize(); return e0.elapsed_time(e1)
print("hop, consumer already waiting:                 %.1f us (producer end -> consumer's tiny kernel end)" % med(hop_waiting))
def hop_stale():
    with torch.cuda.stream(s1):
        tiny(); done = torch.cuda.Event(); done.record()
    with torch.cuda.stream(s2):
        busy(); e0 = ev(); e0.record()
        s2.wait_event(done); tiny(); e1 = ev(); e1.record()
    torch.cuda.synchronize(); return e0.elapsed_time(e1)
print("hop, event completed long before the wait:     %.1f us (consumer's previous kernel end -> tiny kernel end)" % med(hop_stale))
def no_hop():
    with torch.cuda.stream(s2):
        busy(); e0 = ev(); e0.record(); tiny(); e1 = ev(); e1.record()
    torch.cuda.synchronize(); return e0.elapsed_time(e1)
print("no hop (same as line 1, on s2):                %.1f us" % med(no_hop))
def two_events():
    with torch.cuda.stream(s2):
        busy(); e0 = ev(); e0.record(); e1 = ev(); e1.record()
    torch.cuda.synchronize(); return e0.elapsed_time(e1)
print("two timing events back to back:                %.1f us" % med(two_events))
