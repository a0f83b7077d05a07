"""Optional per-kernel HIP-event timing around C-ABI launches (used by bench.py's roofline leg).
Events are recorded on the stream the kernel is launched on (torch's current stream)."""
from __future__ import annotations

from collections import defaultdict

import torch

_active = None


class KernelTimer:
    def __init__(self, prefixes=None):
        self.prefixes = tuple(prefixes) if prefixes else None
        self.events = defaultdict(list)

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = dict(launches=len(ms), avg_us=1e3 * sum(ms) / max(len(ms), 1), total_ms=sum(ms))
        return out


def begin(name):
    t = _active
    if t is None or (t.prefixes is not None and not name.startswith(t.prefixes)):
        return None
    a = torch.cuda.Event(enable_timing=True)
    a.record()
    return (t, name, a)


def end(tok):
    if tok is None:
        return
    t, name, a = tok
    b = torch.cuda.Event(enable_timing=True)
    b.record()
    t.events[name].append((a, b))
