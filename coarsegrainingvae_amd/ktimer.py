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


# ----------------------------------------------------------------------------- in-graph section marks
class Marks:
    """GPU wall-clock marks placed in stream order by ``mark(name)`` calls in the code under measurement (model /
    trainer).  Unlike HIP events they can be captured into a hipGraph: after replaying the captured step, ``sections()``
    gives what each section of the REPLAYED step took.  Each mark is one one-thread launch (~1.5 us)."""

    def __init__(self, capacity: int = 512, device="cuda", capture_only: bool = False):
        """``capture_only``: record marks only while the stream is capturing (the eager warm-up steps of
        ``Trainer.capture`` pass through unmarked)."""
        self.buf = torch.zeros(capacity, dtype=torch.int64, device=device)
        self.names = []
        self.capture_only = capture_only

    def __enter__(self):
        global _marks
        _marks = self
        self.names = []
        return self

    def __exit__(self, *exc):
        global _marks
        _marks = None

    def sections(self):
        """[(name of the mark that ENDS the section, microseconds since the previous mark)]"""
        from . import _lib
        torch.cuda.synchronize()
        hz = float(_lib.load().cgv_timestamp_hz())
        t = self.buf[: len(self.names)].cpu().tolist()
        return [(self.names[i], (t[i] - t[i - 1]) / hz * 1e6) for i in range(1, len(t))]


_marks = None


def mark(name: str):
    m = _marks
    if m is None or (m.capture_only and not torch.cuda.is_current_stream_capturing()):
        return
    from . import _lib
    i = len(m.names)
    if i >= m.buf.numel():
        return
    m.names.append(name)
    _lib.call("cgv_timestamp", m.buf.data_ptr() + 8 * i, _lib.stream_ptr())
