"""TEST INFRASTRUCTURE: ``ShmSync`` -- trainer.GradSync's interface carried over a shared-memory wire (shm_wire.c), so
that N real processes sharing ONE GPU can run the product's data-parallel step, eagerly and inside a captured hipGraph.

A collective is three stream-ordered operations on the calling stream: device -> pinned host copy, a host function
(``hipLaunchHostFunc``: shm_wire.c's ``wire_exec`` -- the exchange between the processes), pinned host -> device copy.
Under stream capture these become two memcpy nodes and a host node, so a replayed step carries its collectives exactly
where the RCCL kernels sit on a multi-GPU node.  RCCL itself cannot be used here: it refuses several ranks on one
device, and the GPU box has one."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "shm_wire.c")
LIB = os.path.join(HERE, "libshm_wire.so")


def build() -> str:
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        tmp = LIB + f".{os.getpid()}.tmp"
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", tmp, SRC, "-lpthread"], check=True)
        os.replace(tmp, LIB)
    return LIB


def create_segment(world: int, slot_floats: int, tag: str = "") -> str:
    """Create (zero-filled) the shared segment the ranks of one job map; the caller unlinks it afterwards."""
    path = f"/dev/shm/cgv_wire_{os.getpid()}_{tag}"
    with open(path, "wb") as f:
        f.truncate(4096 + world * slot_floats * 4)
    return path


def _hip_runtime():
    """The libamdhip64 this process already runs on (torch bundles its own copy: opening another one by name would put
    a second runtime into the process)."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime is loaded in this process (initialise torch.cuda first)")


class _Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("send", C.c_void_p), ("recv", C.c_void_p), ("n", C.c_uint64)]


class _Done:
    def wait(self):
        return True

    def is_completed(self):
        return True


class ShmSync:
    """GradSync's interface (all_reduce_range / wait / all_gather / same_on_all_ranks / mean_scalar / drain)."""

    def __init__(self, rank: int, world: int, path: str, slot_floats: int, timeout_s: float = 120.0):
        self.rank, self.world, self.group = rank, world, None
        self.lib = C.CDLL(build())
        self.lib.wire_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_double]
        self.lib.wire_exec_ptr.restype = C.c_void_p
        self.lib.wire_exec.argtypes = [C.c_void_p]
        self.lib.wire_collectives.restype = C.c_ulonglong
        assert self.lib.wire_op_bytes() == C.sizeof(_Op)
        rc = self.lib.wire_init(path.encode(), rank, world, slot_floats, timeout_s)
        if rc != 0:
            raise RuntimeError(f"wire_init failed: {rc}")
        torch.cuda.init()
        self.hip = _hip_runtime()
        self.hip.hipLaunchHostFunc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.hip.hipLaunchHostFunc.restype = C.c_int
        self.fn = self.lib.wire_exec_ptr()
        self.pending = []
        self._ops = []                 # op records: referenced by queued host functions / captured host nodes -> kept
        self._host = {}                # (role, floats) -> pinned staging buffer (one stream: reuse is stream ordered)
        self.reduced = self.gathered = self.calls = 0

    # -- plumbing
    def _staging(self, role: str, n: int) -> torch.Tensor:
        key = (role, n)
        buf = self._host.get(key)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("a collective of a new size inside a capture: run the step eagerly once first "
                                   "(pinned staging cannot be allocated while capturing)")
            buf = torch.empty(max(n, 1), dtype=torch.float32).pin_memory()
            self._host[key] = buf
        return buf

    def _enqueue(self, kind: int, send: torch.Tensor, recv, n: int):
        op = _Op(kind, 0, send.data_ptr(), recv.data_ptr() if recv is not None else None, n)
        self._ops.append(op)
        rc = self.hip.hipLaunchHostFunc(torch.cuda.current_stream().cuda_stream, self.fn, C.addressof(op))
        if rc != 0:
            raise RuntimeError(f"hipLaunchHostFunc failed: {rc}")
        self.calls += 1

    def _all_reduce(self, t: torch.Tensor):
        n = t.numel()
        h = self._staging("r", n)
        h[:n].copy_(t.reshape(-1), non_blocking=True)
        self._enqueue(0, h, None, n)
        t.reshape(-1).copy_(h[:n], non_blocking=True)
        self.reduced += n

    # -- GradSync's interface
    def all_reduce_range(self, flat: torch.Tensor, lo: int, hi: int):
        self._all_reduce(flat[lo:hi])

    def wait(self):
        pass

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor):
        n = send.numel()
        assert recv.numel() == self.world * n
        hs, hr = self._staging("s", n), self._staging("g", self.world * n)
        hs[:n].copy_(send.reshape(-1), non_blocking=True)
        self._enqueue(1, hs, hr, n)
        recv.reshape(-1).copy_(hr[:self.world * n], non_blocking=True)
        self.gathered += recv.numel()
        return _Done()

    def same_on_all_ranks(self, value: int) -> bool:
        # 3 x 21 bits: exact in fp32 lanes
        parts = [float((value >> s) & 0x1FFFFF) for s in (0, 21, 42)]
        send = torch.tensor(parts, dtype=torch.float32, device="cuda")
        recv = torch.empty(self.world * 3, dtype=torch.float32, device="cuda")
        self.all_gather(recv, send)
        rows = recv.view(self.world, 3).cpu()
        return bool((rows == rows[0:1]).all())

    def mean_scalar(self, x: torch.Tensor) -> torch.Tensor:
        y = x.detach().clone().reshape(1).float()
        self._all_reduce(y)
        return (y / self.world).reshape(())

    def drain(self, timeout_s: float = 30.0):
        torch.cuda.synchronize()

    def gather_host(self, t: torch.Tensor) -> torch.Tensor:
        """Eager helper for the tests: [world, *t.shape] of every rank's (device or host) float tensor."""
        send = t.detach().float().reshape(-1).cuda()
        recv = torch.empty(self.world * send.numel(), dtype=torch.float32, device="cuda")
        self.all_gather(recv, send)
        torch.cuda.synchronize()
        return recv.view(self.world, *t.shape).cpu()

    def collectives(self) -> int:
        return int(self.lib.wire_collectives())

    def close(self):
        torch.cuda.synchronize()
        self.lib.wire_close()
