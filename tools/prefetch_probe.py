#!/usr/bin/env python3
"""Where does the per-batch update go when it is issued on a side stream?  ms/step of
  (a) replay only  (b) replay + update on the side stream, main never waits  (c) double-buffered step (waits)  (d) update alone"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coarsegrainingvae_amd as cg
from coarsegrainingvae_amd.trainer import Trainer
from coarsegrainingvae_amd.data import copy_batch_into
w = cg.data.WORKLOADS["chignolin"]
mk = lambda seed, slack: cg.data.prepare_batch({k: v for k, v in cg.synthetic_batch("chignolin", seed=seed, device="cuda").items() if not k.startswith("_")}, edge_slack=slack)
batch = mk(0, 0.25)
rot = [mk(10 + k, 0.0) for k in range(8)]
model = cg.build_model(600, w["n_rbf"], w["atom_cutoff"], w["cg_cutoff"], w["enc_nconv"], w["dec_nconv"], w["n_cgs"], seed=123).cuda()
tr = Trainer(model, lr=1e-4, beta=w["beta"], gamma=w["gamma"])
for _ in range(3):
    tr.step(batch)
tr.capture(batch, warmup=0)
tr.enable_prefetch()
cap = tr._graphs[True]
twin = cap["twin"]["batch"]
side = torch.cuda.Stream()
def timeit(fn, n=100):
    for i in range(10): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def a(i): cap["graph"].replay()
def b(i):
    cap["graph"].replay()
    with torch.cuda.stream(side):
        copy_batch_into(twin, rot[i % 8])
def c(i): tr.step(rot[i % 8], prefetch=rot[(i + 1) % 8])
def d(i): copy_batch_into(twin, rot[i % 8])
slots = (cap, cap["twin"])
state = {"ev": None}
from coarsegrainingvae_amd import _lib
tsbuf = torch.zeros(4, dtype=torch.int64, device="cuda")
def c_variant(wait_done, wait_load, tick=False):
    def fn(i):
        main = torch.cuda.current_stream()
        s_ = i % 2
        if wait_load and state["ev"] is not None:
            main.wait_event(state["ev"])
        slots[s_]["graph"].replay()
        if tick:
            _lib.call("cgv_timestamp", tsbuf.data_ptr(), _lib.stream_ptr())
        done = torch.cuda.Event(); done.record(main)
        slots[s_]["done2"] = done
        other = slots[1 - s_]
        if wait_done and other.get("done2") is not None:
            side.wait_event(other["done2"])
        with torch.cuda.stream(side):
            copy_batch_into(other["batch"], rot[(i + 1) % 8])
            ev = torch.cuda.Event(); ev.record(side); state["ev"] = ev
    return fn
def e(i):
    t0 = time.perf_counter(); copy_batch_into(twin, rot[i % 8]); host.append(time.perf_counter() - t0)
host = []
gB = cap["twin"]["graph"]
def f(i): (cap["graph"] if i % 2 == 0 else gB).replay()
ev = [torch.cuda.Event() for _ in range(4)]
def g(i):
    torch.cuda.current_stream().wait_event(ev[i % 4]); cap["graph"].replay(); ev[(i + 1) % 4].record()
print(f"(f) alternate the two graphs, no update   {timeit(f):.3f} ms")
for e_ in ev: e_.record()
print(f"(g) one graph + event wait / record       {timeit(g):.3f} ms")
print(f"(a) replay only                         {timeit(a):.3f} ms")
print(f"(b) replay + side-stream update, no wait {timeit(b):.3f} ms")
print(f"(c) double-buffered step                 {timeit(c):.3f} ms")
print(f"(c1) no wait on done, no wait on load     {timeit(c_variant(False, False)):.3f} ms")
print(f"(c2) wait on done only                   {timeit(c_variant(True, False)):.3f} ms")
print(f"(c3) wait on load only                   {timeit(c_variant(False, True)):.3f} ms")
print(f"(c4) both (= double-buffered step)       {timeit(c_variant(True, True)):.3f} ms")
print(f"(c5) both, a tiny kernel before the event {timeit(c_variant(True, True, True)):.3f} ms")
print(f"(d) update alone (main stream)           {timeit(d):.3f} ms")
timeit(e, 50); print(f"    host time to ISSUE one update        {1e3 * sum(host[10:]) / len(host[10:]):.3f} ms")
