#!/usr/bin/env python3
"""Which parameters set the gradient tolerances of the model-level parity tests: runs the three tests of
tests/test_hip_parity.py that compare EVERY parameter gradient of a whole model (golden, live oracle, rotated frames) with
`assert_close` replaced by a recorder, and prints the largest norm-wise relative errors per test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_parity as T

rec = []
def recorder(got, ref, what="", tol=T.REL):
    rec.append((T.rel_err(got, ref), what, tol))
T.assert_close = recorder

def show(title):
    grads = sorted([r for r in rec if r[1].startswith("grad ")], reverse=True)
    others = sorted([r for r in rec if not r[1].startswith("grad ")], reverse=True)
    print(f"== {title}: {len(grads)} gradients, {sum(1 for g in grads if g[0] > 1e-4)} above 1e-4; largest:")
    for e, what, tol in grads[:6]:
        print(f"     {e:9.3e}  {what}   (test bound {tol:.0e})")
    print(f"   outputs / terms, largest: " + ", ".join(f"{w} {e:.1e}" for e, w, _ in others[:4]))
    rec.clear()

for tag in ("ncg3", "ncg6"):
    T.test_model_forward_loss_and_grads_golden(tag, True); show(f"golden {tag}")
for wl, nf, F in (("dipeptide", 4, 600), ("chignolin", 1, 128)):
    T.test_model_step_vs_live_oracle(wl, nf, F); show(f"live oracle {wl} F={F}")
T.test_rotation_equivariance_and_translation_invariance(); show("rotated frames")
