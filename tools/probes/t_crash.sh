python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "test_cgvae_trains_with_the_equivariant_decoder" 2>&1 | tail -3
echo "--- with fused_loss_tail=0"
python - <<'P' 2>&1 | tail -5
import sys; sys.path.insert(0, "tests")
from coarsegrainingvae_amd import options
options.set("fused_loss_tail", 0)
import test_hip_parity as t
t.test_cgvae_trains_with_the_equivariant_decoder()
print("OK with fused_loss_tail=0")
P
