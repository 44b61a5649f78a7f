"""Detail dump for one case: which Gaussians carry the gradient mismatch."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle
import gpu_util as G
from test_gpu_parity import _case
from util import rel_l2
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
cam, sc = _case(name)
bg = (0.3, 0.6, 0.9)
hf = G.hip_forward(cam, sc, bg)
v, of = G.oracle_forward(oracle, cam, sc, bg)
dL = np.random.default_rng(1).random(of["image"].shape, dtype=np.float32)
hg = G.hip_backward(hf, dL); og = oracle.backward(v, of, dL)
for k in og:
    print(k, rel_l2(hg[k], og[k]))
d = np.abs(hg["scales"].astype(np.float64) - og["scales"]).max(1)
top = np.argsort(-d)[:8]
np.set_printoptions(precision=6, linewidth=200)
for i in top:
    print("i", i, "dscale", d[i], "radii", of["radii"][i], "touched", of["tiles_touched"][i])
    print("  scales grad hip", hg["scales"][i], "ora", og["scales"][i])
    print("  conic grad hip", hg["conic"][i], "ora", og["conic"][i])
    print("  rec", of["rec"][i][:6], "scale", sc["scaling"][i], "xyz", sc["xyz"][i])
dc = np.abs(hg["conic"].astype(np.float64) - og["conic"]); print("conic max abs diff", dc.max(), "at", np.unravel_index(dc.argmax(), dc.shape))
