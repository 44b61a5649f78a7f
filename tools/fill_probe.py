"""Timing probe of the binning kernels on the C3 view (tree-ordered, heavy-tailed): projection, scan, fill only
(LOGRAST_STOP_AFTER_FILL=1 is set here: nothing is sorted or composited, so LOGRAST_FILL_ABLATE bits that leave keys
unwritten are safe).  Prints {kernel: us per view}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# the experiment switches exist only in -DLR_EXPERIMENTS builds (log_amd/csrc/common.hpp): build one and load it
from log_amd import build as _build
os.environ["LOGRAST_LIB"] = _build.build(variant="exp", extra_flags=["-DLR_EXPERIMENTS"], verbose=False)
os.environ["LOGRAST_STOP_AFTER_FILL"] = "1"


def main():
    import torch
    import bench_log_step as B
    from log_amd import _lib, lod, get_all
    wl = B.Workload(views=4)
    st = B.State(wl)
    st.model.training = False
    packs = [wl.rasterizer_for(c) for c in wl.cams]

    def one(pack):
        rast, camera = pack
        index_all = lod.traverse(wl.tree, st.gaussian, wl.roots, rast)
        index, index_node = B.split_leaf_node(wl, index_all)
        st.gaussian.visibility_flag = {"index": index, "index_node": index_node}
        act = get_all.get_all(st.model, camera, rast)
        means2D = torch.zeros_like(act["xyz"])
        rast(means3D=act["xyz"], means2D=means2D, shs=None, colors_precomp=act["colors"], opacities=act["opacity"],
             scales=act["scaling"], rotations=act["rotation"], cov3D_precomp=None)
        return int(act["xyz"].shape[0])

    with torch.no_grad():
        for p in packs:
            one(p)
        torch.cuda.synchronize()
        _lib.profile_reset(); _lib.profile_enable(True)
        n = [one(p) for p in packs]
        torch.cuda.synchronize()
        _lib.profile_enable(False)
    out = {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in _lib.profile_read().items()}
    out["gaussians"] = n
    out["env"] = {k: v for k, v in os.environ.items() if k.startswith("LOGRAST_")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
