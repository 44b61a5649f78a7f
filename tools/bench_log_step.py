"""One LoG training view, end to end on the GPU (BASELINE.json configs[2] shape: LoD tree with level selection on, SH
colours, 1080p): select -> gather/activate -> rasterize fwd+bwd -> id histogram -> counter -> sparse Adam.
Two pipelines around the SAME rasterizer (this repo's drop-in packages):
  torch : every other stage as the reference runs it today, restated with torch ops on the device
          (tensor_tree.py:131-185, level_of_gaussian.py:65-88,262-296, activation.py:27-44, renderer.py:156-159,
          counter.py:36-68, sparse_optimizer.py:41-78,163-249);
  fused : the drop-ins of log_amd/{lod,get_all,counter,sparse_optimizer}.py.
    python tools/bench_log_step.py [roots] [levels] [sh_degree] [views] [root_scale]  -> one JSON line"""
import json
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from log_amd import lod, get_all, counter, sparse_optimizer, scenes, _lib, rasterizer as R  # noqa: E402
from log_amd.compute_radius import compute_radius_module  # noqa: E402
from lod_util import synth_tree  # noqa: E402

R0 = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
LV = int(sys.argv[2]) if len(sys.argv) > 2 else 7
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1
V = int(sys.argv[4]) if len(sys.argv) > 4 else 8
RS = float(sys.argv[5]) if len(sys.argv) > 5 else 0.03     # root scale: 3-sigma radius of a root ~ 3 * RS * 2139 / 3 px
assert D <= 1, "the torch pipeline in this tool restates the degree-1 SH terms only"
K = max((D + 1) ** 2 - 1, 3)
MIN_PX = 3.0
W, H = 1920, 1080
dev = torch.device("cuda:0")
s = synth_tree(R0, LV, 4, split_prob=0.5, hole_prob=0.02, seed=0, root_scale=RS)
P = s["xyz"].shape[0]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
gen = torch.Generator(device=dev).manual_seed(0)
bufs = {"scaling": t(s["scaling"]), "colors": torch.randn(P, 3, device=dev, generator=gen),
        "xyz": t(s["xyz"]), "opacity": torch.randn(P, 1, device=dev, generator=gen) + 1.0,
        "rotation": t(s["rotation"]), "shs": torch.randn(P, K, 3, device=dev, generator=gen) * 0.2}
keys = list(bufs)
tree = types.SimpleNamespace(node_index=t(s["node_index"]), tree=t(s["tree"]), depth=t(s["depth"]), max_level=30,
                             min_resolution_pixel=MIN_PX)
roots = t(s["root_index"])
cams = scenes.orbit_cameras(V, W=W, H=H)
wloss = torch.rand(3, H, W, device=dev)
C0, C1 = 0.28209479177387814, 0.4886025119029199
LR = {"colors": 0.0025, "shs": 0.000125, "opacity": 0.05, "rotation": 0.001}
CDT = {"weights_max": torch.float32, "weights_sum": torch.float32, "grad_sum": torch.float32, "radii_max": torch.int16,
       "visible_count": torch.int16, "radii_max_max": torch.int32, "area_sum": torch.int32, "create_steps": torch.int32}


def rasterizer_for(cam):
    T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
        bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
        projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
        debug=False)
    return GaussianRasterizer(raster_settings=rs), {"camera_center": T(cam["camera_center"])}


class State:
    """Model buffers + counter + optimizer of one pipeline (each pipeline trains its own copy)."""

    def __init__(self):
        self.bufs = {k: v.clone() for k, v in bufs.items()}
        self.gaussian = types.SimpleNamespace(
            keys=keys, active_sh_degree=D, items=lambda: ((k, self.bufs[k]) for k in keys), visibility_flag=None,
            activation=types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize),
            **self.bufs)
        self.model = types.SimpleNamespace(gaussian=self.gaussian, fix_parent=True, training=True)
        self.counter = types.SimpleNamespace(**{k: torch.zeros(P, dtype=d, device=dev) for k, d in CDT.items()})
        z = lambda: {k: torch.zeros_like(v) for k, v in self.bufs.items()}
        self.opt = types.SimpleNamespace(global_steps=torch.tensor(0., device=dev), lr_dict=dict(LR), exp_avg=z(),
                                         exp_avg_sq=z(), use_amsgrad=False, xyz_lr=None,
                                         xyz_scheduler_args=lambda st: 1.6e-4, scaling_scheduler_args=lambda st: 5e-3)
        self.model_ns = types.SimpleNamespace(**self.bufs)


def split_leaf_node(index_all):
    leaf = (tree.node_index[index_all] == -1) & (tree.depth[index_all] > 0)     # level_of_gaussian.py:244-251
    return index_all[leaf], index_all[~leaf]


# ---- torch stages ------------------------------------------------------------------------------------------------
def torch_radius(st, index, rast):
    rs = rast.raster_settings
    fx, fy = W / (2 * rs.tanfovx), H / (2 * rs.tanfovy)
    return compute_radius_module.compute_radius(st.bufs["xyz"][index], torch.exp(st.bufs["scaling"][index]),
                                                torch.nn.functional.normalize(st.bufs["rotation"][index]), rs.projmatrix,
                                                rs.viewmatrix, fx, fy, rs.tanfovx, rs.tanfovy)


def torch_select(st, rast):
    ni, tr = tree.node_index, tree.tree
    keep = (torch_radius(st, roots, rast) < MIN_PX) | (ni[roots] == -1)
    out, index, level = [roots[keep]], roots[~keep], 1
    while True:
        if level > tree.max_level:
            out.append(index)
            break
        child = tr[ni[index].long()].flatten().long()
        child = child[child != -1]
        keep = (torch_radius(st, child, rast) < MIN_PX) | (ni[child] == -1)
        out.append(child[keep])
        if (~keep).sum() == 0:
            break
        index, level = child[~keep], level + 1
    return torch.cat(out)


def torch_gather(st, index, index_node, camera):
    params = {k: torch.nn.Parameter(v[index]) for k, v in st.bufs.items()}
    full = {k: torch.cat([params[k], v[index_node]]) for k, v in st.bufs.items()}
    colors = full["colors"] * C0 + 0.5
    if D > 0:
        d = full["xyz"].detach() - camera["camera_center"][None]
        d = d / torch.norm(d, dim=-1, keepdim=True)
        x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        sh = full["shs"]
        colors = colors + (-C1 * y * sh[:, 0] + C1 * z * sh[:, 1] - C1 * x * sh[:, 2])   # degree-1 terms (D = 1 default)
    act = {"xyz": full["xyz"], "scaling": torch.exp(full["scaling"]), "opacity": torch.sigmoid(full["opacity"]),
           "rotation": torch.nn.functional.normalize(full["rotation"]), "colors": colors}
    return params, act


def torch_unique(pid):
    i, c = torch.unique(pid, sorted=True, return_counts=True)
    if i[0] == -1:
        i, c = i[1:], c[1:]
    return i, c


def torch_counter(c, visible_index, grad, radii, pw, ids, counts):
    grad_norm = torch.norm(grad[:, :2], dim=-1)
    flag_vis = radii > 0
    index_vis = torch.where(flag_vis)[0]
    pid = ids.long()
    c.area_sum[visible_index[pid]] += counts
    vvi = visible_index[index_vis]
    c.create_steps[vvi] += 1
    c.visible_count[vvi] += 1
    c.weights_max[vvi] = torch.max(c.weights_max[vvi], pw[index_vis])
    c.weights_sum[vvi] += pw[index_vis]
    c.grad_sum[visible_index[pid]] += grad_norm[pid] * counts
    c.radii_max[vvi] = torch.max(c.radii_max[vvi], radii[index_vis].short())
    c.radii_max_max[visible_index[pid]] = torch.maximum(counts.int(), c.radii_max_max[visible_index[pid]])
    return flag_vis


def torch_adam(st, index, params, flag_vis):
    opt = st.opt
    opt.global_steps += 1
    idx = index[flag_vis]
    idx.cpu()
    step = int(opt.global_steps.item())
    ea = {k: opt.exp_avg[k][idx] for k in keys}
    es = {k: opt.exp_avg_sq[k][idx] for k in keys}
    for k, param in params.items():
        if param.grad is None:
            continue
        lr = 1.6e-4 if k == "xyz" else (5e-3 if k == "scaling" else LR[k])
        p, g = param.data[flag_vis], param.grad[flag_vis]
        ea[k].mul_(0.9).add_(g, alpha=0.1)
        es[k].mul_(0.999).addcmul_(g, g, value=0.001)
        bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
        denom = (es[k].sqrt() / math.sqrt(bc2)).add_(1e-15)
        p.add_(-(lr / bc1) * (ea[k] / denom))
        st.bufs[k][idx] = p
    for k in keys:
        opt.exp_avg[k][idx] = ea[k]
        opt.exp_avg_sq[k][idx] = es[k]


# ---- one view -----------------------------------------------------------------------------------------------------
def view(st, cam_pack, fused, clock):
    rast, camera = cam_pack
    tick = clock("select")
    index_all = lod.traverse(tree, st.gaussian, roots, rast) if fused else torch_select(st, rast)
    index, index_node = split_leaf_node(index_all)
    tick = clock("gather_activate")
    if fused:
        st.gaussian.visibility_flag = {"index": index, "index_node": index_node}
        act = get_all.get_all(st.model, camera, rast)
        params = st.gaussian.visibility_flag["params"]
    else:
        params, act = torch_gather(st, index, index_node, camera)
    tick = clock("rasterize_fwd_bwd")
    means2D = torch.zeros_like(act["xyz"], requires_grad=True)
    image, radii, pid, pwp, pw = rast(means3D=act["xyz"], means2D=means2D, shs=None, colors_precomp=act["colors"],
                                      opacities=act["opacity"], scales=act["scaling"], rotations=act["rotation"],
                                      cov3D_precomp=None)
    image.backward(gradient=wloss)
    tick = clock("id_histogram")
    n = int(act["xyz"].shape[0])
    ids, counts = counter.unique_ids(pid, n) if fused else torch_unique(pid)
    tick = clock("counter")
    visible_index = torch.cat([index, index_node])
    if fused:
        out = {"render": [image], "visibility_flag": [{"index": index, "index_node": index_node}],
               "viewspace_points": [means2D], "radii": [radii], "point_weight": [pw], "point_id": [ids], "point_count": [counts]}
        counter.update_by_output(st.counter, out)
        flag_vis = out["visibility_flag"][0]["flag_vis"]
    else:
        flag_vis = torch_counter(st.counter, visible_index, means2D.grad, radii, pw, ids, counts)
    tick = clock("adam")
    flag_leaf = flag_vis[:index.shape[0]]                                    # level_of_gaussian.py:383-385
    if fused:
        sparse_optimizer.step(st.opt, st.model_ns, index, params, flag_leaf)
    else:
        torch_adam(st, index, params, flag_leaf)
    clock(None)
    return n, int(ids.numel())


def run(fused, staged):
    st = State()
    st.model_ns = types.SimpleNamespace(**st.bufs)
    packs = [rasterizer_for(c) for c in cams]
    stages = {}

    def clock(name, _s={"t": None, "name": None}):
        if staged:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if _s["name"] is not None:
                stages[_s["name"]] = stages.get(_s["name"], 0.0) + (now - _s["t"])
            _s["t"], _s["name"] = now, name

    view(st, packs[0], fused, clock)             # warm-up (allocator, lazy inits)
    stages.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = [view(st, p, fused, clock) for p in packs]
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / len(packs) * 1e3
    return total, {k: v / len(packs) * 1e3 for k, v in stages.items()}, info, st


tot_f, _, info, st_f = run(True, False)
tot_t, _, _, st_t = run(False, False)
_lib.profile_enable(True)
_lib.profile_reset()
_, stg_f, _, _ = run(True, True)
prof = _lib.profile_read()
_lib.profile_enable(False)
kern = {k: round(v[0] / (V + 1) * 1e3, 1) for k, v in prof.items()}       # us per view (V timed views + the warm-up)
inst = R.last_state_info()
_, stg_t, _, _ = run(False, True)
same = {k: float((st_f.bufs[k] - st_t.bufs[k]).norm() / st_t.bufs[k].norm()) for k in keys}
print(json.dumps({
    "bench": "log_step", "points": P, "nodes": int(s["tree"].shape[0]), "roots": R0, "tree_levels": int(s["depth"].max()),
    "sh_degree": D, "views": V, "root_scale": RS, "selected_per_view": float(np.mean([i[0] for i in info])),
    "distinct_winners_per_view": float(np.mean([i[1] for i in info])),
    "ms_per_view_fused": tot_f, "ms_per_view_torch": tot_t, "speedup": tot_t / tot_f,
    "stages_ms_fused": stg_f, "kernels_us_per_view_fused": kern,
    "last_view_tile_instances": int(inst[0]), "last_view_longest_tile_list": int(inst[2]), "stages_ms_torch": stg_t, "model_rel_l2_fused_vs_torch_after_views": same}))
