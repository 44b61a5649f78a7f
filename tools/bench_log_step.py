"""One LoG training view, end to end on the GPU (BASELINE.json configs[2], "C3": LoD tree with level selection on, SH
colours, 1080p): select -> gather/activate -> rasterize fwd+bwd -> id histogram -> counter -> sparse Adam.
Two pipelines around the SAME rasterizer (this repo's drop-in packages):
  torch : every other stage as the reference runs it today, restated with torch ops on the device
          (tensor_tree.py:131-185, level_of_gaussian.py:65-88,262-296, activation.py:27-44, renderer.py:156-159,
          counter.py:36-68, sparse_optimizer.py:41-78,163-249);
  fused : the drop-ins of log_amd/{lod,get_all,counter,sparse_optimizer}.py.
    python tools/bench_log_step.py [roots] [levels] [sh_degree] [views] [root_scale]  -> one JSON line
Importable: ``c3_pipeline(...)`` is the C3 leg of bench.py."""
import json
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MIN_PX = 3.0
W, H = 1920, 1080
LR = {"colors": 0.0025, "shs": 0.000125, "opacity": 0.05, "rotation": 0.001}
CDT = {"weights_max": torch.float32, "weights_sum": torch.float32, "grad_sum": torch.float32, "radii_max": torch.int16,
       "visible_count": torch.int16, "radii_max_max": torch.int32, "area_sum": torch.int32, "create_steps": torch.int32}

# Real SH basis without the DC term, degrees 1..3, as (coefficient, monomial) in the order LoG stores the coefficients
# (the basis of /root/reference/LoG/model/sh_utils.py:31-68); only used by the "torch" pipeline of this tool.
_A, _B, _C = 0.4886025119029199, 1.0925484305920792, 0.5900435899266435
SH_TERMS = [
    (-_A, lambda x, y, z: y), (_A, lambda x, y, z: z), (-_A, lambda x, y, z: x),
    (_B, lambda x, y, z: x * y), (-_B, lambda x, y, z: y * z), (0.31539156525252005, lambda x, y, z: 2 * z * z - x * x - y * y),
    (-_B, lambda x, y, z: x * z), (0.5462742152960396, lambda x, y, z: x * x - y * y),
    (-_C, lambda x, y, z: y * (3 * x * x - y * y)), (2.890611442640554, lambda x, y, z: x * y * z),
    (-0.4570457994644658, lambda x, y, z: y * (4 * z * z - x * x - y * y)),
    (0.3731763325901154, lambda x, y, z: z * (2 * z * z - 3 * x * x - 3 * y * y)),
    (-0.4570457994644658, lambda x, y, z: x * (4 * z * z - x * x - y * y)),
    (1.445305721320277, lambda x, y, z: z * (x * x - y * y)), (-_C, lambda x, y, z: x * (x * x - 3 * y * y)),
]


class Workload:
    """Tree, model buffers, cameras and loss weights of one C3-shaped run (all on `dev`)."""

    def __init__(self, roots=40000, levels=7, sh_degree=3, views=8, root_scale=0.03, dev=None):
        from log_amd import scenes
        self.dev = dev or torch.device("cuda:0")
        self.D, self.V, self.R0, self.LV, self.RS = int(sh_degree), int(views), int(roots), int(levels), float(root_scale)
        assert 0 <= self.D <= 3
        self.K = max((self.D + 1) ** 2 - 1, 3)
        s = scenes.synth_tree(self.R0, self.LV, 4, split_prob=0.5, hole_prob=0.02, seed=0, root_scale=self.RS)
        self.P = P = s["xyz"].shape[0]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        gen = torch.Generator(device=self.dev).manual_seed(0)
        self.bufs = {"scaling": t(s["scaling"]), "colors": torch.randn(P, 3, device=self.dev, generator=gen),
                     "xyz": t(s["xyz"]), "opacity": torch.randn(P, 1, device=self.dev, generator=gen) + 1.0,
                     "rotation": t(s["rotation"]), "shs": torch.randn(P, self.K, 3, device=self.dev, generator=gen) * 0.2}
        self.keys = list(self.bufs)
        self.tree = types.SimpleNamespace(node_index=t(s["node_index"]), tree=t(s["tree"]), depth=t(s["depth"]),
                                          max_level=30, min_resolution_pixel=MIN_PX)
        self.num_nodes, self.tree_levels = int(s["tree"].shape[0]), int(s["depth"].max())
        self.roots = t(s["root_index"])
        self.cams = scenes.orbit_cameras(self.V, W=W, H=H)
        self.wloss = torch.rand(3, H, W, device=self.dev)

    def rasterizer_for(self, cam):
        from diff_gaussian_rasterization_wodilate import GaussianRasterizationSettings, GaussianRasterizer
        T = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=self.dev)
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["FoVx"] * 0.5), tanfovy=math.tan(cam["FoVy"] * 0.5),
            bg=T([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=T(cam["world_view_transform"]),
            projmatrix=T(cam["full_proj_transform"]), sh_degree=0, campos=T(cam["camera_center"]), prefiltered=False,
            debug=False)
        return GaussianRasterizer(raster_settings=rs), {"camera_center": T(cam["camera_center"])}


class State:
    """Model buffers + counter + optimizer of one pipeline (each pipeline trains its own copy)."""

    def __init__(self, wl):
        P, dev, D = wl.P, wl.dev, wl.D
        self.bufs = {k: v.clone() for k, v in wl.bufs.items()}
        self.gaussian = types.SimpleNamespace(
            keys=wl.keys, active_sh_degree=D, items=lambda: ((k, self.bufs[k]) for k in wl.keys), visibility_flag=None,
            activation=types.SimpleNamespace(scaling_activation=torch.exp, rotation_activation=torch.nn.functional.normalize),
            **self.bufs)
        self.model = types.SimpleNamespace(gaussian=self.gaussian, fix_parent=True, training=True)   # (+ .optimizer below, as LoG has it)
        self.counter = types.SimpleNamespace(**{k: torch.zeros(P, dtype=d, device=dev) for k, d in CDT.items()})
        z = lambda: {k: torch.zeros_like(v) for k, v in self.bufs.items()}
        self.opt = types.SimpleNamespace(global_steps=torch.tensor(0., device=dev), lr_dict=dict(LR), exp_avg=z(),
                                         exp_avg_sq=z(), use_amsgrad=False, xyz_lr=None,
                                         xyz_scheduler_args=lambda st: 1.6e-4, scaling_scheduler_args=lambda st: 5e-3)
        self.model_ns = types.SimpleNamespace(**self.bufs)
        self.model.optimizer = self.opt          # level_of_gaussian.py:352 (read by log_amd.get_all's fused step only)


def split_leaf_node(wl, index_all):
    leaf = (wl.tree.node_index[index_all] == -1) & (wl.tree.depth[index_all] > 0)     # level_of_gaussian.py:244-251
    return index_all[leaf], index_all[~leaf]


# ---- torch stages ------------------------------------------------------------------------------------------------
def torch_radius(wl, st, index, rast):
    from log_amd.compute_radius import compute_radius_module
    rs = rast.raster_settings
    fx, fy = W / (2 * rs.tanfovx), H / (2 * rs.tanfovy)
    return compute_radius_module.compute_radius(st.bufs["xyz"][index], torch.exp(st.bufs["scaling"][index]),
                                                torch.nn.functional.normalize(st.bufs["rotation"][index]), rs.projmatrix,
                                                rs.viewmatrix, fx, fy, rs.tanfovx, rs.tanfovy)


def torch_select(wl, st, rast):
    ni, tr, roots = wl.tree.node_index, wl.tree.tree, wl.roots
    keep = (torch_radius(wl, st, roots, rast) < MIN_PX) | (ni[roots] == -1)
    out, index, level = [roots[keep]], roots[~keep], 1
    while True:
        if level > wl.tree.max_level:
            out.append(index)
            break
        child = tr[ni[index].long()].flatten().long()
        child = child[child != -1]
        keep = (torch_radius(wl, st, child, rast) < MIN_PX) | (ni[child] == -1)
        out.append(child[keep])
        if (~keep).sum() == 0:
            break
        index, level = child[~keep], level + 1
    return torch.cat(out)


def torch_gather(wl, st, index, index_node, camera):
    params = {k: torch.nn.Parameter(v[index]) for k, v in st.bufs.items()}
    full = {k: torch.cat([params[k], v[index_node]]) for k, v in st.bufs.items()}
    colors = full["colors"] * 0.28209479177387814 + 0.5
    if wl.D > 0:
        d = full["xyz"].detach() - camera["camera_center"][None]
        d = d / torch.norm(d, dim=-1, keepdim=True)
        x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        for k in range((wl.D + 1) ** 2 - 1):
            c, mono = SH_TERMS[k]
            colors = colors + (c * mono(x, y, z)) * full["shs"][:, k]
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


def torch_adam(wl, st, index, params, flag_vis):
    opt = st.opt
    opt.global_steps += 1
    idx = index[flag_vis]
    idx.cpu()
    step = int(opt.global_steps.item())
    ea = {k: opt.exp_avg[k][idx] for k in wl.keys}
    es = {k: opt.exp_avg_sq[k][idx] for k in wl.keys}
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
    for k in wl.keys:
        opt.exp_avg[k][idx] = ea[k]
        opt.exp_avg_sq[k][idx] = es[k]


# ---- one view -----------------------------------------------------------------------------------------------------
def view(wl, st, cam_pack, fused, clock):
    from log_amd import lod, get_all, counter, sparse_optimizer
    rast, camera = cam_pack
    clock("select")
    index_all = lod.traverse(wl.tree, st.gaussian, wl.roots, rast) if fused else torch_select(wl, st, rast)
    index, index_node = split_leaf_node(wl, index_all)
    clock("gather_activate")
    if fused:
        st.gaussian.visibility_flag = {"index": index, "index_node": index_node}
        act = get_all.get_all(st.model, camera, rast)
        params = st.gaussian.visibility_flag["params"]
    else:
        params, act = torch_gather(wl, st, index, index_node, camera)
    clock("rasterize_fwd_bwd")
    means2D = torch.zeros_like(act["xyz"], requires_grad=True)
    image, radii, pid, pwp, pw = rast(means3D=act["xyz"], means2D=means2D, shs=None, colors_precomp=act["colors"],
                                      opacities=act["opacity"], scales=act["scaling"], rotations=act["rotation"],
                                      cov3D_precomp=None)
    image.backward(gradient=wl.wloss)
    clock("id_histogram")
    n = int(act["xyz"].shape[0])
    ids, counts = counter.unique_ids(pid, n) if fused else torch_unique(pid)
    clock("counter")
    visible_index = torch.cat([index, index_node])
    if fused:
        out = {"render": [image], "visibility_flag": [{"index": index, "index_node": index_node}],
               "viewspace_points": [means2D], "radii": [radii], "point_weight": [pw], "point_id": [ids], "point_count": [counts]}
        counter.update_by_output(st.counter, out)
        flag_vis = out["visibility_flag"][0]["flag_vis"]
    else:
        flag_vis = torch_counter(st.counter, visible_index, means2D.grad, radii, pw, ids, counts)
    clock("adam")
    flag_leaf = flag_vis[:index.shape[0]]                                    # level_of_gaussian.py:383-385
    if fused:
        sparse_optimizer.step(st.opt, st.model_ns, index, params, flag_leaf)
    else:
        torch_adam(wl, st, index, params, flag_leaf)
    clock(None)
    return n, int(ids.numel())


def run(wl, fused, staged):
    st = State(wl)
    packs = [wl.rasterizer_for(c) for c in wl.cams]
    stages, cur = {}, {"t": None, "name": None}

    def clock(name):
        if staged:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if cur["name"] is not None:
                stages[cur["name"]] = stages.get(cur["name"], 0.0) + (now - cur["t"])
            cur["t"], cur["name"] = now, name

    # warm-up over EVERY view (allocator, lazy inits): each camera selects a different number of Gaussians, and a buffer
    # size the caching allocator has not seen yet costs a hipMalloc (tens of ms for GB-sized blocks) inside the timed pass
    # -- the 10-40x outliers of round 2's capacity-hint stage clocks on a fresh box
    for p in packs:
        view(wl, st, p, fused, lambda name: None)
    stages.clear()
    cur["t"], cur["name"] = None, None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = [view(wl, st, p, fused, clock) for p in packs]
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / len(packs) * 1e3
    return total, {k: v / len(packs) * 1e3 for k, v in stages.items()}, info, st


def render_only(wl, reps=2):
    """Inference view (what the reference times around renderer.vis, apps/train.py:53-59,100-108): select -> gather /
    activate -> rasterizer forward, all under torch.no_grad().  -> ms per view (after one warm-up pass over the views)."""
    from log_amd import lod, get_all
    st = State(wl)
    st.model.training = False
    packs = [wl.rasterizer_for(c) for c in wl.cams]

    def one(pack):
        rast, camera = pack
        index_all = lod.traverse(wl.tree, st.gaussian, wl.roots, rast)
        index, index_node = split_leaf_node(wl, index_all)
        st.gaussian.visibility_flag = {"index": index, "index_node": index_node}
        act = get_all.get_all(st.model, camera, rast)
        means2D = torch.zeros_like(act["xyz"])
        return rast(means3D=act["xyz"], means2D=means2D, shs=None, colors_precomp=act["colors"], opacities=act["opacity"],
                    scales=act["scaling"], rotations=act["rotation"], cov3D_precomp=None)

    with torch.no_grad():
        for p in packs:
            one(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for p in packs:
                one(p)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (reps * len(packs)) * 1e3
    return {"ms_per_view": ms, "fps": 1e3 / ms}


def c3_pipeline(roots=40000, levels=7, sh_degree=3, views=8, root_scale=0.03, with_torch=False, dev=None,
                forward_only=False, fused_step=True):
    """-> dict for the C3 leg: ms per training view of the fused (drop-in) pipeline, its stage and kernel breakdown,
    and (with_torch) the same step with everything except the rasterizer done the reference's way in torch."""
    from log_amd import _lib, rasterizer as R
    wl = Workload(roots, levels, sh_degree, views, root_scale, dev)
    R.overflow_since_reset(wl.dev)          # (the status block's running maxima start with THIS workload's forwards)
    tot_f, _, info, st_f = run(wl, True, False)
    _lib.profile_reset()
    _lib.profile_enable(True)
    _, stg_f, _, _ = run(wl, True, True)
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    inst = R.last_state_info()
    sel = float(np.mean([i[0] for i in info]))
    out = {"workload": "C3: %d-point 4-ary LoD tree (%d roots, %d levels), SH degree %d, level selection on "
                       "(min_resolution_pixel %g), %dx%d, %d orbit views; one LoG training view = select -> gather/activate"
                       " -> rasterize fwd+bwd -> id histogram -> counter -> sparse Adam, through the drop-ins"
                       % (wl.P, wl.R0, wl.tree_levels, wl.D, MIN_PX, W, H, wl.V),
           "points": wl.P, "nodes": wl.num_nodes, "roots": wl.R0, "tree_levels": wl.tree_levels, "sh_degree": wl.D,
           "views": wl.V, "root_scale": wl.RS, "selected_per_view": sel,
           "distinct_winners_per_view": float(np.mean([i[1] for i in info])),
           "ms_per_view": tot_f, "selected_gaussians_per_s": sel / (tot_f * 1e-3),
           "stages_ms": stg_f, "kernels_us_per_view": {k: round(v[0] / (2 * wl.V) * 1e3, 1) for k, v in prof.items()},
           "last_view_tile_instances": int(inst[0]), "last_view_longest_tile_list": int(inst[2])}
    # (the "rasterize_fwd_bwd" stage clock spans image.backward(), i.e. also the activation backward of get_all: the
    # rasterizer's own kernels, from the HIP-event profile, are summed here)
    rk = ("project", "count_huge", "scan_tiles", "rebase_slots", "fill_keys", "sort_small", "sort_large", "sort_huge",
          "blend_fwd", "blend_bwd", "project_bwd")
    out["rasterizer_kernels_us_per_view"] = round(sum(out["kernels_us_per_view"].get(k, 0.0) for k in rk), 1)
    if with_torch:
        tot_t, _, _, st_t = run(wl, False, False)
        _, stg_t, _, _ = run(wl, False, True)
        out.update(ms_per_view_torch=tot_t, speedup_vs_torch=tot_t / tot_f, stages_ms_torch=stg_t,
                   model_rel_l2_fused_vs_torch_after_views={
                       k: float((st_f.bufs[k] - st_t.bufs[k]).norm() / st_t.bufs[k].norm()) for k in wl.keys})
    if forward_only:
        out["forward_only"] = render_only(wl)
    if fused_step:
        # round 6: the same view with the activation backward and sparse Adam in ONE kernel (log_amd.get_all.set_fused_step:
        # opt-in; the compact raw gradients are never written)
        from log_amd import get_all
        prev = get_all.set_fused_step(True)
        try:
            tot_s, _, _, st_s = run(wl, True, False)
            _lib.profile_reset()
            _lib.profile_enable(True)
            _, stg_s, _, _ = run(wl, True, True)
            prof_s = _lib.profile_read()
            _lib.profile_enable(False)
        finally:
            get_all.set_fused_step(prev)
        out.update(ms_per_view_fused_step=tot_s, stages_ms_fused_step=stg_s,
                   kernels_us_per_view_fused_step={k: round(v[0] / (2 * wl.V) * 1e3, 1) for k, v in prof_s.items()},
                   # (the two pipelines' rasterizer gradients differ by their atomics' summation order, so the models agree to
                   # that noise, not bit for bit; the kernels themselves are bit-identical on identical inputs: tests/test_gpu_train_ops.py)
                   fused_step_model_rel_l2_vs_unfused={
                       k: float((st_s.bufs[k] - st_f.bufs[k]).double().norm() /
                                max(float((st_f.bufs[k] - wl.bufs[k]).double().norm()), 1e-30)) for k in wl.keys})
    # the same view with the rasterizer's sync-free mode (log_amd.rasterizer.set_instance_capacity: no read-back in the
    # forward at all, one C-ABI call): capacity and longest-list hint from the views just run, +10 %, checked afterwards.
    # (Since round 3 the default forward does not stall the stream on its read-back either: the two should agree.)
    dev0 = st_f.bufs["xyz"].device
    chk = R.overflow_since_reset(dev0)
    R.set_instance_capacity(int(chk["max_instances"] * 1.1) + 1024, max_tile_len=int(chk["max_tile_len"] * 1.1) + 64)
    try:
        tot_h, _, _, _ = run(wl, True, False)
        _, stg_h, _, _ = run(wl, True, True)
        chk = R.overflow_since_reset(dev0)
        if not chk["overflowed"]:
            out.update(ms_per_view_capacity_hint=tot_h, stages_ms_capacity_hint=stg_h)
        else:
            out.update(capacity_hint_overflowed=True)
    finally:
        R.set_instance_capacity(None)
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    res = c3_pipeline(int(a[0]) if len(a) > 0 else 40000, int(a[1]) if len(a) > 1 else 7, int(a[2]) if len(a) > 2 else 3,
                      int(a[3]) if len(a) > 3 else 8, float(a[4]) if len(a) > 4 else 0.03, with_torch="--torch" in a)
    res["bench"] = "log_step"
    print(json.dumps(res))
