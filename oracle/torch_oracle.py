"""Dense float64 autograd restatement of the rasterizer.  TEST INFRASTRUCTURE ONLY.

Second, independent oracle: the C oracle (lograst_oracle.c) writes the backward by hand (reverse
walk + chain rule); this file writes only the FORWARD as dense torch tensor algebra
([pixels, Gaussians] matrices in global depth order) and lets autograd derive every gradient, in
float64.  It pins the hand-written backward of both the C oracle and the HIP kernels.  Small scenes
only (O(H*W*N) memory).

Follows the same sources as the C oracle: LoG/cuda/compute_radius_kernel.cu:4-156 and
LoG/model/geometry.py:4-151 for projection/EWA, the published blend rule for compositing,
LoG/render/renderer.py:141-165 for the call contract.  Conventions fixed here:
  * min(0.99, .) passes gradient straight through (published behaviour of the third-party kernel);
  * means2D grad is d/d(ndc): pixel derivative times 0.5*W / 0.5*H;
  * rotations are NOT re-normalised (compute_radius_kernel.cu:36);
  * fork low-pass max(.,0.3) has the standard sub-gradient (0 where clamped).
"""
import torch

FILTER_NONE, FILTER_DILATE, FILTER_CLAMP = 0, 1, 2


def _rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def render(width, height, tanfovx, tanfovy, viewmatrix, projmatrix, bg, means3D, means2D, scales,
           rotations, opacities, colors, scale_modifier=1.0, filter_mode=FILTER_CLAMP, ndc_cull=True, cov3D_precomp=None):
    """Returns (image[3,H,W], radii[N] int, aux dict).  All tensor inputs float64 (cast inside).
    cov3D_precomp ([N, 6]: xx, xy, xz, yy, yz, zz) replaces scales / rotations when given (both then unused)."""
    dt = torch.float64
    V = viewmatrix.to(dt)
    P = projmatrix.to(dt)
    bg = bg.to(dt)
    p = means3D.to(dt)
    N = p.shape[0]
    W, H = int(width), int(height)
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    t = p @ V[:3, :3] + V[3, :3]
    hom = p @ P[:3, :] + P[3, :]
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None] + means2D.to(dt)[:, :2]
    tz = t[:, 2]
    vis = tz > 0.2
    if ndc_cull:
        nd = ndc.detach()
        vis = vis & (nd[:, 0] >= -1.3) & (nd[:, 0] <= 1.3) & (nd[:, 1] >= -1.3) & (nd[:, 1] <= 1.3)
    if cov3D_precomp is not None:
        c6 = cov3D_precomp.to(dt)
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                            dim=-1).reshape(-1, 3, 3)
    else:
        s = scales.to(dt) * scale_modifier
        R = _rot(rotations.to(dt))
        M = R * s[:, None, :]
        Sigma = M @ M.transpose(1, 2)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(vis, tz, torch.ones_like(tz))  # keep culled rows finite
    ux = torch.clamp(t[:, 0] / tzs, -limx, limx)
    uy = torch.clamp(t[:, 1] / tzs, -limy, limy)
    txc, tyc = ux * tzs, uy * tzs
    zero = torch.zeros_like(tzs)
    J = torch.stack([fx / tzs, zero, -(fx * txc) / (tzs * tzs),
                     zero, fy / tzs, -(fy * tyc) / (tzs * tzs)], dim=-1).reshape(-1, 2, 3)
    Rw = V[:3, :3].t()
    Tm = J @ Rw
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a, b, c = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    if filter_mode == FILTER_DILATE:
        a, c = a + 0.3, c + 0.3
    elif filter_mode == FILTER_CLAMP:
        a, c = torch.clamp_min(a, 0.3), torch.clamp_min(c, 0.3)
    det = a * c - b * b
    vis = vis & (det.detach() != 0)
    dets = torch.where(vis, det, torch.ones_like(det))
    cA, cB, cC = c / dets, -b / dets, a / dets
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    mx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    my = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mxd, myd = mx.detach(), my.detach()
    x0 = torch.clamp(torch.trunc((mxd - radius) / 16), 0, gx)
    y0 = torch.clamp(torch.trunc((myd - radius) / 16), 0, gy)
    x1 = torch.clamp(torch.trunc((mxd + radius + 15) / 16), 0, gx)
    y1 = torch.clamp(torch.trunc((myd + radius + 15) / 16), 0, gy)
    vis = vis & (((x1 - x0) * (y1 - y0)) > 0)
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    # global front-to-back order: (depth as fp32 bits, index) -- depth compared in fp32 like the kernels
    depth32 = tz.detach().to(torch.float32)
    order = torch.argsort(depth32.to(dt) * 1.0, stable=True)
    order = order[vis[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    px, py = xs.reshape(-1), ys.reshape(-1)
    tix, tiy = torch.div(px, 16, rounding_mode="floor"), torch.div(py, 16, rounding_mode="floor")
    o = order
    member = (tix[:, None] >= x0[o][None]) & (tix[:, None] < x1[o][None]) & \
             (tiy[:, None] >= y0[o][None]) & (tiy[:, None] < y1[o][None])          # [Px, K]
    dx = mx[o][None, :] - px[:, None]
    dy = my[o][None, :] - py[:, None]
    power = -0.5 * (cA[o][None] * dx * dx + cC[o][None] * dy * dy) - cB[o][None] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    raw = opacities.to(dt).reshape(-1)[o][None] * G
    alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()     # straight-through cap
    ok = member & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha_eff
    Tincl = torch.cumprod(one_m, dim=1)
    Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], dim=1)
    stop = ok & (Tincl.detach() < 1e-4)
    stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0      # true from the first stopping Gaussian on
    use = ok & ~stopped
    w = torch.where(use, alpha_eff * Texcl, torch.zeros_like(alpha_eff))
    col = colors.to(dt)[o]
    C = w @ col                                                    # [Px, 3]
    # final T = product over used Gaussians
    T_final = torch.prod(torch.where(use, one_m, torch.ones_like(one_m)), dim=1)
    img = C + T_final[:, None] * bg[None, :]
    image = img.t().reshape(3, H, W)
    K = o.shape[0]
    if K > 0:
        wmax, warg = w.detach().max(dim=1)
        pid = torch.where(wmax > 0, o[warg], torch.full_like(warg, -1))
        pw_g = torch.zeros(N, dtype=dt)
        pw_g[o] = w.detach().max(dim=0).values
        n_used = use.sum(dim=1)
    else:
        wmax = torch.zeros(H * W, dtype=dt)
        pid = torch.full((H * W,), -1, dtype=torch.int64)
        pw_g = torch.zeros(N, dtype=dt)
        n_used = torch.zeros(H * W, dtype=torch.int64)
    aux = dict(point_id_pixel=pid.reshape(H, W), point_weight_pixel=wmax.reshape(H, W), point_weight=pw_g,
               final_T=T_final.detach().reshape(H, W), n_used=n_used.reshape(H, W))
    return image, radii, aux


def sh_colors(means3D, campos, shs, degree):
    """N2 in float64 torch (autograd supplies the gradients): max(0, 0.5 + sum_k basis_k(dir) sh_k)."""
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    d = means3D - campos[None]
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = 0.5 + C0 * shs[:, 0]
    if degree > 0:
        r = r - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6] + \
            C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8]
    if degree > 2:
        r = r + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10] + \
            C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] + \
            C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14] + \
            C3[6] * x * (xx - 3 * yy) * shs[:, 15]
    return torch.clamp_min(r, 0.0)
