"""View-sharded data parallelism for the rasterizer path (new design; the reference is single-GPU,
SURVEY.md 2.2 / 8e).

One process per GPU.  Gaussian attributes are replicated; rank r renders views {v : v % world == r};
each view's backward accumulates into ONE flat fp32 gradient buffer per rank (every attribute's ``.grad``
is a view into it), so a step needs exactly one exchange.  Two forms:

  * ``GradientBucket.reduce()``: sum of the whole buffer across ranks, issued as reduce-scatter + all-gather over
    the Gaussian dimension (RCCL over xGMI on the MI355X node; ``gloo`` for the CPU tests) -- every rank then holds
    every gradient and steps its own replica;
  * owner-computes (``OwnerAdam``): after the reduce-scatter rank r owns rows
    [r P/world, (r+1) P/world) of every attribute; it alone runs the sparse Adam of
    /root/reference/LoG/model/sparse_optimizer.py:41-78,163-196 on them (the HIP kernel behind
    ``lograst_sparse_adam``), with moments that exist for ITS rows only -- optimizer state and optimizer work are
    divided by world -- and the updated attribute rows are all-gathered (same bytes as the gradient all-gather they
    replace).  The reference updates only the rows seen this step (``flag_vis``, sparse_optimizer.py:167): the bucket
    carries a per-row ``seen`` count next to the gradients for exactly that.

With world_size == 1 nothing is communicated and results are bit-identical to the single-GPU path.
"""
import math

import torch
import torch.distributed as dist

# per-Gaussian gradient columns: means3D 3, scales 3, rotations 4, opacities 1, colors 3  (14 floats)
LAYOUT = (("means3D", 3), ("scales", 3), ("rotations", 4), ("opacities", 1), ("colors", 3))
COLS = sum(c for _, c in LAYOUT)


def layout(sh_coeffs=0):
    """Columns of the exchange: LAYOUT (14 floats) + the SH coefficients [K, 3] when the model has them (SURVEY 8e:
    +9 at degree 1, +45 at degree 3)."""
    return LAYOUT + ((("shs", 3 * int(sh_coeffs)),) if sh_coeffs else ())


def shard_views(n_views, rank, world):
    """Round-robin view ownership."""
    return list(range(rank, n_views, world))


def rows_per_rank(num_points, world):
    return (int(num_points) + max(int(world), 1) - 1) // max(int(world), 1)


def _shape(name, rows, cols):
    return (rows, cols // 3, 3) if name == "shs" else (rows, cols)


class _Flat:
    """One flat fp32 buffer holding every attribute as a contiguous [P_pad, c] block, P_pad = world * ceil(P / world), so
    that rank r's rows [r * Pr, (r + 1) * Pr) are a contiguous slice of every block."""

    def __init__(self, num_points, device, world=1, sh_coeffs=0):
        self.P, self.world = int(num_points), max(int(world), 1)
        self.layout = layout(sh_coeffs)
        self.Pr = rows_per_rank(self.P, self.world)
        self.Ppad = self.Pr * self.world
        self.cols = sum(c for _, c in self.layout)
        self.flat = torch.zeros(self.Ppad * self.cols, dtype=torch.float32, device=device)
        self.blocks, self.views, off = {}, {}, 0
        for name, c in self.layout:
            blk = self.flat[off:off + self.Ppad * c]
            self.blocks[name] = blk                                           # [P_pad * c], padding rows included
            self.views[name] = blk[:self.P * c].view(_shape(name, self.P, c))  # what the kernels see
            off += self.Ppad * c

    def rows(self, name, rank):
        """Rank `rank`'s rows of attribute `name`: a contiguous [Pr, c] view."""
        c = dict(self.layout)[name]
        return self.blocks[name][rank * self.Pr * c:(rank + 1) * self.Pr * c].view(self.Pr, c)


class GradientBucket(_Flat):
    """Flat gradient buffer, attribute-major, with one contiguous view per attribute, and the per-row `seen` count."""

    def __init__(self, num_points, device, world=1, sh_coeffs=0):
        super().__init__(num_points, device, world, sh_coeffs)
        self.pad = self.flat.numel() - self.P * self.cols
        self.seen = torch.zeros(self.Ppad, dtype=torch.float32, device=device)   # views that saw the row this step
        self._seen_reduced = False

    def attach(self, params):
        """params: dict name -> leaf tensor (requires_grad).  Their .grad become views of the bucket,
        so autograd accumulates every view's gradient in place."""
        for name, _ in self.layout:
            p = params[name]
            assert p.shape == self.views[name].shape, (name, p.shape)
            p.grad = self.views[name]

    def zero(self):
        self.flat.zero_()
        self.seen.zero_()
        self._seen_reduced = False

    def mark_seen(self, radii, index=None):
        """Record which rows one view touched: radii > 0 (what the reference's step calls flag_vis,
        /root/reference/LoG/model/counter.py:48,50), for all rows or for the rows `index` a level-of-detail selection
        handed to the rasterizer."""
        vis = (radii > 0).to(torch.float32)
        if index is None:
            self.seen[:vis.numel()] += vis
        else:
            self.seen.index_add_(0, index, vis)

    def _sum_seen(self, group):
        if self.world > 1 and dist.is_initialized() and not self._seen_reduced:
            dist.all_reduce(self.seen, op=dist.ReduceOp.SUM, group=group)
        self._seen_reduced = True

    def reduce(self, group=None):
        """Sum across ranks.  reduce-scatter + all-gather: every xGMI link carries 1/world of the buffer."""
        if self.world <= 1 or not dist.is_initialized():
            return self.flat
        self._sum_seen(group)
        if dist.get_backend(group) == "gloo":
            # gloo has no reduce_scatter_tensor: same result via all_reduce (CPU tests only)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            return self.flat
        mine = torch.empty(self.flat.numel() // self.world, dtype=self.flat.dtype, device=self.flat.device)
        dist.reduce_scatter_tensor(mine, self.flat, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(self.flat, mine, group=group)
        return self.flat

    def reduce_scatter_rows(self, rank, group=None):
        """Owner-computes exchange, first half: -> dict name -> [Pr, c] = the sum over ranks of this rank's rows (one
        reduce-scatter per attribute block, all issued back to back), and the summed `seen` counts of all rows."""
        if self.world <= 1 or not dist.is_initialized():
            return {name: self.rows(name, 0) for name, _ in self.layout}
        self._sum_seen(group)
        out = {}
        gloo = dist.get_backend(group) == "gloo"
        for name, c in self.layout:
            if gloo:
                dist.all_reduce(self.blocks[name], op=dist.ReduceOp.SUM, group=group)
                out[name] = self.rows(name, rank)
            else:
                mine = torch.empty(self.Pr * c, dtype=torch.float32, device=self.flat.device)
                dist.reduce_scatter_tensor(mine, self.blocks[name], op=dist.ReduceOp.SUM, group=group)
                out[name] = mine.view(self.Pr, c)
        return out


class FlatParams(_Flat):
    """The Gaussian attributes themselves in the bucket's layout (``views[name]`` are the tensors to render from), so
    that the owner-computes step can publish the rows it updated with one all-gather per attribute block."""

    def __init__(self, tensors, device, world=1):
        k = int(tensors["shs"].shape[1]) if "shs" in tensors else 0
        super().__init__(next(iter(tensors.values())).shape[0], device, world, k)
        for name, _ in self.layout:
            self.views[name].copy_(tensors[name].reshape(self.views[name].shape))

    def all_gather_rows(self, rank, group=None):
        """Owner-computes exchange, second half: every rank publishes its (updated) rows of every attribute."""
        if self.world <= 1 or not dist.is_initialized():
            return
        for name, c in self.layout:
            if dist.get_backend(group) == "gloo":
                parts = [torch.empty(self.Pr * c, dtype=torch.float32) for _ in range(self.world)]
                dist.all_gather(parts, self.rows(name, rank).reshape(-1).clone(), group=group)
                self.blocks[name].copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(self.blocks[name], self.rows(name, rank).reshape(-1).clone(), group=group)


class OwnerAdam:
    """Sparse Adam on the rows this rank owns (/root/reference/LoG/model/sparse_optimizer.py:41-78,163-196): same update,
    same ``flag_vis`` rule (only rows seen this step move, and only their moments), eps 1e-15, optional amsgrad; moments
    allocated for Pr = ceil(P / world) rows only.  ``lr``: dict name -> learning rate of this step (the caller runs the
    reference's schedules: xyz / scaling follow ``get_expon_lr_func``, sparse_optimizer.py:6-39,171-177)."""
    BETA1, BETA2, EPS = 0.9, 0.999, 1e-15

    def __init__(self, params, rank, amsgrad=False):
        self.rank, self.steps = int(rank), 0
        z = lambda name, c: torch.zeros(params.Pr, c, dtype=torch.float32, device=params.flat.device)
        self.exp_avg = {name: z(name, c) for name, c in params.layout}
        self.exp_avg_sq = {name: z(name, c) for name, c in params.layout}
        self.max_exp_avg_sq = {name: z(name, c) for name, c in params.layout} if amsgrad else None
        self._index = torch.arange(params.Pr, dtype=torch.int64, device=params.flat.device)

    def step(self, bucket, params, lr, group=None):
        """One optimizer step from the gradients accumulated in `bucket` (all views of all ranks): reduce-scatter,
        Adam on the owned rows that some view saw, all-gather of the attributes.  Returns the number of rows of this
        rank that moved (a device tensor; no synchronisation)."""
        from . import rasterizer as _r
        self.steps += 1
        grads = bucket.reduce_scatter_rows(self.rank, group)
        r0 = self.rank * params.Pr
        flag = (bucket.seen[r0:r0 + params.Pr] > 0)
        bc1, bc2 = 1 - self.BETA1 ** self.steps, 1 - self.BETA2 ** self.steps
        entries = []
        for name, _ in params.layout:
            if name not in lr:
                continue
            rows = params.rows(name, self.rank)            # updated in place: "model" rows and gathered parameter are the same memory
            entries.append((rows, rows, grads[name], self.exp_avg[name], self.exp_avg_sq[name],
                            self.max_exp_avg_sq[name] if self.max_exp_avg_sq is not None else None, lr[name] / bc1))
        if entries:
            with torch.no_grad():
                _r._backend.sparse_adam(self._index, flag, entries, self.BETA1, self.BETA2, math.sqrt(bc2), self.EPS)
        params.all_gather_rows(self.rank, group)
        return flag.sum()


# ---- second axis: the image split into bands of tile rows, one band per rank ----------------------
# Every rank keeps all Gaussians and renders only its rows (log_amd.rasterizer.tile_rows clips every rect to the band
# in the projection kernel, so a rank bins, sorts and composites only what reaches its rows); no compositing across
# ranks is needed.  Exchange per view: an all-gather of the bands (3 * H * W * 4 bytes in total) and the same gradient
# sum as above.
TILE = 16


def band_rows(rank, world, height):
    """Tile rows [begin, end) owned by `rank`: contiguous, sizes differing by at most one row of tiles."""
    gy = (int(height) + TILE - 1) // TILE
    return rank * gy // world, (rank + 1) * gy // world


def band_pixels(rank, world, height):
    b, e = band_rows(rank, world, height)
    return b * TILE, min(e * TILE, int(height))


def gather_bands(image, rank, world, group=None):
    """image: [C, H, W] with this rank's band rendered (other rows: anything).  Returns the full image on every rank.
    Bands are padded to a common height for the collective (all_gather needs equal shapes)."""
    C, H, W = image.shape
    if world <= 1 or not dist.is_initialized():
        return image
    rows = [band_pixels(r, world, H) for r in range(world)]
    hmax = max(e - b for b, e in rows)
    b, e = rows[rank]
    mine = torch.zeros(C, hmax, W, dtype=image.dtype, device=image.device)
    mine[:, :e - b] = image[:, b:e]
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = torch.empty_like(image)
    for (rb, re), part in zip(rows, parts):
        out[:, rb:re] = part[:, :re - rb]
    return out
