"""View-sharded data parallelism for the rasterizer path (new design; the reference is single-GPU,
SURVEY.md 2.2 / 8e).

One process per GPU.  Gaussian attributes are replicated; rank r renders views {v : v % world == r};
each view's backward accumulates into ONE flat fp32 gradient buffer per rank (every attribute's ``.grad``
is a view into it), so a step needs exactly one collective: a sum of that buffer across ranks, issued
as reduce-scatter + all-gather over the Gaussian dimension (RCCL over xGMI on the MI355X node; ``gloo``
for the CPU tests).  With world_size == 1 nothing is communicated and results are bit-identical to the
single-GPU path.
"""
import torch
import torch.distributed as dist

# per-Gaussian gradient columns: means3D 3, scales 3, rotations 4, opacities 1, colors 3  (14 floats)
LAYOUT = (("means3D", 3), ("scales", 3), ("rotations", 4), ("opacities", 1), ("colors", 3))
COLS = sum(c for _, c in LAYOUT)


def shard_views(n_views, rank, world):
    """Round-robin view ownership."""
    return list(range(rank, n_views, world))


class GradientBucket:
    """Flat [P*14] fp32 buffer, attribute-major, with one contiguous [P, c] view per attribute."""

    def __init__(self, num_points, device, world=1):
        self.P = int(num_points)
        self.world = int(world)
        n = self.P * COLS
        self.pad = (-n) % max(self.world, 1)
        self.flat = torch.zeros(n + self.pad, dtype=torch.float32, device=device)
        self.views = {}
        off = 0
        for name, c in LAYOUT:
            self.views[name] = self.flat[off:off + self.P * c].view(self.P, c)
            off += self.P * c

    def attach(self, params):
        """params: dict name -> leaf tensor [P, c] (requires_grad).  Their .grad become views of the bucket,
        so autograd accumulates every view's gradient in place."""
        for name, _ in LAYOUT:
            p = params[name]
            assert p.shape == self.views[name].shape, (name, p.shape)
            p.grad = self.views[name]

    def zero(self):
        self.flat.zero_()

    def reduce(self, group=None):
        """Sum across ranks.  reduce-scatter + all-gather: every xGMI link carries 1/world of the buffer."""
        if self.world <= 1 or not dist.is_initialized():
            return self.flat
        shard = self.flat.numel() // self.world
        mine = torch.empty(shard, dtype=self.flat.dtype, device=self.flat.device)
        if dist.get_backend(group) == "gloo":
            # gloo has no reduce_scatter_tensor: same result via all_reduce (CPU tests only)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            return self.flat
        dist.reduce_scatter_tensor(mine, self.flat, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(self.flat, mine, group=group)
        return self.flat


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
