"""Drop-ins for the per-view bookkeeping LoG runs on the rasterizer's outputs -- SURVEY 8f row N4:

* ``unique_ids(point_id_pixel, n)`` = ``torch.unique(point_id_pixel, sorted=True, return_counts=True)`` with the
  leading -1 dropped (/root/reference/LoG/render/renderer.py:156-159): a count per Gaussian + an ordered
  compaction instead of a sort of all H*W ids;
* ``update_by_output(counter, output, fix_parent)`` = ``Counter.update_by_output``
  (/root/reference/LoG/model/counter.py:36-68): one kernel per view instead of ~25 indexing kernels.

Both go through liblograst (include/lograst.h: lograst_id_histogram, lograst_counter_update).  Install under an
unmodified LoG checkout with ``log_amd.counter.install()``: ``Counter.update_by_output = update_by_output``, and the
name ``torch`` inside the module ``LoG.render.renderer`` is rebound to a pass-through stand-in whose ``unique`` sends the
rasterizer's ``point_id_pixel`` map (recognised by the tag ``GaussianRasterizer.forward`` leaves on it) through
``unique_ids`` and hands every other call, and every other attribute, to the real ``torch`` -- no LoG line is edited."""
import logging
import types

import torch

from . import rasterizer as _r

COUNTER_BUFFERS = ("weights_max", "weights_sum", "grad_sum", "radii_max", "visible_count", "radii_max_max",
                   "area_sum", "create_steps")


def unique_ids(point_id_pixel, num_gaussians):
    """-> (point_id int32[k] ascending, point_count int64[k]); ``num_gaussians`` = how many Gaussians the rasterizer
    call was given (ids are indices into that list)."""
    with torch.no_grad():
        return _r._backend.id_histogram(point_id_pixel, int(num_gaussians))


def update_by_output(self, output, fix_parent=False):
    """Same signature and side effects as Counter.update_by_output: the eight counter buffers are updated in place
    and ``flag_vis`` / ``index_vis`` are stored in ``output['visibility_flag'][i]`` (counter.py:48-51)."""
    buffers = {k: getattr(self, k) for k in COUNTER_BUFFERS}
    with torch.no_grad():
        for i in range(len(output["render"])):
            vf = output["visibility_flag"][i]
            visible_index = vf["index"]
            if "index_node" in vf:
                visible_index = torch.cat([visible_index, vf["index_node"]])
            flag_vis = _r._backend.counter_update(
                buffers, visible_index, output["viewspace_points"][i].grad, output["radii"][i],
                output["point_weight"][i].data, output["point_id"][i], output["point_count"][i])
            vf["flag_vis"] = flag_vis
            vf["index_vis"] = torch.where(flag_vis)[0]


_fallback_logged = False


def torch_unique(input, *args, **kwargs):
    """``torch.unique`` as LoG/render/renderer.py:156 calls it (``sorted=True, return_counts=True`` on the rasterizer's
    per-pixel id map) through the histogram kernel: ascending ids behind a leading -1 whose count is the number of pixels
    nothing contributed to, int64 counts.  The -1 entry is ALWAYS there, with a count that may be zero -- torch.unique
    would leave it out then, but deciding that here costs a device read-back per view, and renderer.py:157-159
    (`if point_id[0] == -1`) strips the entry either way.  This stand-in is installed into that module only.
    Anything else -- other arguments, or a map that lost the rasterizer's tag on the way (`.clone()`, `.to()`, indexing
    create new tensor objects) -- is torch.unique itself; the first such fall-back on a rasterizer-shaped map is logged."""
    global _fallback_logged
    n = getattr(input, "_lograst_num_gaussians", None)
    if (n is None or args or set(kwargs) - {"sorted", "return_counts"} or not kwargs.get("return_counts", False)
            or not kwargs.get("sorted", True) or input.dtype != torch.int32):
        if (n is None and not _fallback_logged and input.dtype == torch.int32 and input.dim() == 2 and input.is_cuda
                and kwargs.get("return_counts", False)):
            _fallback_logged = True
            logging.getLogger("log_amd").warning(
                "log_amd.counter: torch.unique on an id map without the rasterizer's tag (copied / moved / indexed since the "
                "rasterizer returned it?): falling back to torch.unique (correct, slower); logged once")
        return torch.unique(input, *args, **kwargs)
    ids, counts = unique_ids(input, n)
    empty = input.numel() - counts.sum()          # pixels whose id is -1 (stays on the device)
    return torch.cat([ids.new_full((1,), -1), ids]), torch.cat([empty.reshape(1), counts])


class _TorchForRenderer(types.ModuleType):
    """What the name ``torch`` means inside LoG.render.renderer after install(): torch itself, except ``unique``."""

    def __init__(self):
        super().__init__("torch")
        self.unique = torch_unique

    def __getattr__(self, name):      # (only called for names not set on this object: everything except `unique`)
        return getattr(torch, name)


def install(renderer=True):
    """Patch the reference in place (needs LoG importable): Counter.update_by_output, and -- when the module can be
    imported (it needs cv2) -- the torch.unique call of LoG.render.renderer."""
    from LoG.model.counter import Counter
    Counter.update_by_output = update_by_output
    if renderer:
        try:
            import LoG.render.renderer as rr
        except ImportError:
            rr = None
        if rr is not None and not isinstance(rr.torch, _TorchForRenderer):
            rr.torch = _TorchForRenderer()
    return Counter
