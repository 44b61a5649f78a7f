"""Drop-ins for the per-view bookkeeping LoG runs on the rasterizer's outputs -- SURVEY 8f row N4:

* ``unique_ids(point_id_pixel, n)`` = ``torch.unique(point_id_pixel, sorted=True, return_counts=True)`` with the
  leading -1 dropped (/root/reference/LoG/render/renderer.py:156-159): a count per Gaussian + an ordered
  compaction instead of a sort of all H*W ids;
* ``update_by_output(counter, output, fix_parent)`` = ``Counter.update_by_output``
  (/root/reference/LoG/model/counter.py:36-68): one kernel per view instead of ~25 indexing kernels.

Both go through liblograst (include/lograst.h: lograst_id_histogram, lograst_counter_update).  Install under an
unmodified LoG checkout with ``log_amd.counter.install()`` (= ``Counter.update_by_output = update_by_output``); the
``torch.unique`` call sits inside ``renderer.py`` and is replaced by editing that one line."""
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


def install():
    """Patch the reference class in place (needs LoG importable)."""
    from LoG.model.counter import Counter
    Counter.update_by_output = update_by_output
    return Counter
