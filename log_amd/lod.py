"""Drop-in for ``TensorTree.traverse`` (/root/reference/LoG/model/tensor_tree.py:167-185, with the level loop
``_query_tree_torch`` :131-165 and the ``Gaussian.compute_radius`` it calls per level,
LoG/model/level_of_gaussian.py:65-88) -- SURVEY 8f row N3.

Same signature and result as the reference method: ``traverse(tree, model, root_index, camera, max_depth)`` returns
the int64 indices of the points selected for this camera, roots that are kept first, then the kept children level
by level, then whatever is left on the frontier at the depth limit.  The whole selection is one call into
liblograst (include/lograst.h: lograst_lod_traverse) and one host synchronisation.

Install under an unmodified LoG checkout with ``log_amd.lod.install()`` (= ``TensorTree.traverse = traverse``,
see INTEGRATION.md)."""
import torch

from . import rasterizer as _r


def _check_activations(model):
    act = getattr(model, "activation", None)
    if act is None:
        return
    if act.scaling_activation is not torch.exp or act.rotation_activation is not torch.nn.functional.normalize:
        raise NotImplementedError("log_amd.lod.traverse fuses the default activations (exp scales, normalised "
                                  "quaternions: LoG/model/activation.py:5-8,17); this model uses others")


def traverse(self, model, root_index, camera, max_depth=1000):
    """``self``: the TensorTree (node_index, tree, max_level, min_resolution_pixel); ``model``: the Gaussian module
    (raw xyz / scaling / rotation); ``camera``: the rasterizer whose ``raster_settings`` the reference reads at
    level_of_gaussian.py:73-80."""
    _check_activations(model)
    rs = camera.raster_settings
    fx = rs.image_width / (2.0 * rs.tanfovx)      # level_of_gaussian.py:79-80
    fy = rs.image_height / (2.0 * rs.tanfovy)
    levels = max(0, min(int(self.max_level), int(max_depth)))
    with torch.no_grad():
        return _r._backend.lod_traverse(self.node_index, self.tree, model.xyz.detach(), model.scaling.detach(),
                                        model.rotation.detach(), root_index, rs.projmatrix, rs.viewmatrix, fx, fy,
                                        rs.tanfovx, rs.tanfovy, float(self.min_resolution_pixel), levels,
                                        depth_hint=_tree_depth(self))


def _tree_depth(tree):
    """Depth of the tree, cached on the tree object and keyed on its `depth` buffer (TensorTree replaces the buffer
    whenever it splits or removes, tensor_tree.py:57-119).  Only a hint: LoG passes max_depth = 20 / 1000 for trees
    a handful of levels deep, and every level costs three launches; a stale value is detected on the device
    (lograst_lod_read: frontier_left) and the selection repeated with the full depth."""
    depth = getattr(tree, "depth", None)
    if depth is None or depth.numel() == 0:
        return None
    key = (depth.data_ptr(), int(depth.numel()))
    cached = getattr(tree, "_lograst_depth", None)
    if cached is None or cached[0] != key:
        cached = (key, int(depth.max()))
        try:
            tree._lograst_depth = cached
        except Exception:      # an object that refuses new attributes: no caching
            pass
    return cached[1]


def install():
    """Patch the reference class in place (needs LoG importable)."""
    from LoG.model.tensor_tree import TensorTree
    TensorTree.traverse = traverse
    return TensorTree
