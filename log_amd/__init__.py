"""lograst: the LoG rasterizer hot path (and the callers either side of it) on MI355X.  See README.md / DESIGN.md."""


def install_all():
    """Assign every drop-in method onto LoG's own classes (needs LoG importable):
    LoG.get_all, TensorTree.traverse, Counter.update_by_output, SparseOptimizer.step."""
    from . import counter, get_all, lod, sparse_optimizer
    return [m.install() for m in (get_all, lod, counter, sparse_optimizer)]
