"""Drop-in for the third-party ``simple_knn`` package (gitlab.inria.fr/bkerbl/simple-knn, un-vendored by the
reference, docs/install.md:50-51).  Only ``simple_knn._C.distCUDA2`` is used by LoG
(/root/reference/LoG/utils/file.py:88-91, LoG/model/base_gaussian.py:39-42)."""
