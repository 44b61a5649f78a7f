#!/bin/bash
# usage: tools/r02_c3.sh "ENV=.." ...  -> C3 (LoD tree, SH1, 4 views) kernel breakdown per setting
cd "$(dirname "$0")/.." || exit 1
for envs in "$@"; do
  env $envs python - <<'PY'
import sys, os, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_log_step as B
r = B.c3_pipeline(views=4, sh_degree=int(os.environ.get("SHD", "1")))
print(os.environ.get("TAGX",""), "ms/view %.3f raster %.3f" % (r["ms_per_view"], r["stages_ms"]["rasterize_fwd_bwd"]), " ".join("%s=%.0f" % kv for kv in r["kernels_us_per_view"].items() if kv[0] in ("project","count_huge","scan_tiles","fill_keys","sort_small","sort_large","sort_huge","blend_fwd","blend_bwd","project_bwd","rebase_slots","gather_activate","activate_bwd","sparse_adam","lod_traverse")))
PY
done
