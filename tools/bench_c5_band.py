"""The C5 band leg of bench.py on its own (one of 8 bands of a 4K view of 100 M Gaussians, on one GPU).
usage: python tools/bench_c5_band.py [--band K] [--bands B] [--gaussians N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--band", type=int, default=3)
    ap.add_argument("--bands", type=int, default=8)
    ap.add_argument("--gaussians", type=int, default=100_000_000)
    a = ap.parse_args()
    import torch
    import bench
    sys.argv = sys.argv[:1]
    args = bench.parse()
    import log_amd.rasterizer  # noqa: F401  (registers the drop-in packages)
    print(json.dumps(bench.c5_band(args, torch.device("cuda:0"), bands=a.bands, band=a.band, N=a.gaussians)))


if __name__ == "__main__":
    main()
