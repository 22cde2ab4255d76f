#!/usr/bin/env python3
"""Second, finer sweep around the best points of tools/sweep.py (non-temporal loads on)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep import run
for name, tiles, us in (("q1_sf10", ("32768", "49152", "65536", "98304", "131072"), ("2", "4")),
                        ("q6_sf10", ("2048", "4096", "6144", "8192", "12288"), ("2", "4")),
                        ("q6_sf1", ("2048", "4096", "8192"), ("2", "4"))):
    for u in us:
        for tile in tiles:
            env = {"LLKV_HIP_FORCE_JIT": "1", "LLKV_HIP_UNROLL": u, "LLKV_HIP_TILE_ROWS": tile}
            for rep in range(2):
                print(name, "U", u, "tile", tile, run(name, env), flush=True)
