#!/usr/bin/env python3
"""Same-device A/B: cost of the in-kernel octant fold (ticket + sc1 hand-off) per workload and tile size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep import run
for rnd in range(2):
    for name, tiles in (("q1_sf10", ("65536",)), ("q6_sf10", ("4096", "8192", "16384", "32768")), ("q6_sf1", ("4096", "8192", "16384"))):
        for tile in tiles:
            for nofold in ("0", "1"):
                env = {"LLKV_HIP_FORCE_JIT": "1", "LLKV_HIP_TILE_ROWS": tile, "LLKV_HIP_JIT_DEFINES": f"-DLLKV_NO_FOLD={nofold}"}
                print(rnd, name, "tile", tile, "nofold", nofold, run(name, env), flush=True)
