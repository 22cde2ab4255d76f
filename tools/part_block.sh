#!/usr/bin/env bash
# The partitioned GROUP BY (tools/groupby_bench.py sf10 by_partkey) with scatter workgroups of 1 024 / 512 / 256 threads, and the
# partitioned parity tests under each → gpurun_out/r03/part_block.txt.  Run on the GPU box.
set -uo pipefail
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03
: > gpurun_out/r03/part_block.txt
for b in 1024 512 256; do
  echo "## LLKV_HIP_PART_BLOCK=$b" >> gpurun_out/r03/part_block.txt
  LLKV_HIP_PART_BLOCK=$b LLKV_HIP_TRACE=1 timeout -k 10 300 python tools/groupby_bench.py sf10 by_partkey 2>&1 | tail -8 >> gpurun_out/r03/part_block.txt || exit 1
  LLKV_HIP_PART_BLOCK=$b timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "partition" 2>&1 | tail -2 >> gpurun_out/r03/part_block.txt || exit 1
done
cat gpurun_out/r03/part_block.txt
