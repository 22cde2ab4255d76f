#!/usr/bin/env bash
# Refills rust-llkv_amd/jit_seed/ — code objects hiprtc produced for the plans of the GPU test suite, which the library
# loads instead of compiling them again (csrc/jit.cpp: seed_dir; the directory is a build artefact like the .so files:
# git-ignored, shipped with the tree).  Needs a GPU box: run it there (gpurun), then copy gpurun_out/jit_cache back:
#   gpurun -- 'bash tools/refresh_jit_seed.sh collect'      # on the GPU box: the suite with an empty, capturable cache
#   bash tools/refresh_jit_seed.sh install                   # here: gpurun_out/jit_cache/*.hsaco → rust-llkv_amd/jit_seed/
# Entries are keyed by the kernel source AND the compiler identity: after a change to csrc/*.hip.h or a ROCm upgrade the
# old ones are never asked for again (delete them and refresh).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
case "${1:-}" in
  collect)
    rm -rf "$ROOT/gpurun_out/jit_cache"; mkdir -p "$ROOT/gpurun_out/jit_cache"; chmod 700 "$ROOT/gpurun_out/jit_cache"
    cd "$ROOT" && LLKV_HIP_NO_JIT_SEED=1 LLKV_HIP_CACHE_DIR="$ROOT/gpurun_out/jit_cache" python3 -m pytest tests -m gpu -q -x
    ls "$ROOT/gpurun_out/jit_cache" | wc -l ;;
  install)
    rm -rf "$ROOT/rust-llkv_amd/jit_seed"; mkdir -p "$ROOT/rust-llkv_amd/jit_seed"
    cp "$ROOT"/gpurun_out/jit_cache/*.hsaco "$ROOT/rust-llkv_amd/jit_seed/"
    ls "$ROOT/rust-llkv_amd/jit_seed" | wc -l ;;
  *) echo "usage: $0 collect|install" >&2; exit 2 ;;
esac
