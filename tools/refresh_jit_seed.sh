#!/usr/bin/env bash
# Refills rust-llkv_amd/jit_seed/ — code objects hiprtc produced for the plans of the GPU test suite, which the library
# loads instead of compiling them again (csrc/jit.cpp: seed_dir; the directory is a build artefact like the .so files:
# git-ignored, shipped with the tree).  Needs a GPU box: run it there (gpurun), then copy gpurun_out/jit_cache back:
#   gpurun -- 'bash tools/refresh_jit_seed.sh collect'      # on the GPU box: the suite with an empty, capturable cache
#   bash tools/refresh_jit_seed.sh install                   # here: gpurun_out/jit_cache/*.hsaco → rust-llkv_amd/jit_seed/
#   bash tools/refresh_jit_seed.sh rebuild                   # here, no GPU: the plans the present seeds name, recompiled from today's source
#   bash tools/refresh_jit_seed.sh list                      # the plans the seeds name → rust-llkv_amd/jit_seed_plans.txt (tracked)
#   bash tools/refresh_jit_seed.sh from-list                 # here, no GPU: that list compiled into a fresh seed directory (build() does it for a clean checkout)
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
    rm -rf "$ROOT/rust-llkv_amd/jit_seed"; mkdir -p "$ROOT/rust-llkv_amd/jit_seed"; chmod 755 "$ROOT/rust-llkv_amd/jit_seed" # (csrc/jit.cpp: trusted_dir)
    for f in "$ROOT"/gpurun_out/jit_cache/*.hsaco; do # only code objects that carry their identity (csrc/jit.cpp: kBlobMagic): leftovers of older runs stay behind
      [ "$(tail -c 8 "$f")" = "LLKVJIT1" ] && cp "$f" "$ROOT/rust-llkv_amd/jit_seed/"
    done
    ls "$ROOT/rust-llkv_amd/jit_seed" | wc -l ;;
  rebuild) # no GPU: every plan the old seeds (and any gpurun_out/jit_cache*) name, compiled from the tracked source of today
    cd "$ROOT" && rm -rf rust-llkv_amd/jit_seed.new && mkdir -p rust-llkv_amd/jit_seed.new && chmod 755 rust-llkv_amd/jit_seed.new
    for src in rust-llkv_amd/jit_seed gpurun_out/jit_cache*; do
      [ -d "$src" ] || continue
      for shard in 0 1 2 3 4 5 6 7; do
        python3 - "$src" "$shard" <<'PY' &
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
lib = importlib.import_module("rust-llkv_amd.runtime").lib()
built, failed = C.c_uint64(), C.c_uint64()
rc = lib.llkv_hip_jit_rebuild_dir(sys.argv[1].encode(), b"rust-llkv_amd/jit_seed.new", C.c_uint32(int(sys.argv[2])), C.c_uint32(8), C.byref(built), C.byref(failed))
print(f"{sys.argv[1]} shard {sys.argv[2]}: rc={rc} built={built.value} failed={failed.value}", flush=True)
PY
      done
      wait
    done
    rm -rf rust-llkv_amd/jit_seed && mv rust-llkv_amd/jit_seed.new rust-llkv_amd/jit_seed
    ls rust-llkv_amd/jit_seed | wc -l ;;
  list) # the plans the present seeds name → rust-llkv_amd/jit_seed_plans.txt (TRACKED: a clean checkout rebuilds its seeds from it)
    cd "$ROOT" && python3 - <<'PY'
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
lib = importlib.import_module("rust-llkv_amd.runtime").lib()
n = C.c_uint64()
rc = lib.llkv_hip_jit_list_dir(b"rust-llkv_amd/jit_seed", b"rust-llkv_amd/jit_seed_plans.txt", C.byref(n))
print(f"rc={rc} plans={n.value}")
sys.exit(1 if rc else 0)
PY
    ;;
  from-list) # no GPU: the tracked plan list, compiled from today's source into a fresh seed directory (what build() does for a clean checkout)
    cd "$ROOT" && rm -rf rust-llkv_amd/jit_seed.new && mkdir -p rust-llkv_amd/jit_seed.new && chmod 755 rust-llkv_amd/jit_seed.new
    for shard in 0 1 2 3 4 5 6 7; do
      python3 - "$shard" <<'PY' &
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
lib = importlib.import_module("rust-llkv_amd.runtime").lib()
b, f, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
rc = lib.llkv_hip_jit_build_list(b"rust-llkv_amd/jit_seed_plans.txt", b"rust-llkv_amd/jit_seed.new", C.c_uint32(int(sys.argv[1])), C.c_uint32(8), C.c_double(0.0), C.byref(b), C.byref(f), C.byref(s))
print(f"shard {sys.argv[1]}: rc={rc} built={b.value} failed={f.value}", flush=True)
PY
    done
    wait
    rm -rf rust-llkv_amd/jit_seed && mv rust-llkv_amd/jit_seed.new rust-llkv_amd/jit_seed
    ls rust-llkv_amd/jit_seed | wc -l ;;
  verify) # every seed against a fresh hiprtc build of the tracked kernel source (no GPU needed; ~0.35 s per file)
    cd "$ROOT" && python3 - <<'PY'
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
rt = importlib.import_module("rust-llkv_amd.runtime")
lib = rt.lib()
checked, bad, first = C.c_uint64(), C.c_uint64(), C.create_string_buffer(256)
every = int(os.environ.get("EVERY", "1"))
rc = lib.llkv_hip_jit_verify_dir(os.path.join(os.getcwd(), "rust-llkv_amd", "jit_seed").encode(), C.c_uint32(every), C.byref(checked), C.byref(bad), first, C.c_uint64(256))
print(f"rc={rc} checked={checked.value} bad={bad.value} first_bad={first.value.decode()}")
sys.exit(1 if rc or bad.value else 0)
PY
    ;;
  *) echo "usage: $0 collect|install|rebuild|list|from-list|verify" >&2; exit 2 ;;
esac
