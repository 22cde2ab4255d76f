// jit.cpp — run-time specialisation of the hand-written kernel templates for plans outside
// the AOT catalog.  The kernel source is the same header set the catalog is built from
// (embedded at build time as kFusedScanSource); hiprtc instantiates it for ONE plan type for
// gfx950.  Code objects are cached per (kind, plan type) in memory and on disk.
#include "engine.hpp"

#include <hip/hiprtc.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <dlfcn.h>
#include <fstream>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_map>

namespace llkv {

#include "fused_scan_source.inc" // generated: static const char *const kFusedScanSource

namespace {
std::mutex g_jit_mu;
std::unordered_map<std::string, JitKernel> g_jit_cache;
uint64_t g_jit_compiled = 0, g_jit_from_cache = 0, g_jit_from_seed = 0; // (under g_jit_mu) where this process's plan kernels came from

const char *kind_name(JitKind k) { return k == JitKind::Scan ? "scan" : k == JitKind::Select ? "select" : k == JitKind::Project ? "project" : k == JitKind::Probe ? "probe" : k == JitKind::Reduce ? "reduce" : k == JitKind::Image ? "image" : k == JitKind::KeyBits ? "keybits" : k == JitKind::Part ? "part" : "emit"; }

std::string wrapper_source(JitKind kind, const std::string &ts) {
  std::string s = "\nusing namespace llkv;\n";
  switch (kind) {
  case JitKind::Scan:
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ScanParams p) { fused_scan_body<" + ts + ">(p); }\n";
    break;
  case JitKind::KeyBits:
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ScanParams p) { keybits_body<" + ts + ">(p); }\n";
    break;
  case JitKind::Image:
    s += "extern \"C\" __global__ __launch_bounds__(1024) void llkv_jit_a(const ScanParams p) { image_scan_body<" + ts + ">(p); }\n";
    break;
  case JitKind::Part:
    // every tile's records in partition order (part_block_threads() threads per workgroup); "<plan>;lines": the form that writes
    // whole 128-byte lines (group_part.cpp admits it: short records, ≤ 512 partitions, 1 024 threads)
    if (ts.size() > 6 && ts.compare(ts.size() - 6, 6, ";lines") == 0)
      s += "extern \"C\" __global__ __launch_bounds__(1024) void llkv_jit_a(const ScanParams p) { part_scatter_body<" + ts.substr(0, ts.size() - 6) + ", 1024, true>(p); }\n";
    else
      s += "extern \"C\" __global__ __launch_bounds__(" + std::to_string(part_block_threads()) + ") void llkv_jit_a(const ScanParams p) { part_scatter_body<" + ts + ", " +
           std::to_string(part_block_threads()) + ">(p); }\n";
    break;
  case JitKind::Select:
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ScanParams p) { select_body<" + ts + ", false>(p); }\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_b(const ScanParams p) { select_body<" + ts + ", true>(p); }\n";
    break;
  case JitKind::Project:
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ProjParams p) { project_body<" + ts + ">(p); }\n";
    break;
  case JitKind::Reduce:
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ReduceParams p) { group_reduce_body<" + ts + ">(p); }\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_b(const ReduceParams p) { group_reduce_body<" + ts + ",8>(p); }\n";
    break;
  case JitKind::Emit:
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ScanParams p) { emit_body<" + ts + ", false>(p); }\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_b(const ScanParams p) { emit_body<" + ts + ", true>(p); }\n";
    break;
  case JitKind::Probe:
    // a = the hash-table form, b = the direct (bitmap + rank) form
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_a(const ScanParams p) { probe_emit_body<" + ts + ", false>(p); }\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void llkv_jit_b(const ScanParams p) { probe_emit_body<" + ts + ", true>(p); }\n";
    break;
  }
  return s;
}

uint64_t fnv1a(const std::string &s) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (unsigned char c : s) h = (h ^ c) * 0x100000001b3ull;
  return h;
}

// Directory of cached code objects, or "" when there is no place this user alone can write to (no cache then: a
// world-writable fallback would let another local user plant a code object this process loads onto the GPU).
std::string cache_dir() {
  if (const char *e = std::getenv("LLKV_HIP_CACHE_DIR")) return e;
  if (const char *h = std::getenv("HOME")) if (*h) return std::string(h) + "/.cache/llkv_hip";
  return "";
}

// mkdir -p with mode 0700; the directory must end up owned by this user and closed to group / others
bool private_dir(const std::string &dir) {
  if (dir.empty()) return false;
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') (void)::mkdir(dir.substr(0, i).c_str(), 0700);
  struct stat st;
  if (::stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  return st.st_uid == ::geteuid() && (st.st_mode & 022) == 0;
}

// Code objects shipped beside the library (`jit_seed/` in the directory of libllkv_hip.so): what hiprtc produced for
// the plans of an earlier run, under the same key as the user's cache — source AND compiler identity, so an entry that
// does not fit this build is simply never asked for.  Read-only; as trustworthy as the library next to it.  The
// ahead-of-time catalog covers the benchmark plans; this covers the long tail a deployment (or the test suite: about a
// thousand plans, 0.35 s of hiprtc each) sees again and again.
std::string seed_dir() {
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void *>(&seed_dir), &info) || !info.dli_fname) return "";
  const std::string lib = info.dli_fname;
  const size_t cut = lib.rfind('/');
  return cut == std::string::npos ? "" : lib.substr(0, cut) + "/jit_seed";
}

// What besides the source decides the code object: the compiler and its options
std::string compile_identity() {
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  const char *extra = std::getenv("LLKV_HIP_JIT_DEFINES");
  return "hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " --offload-arch=gfx950 -O3 -std=c++17 " + (extra ? extra : "");
}

int compile_to_code(const std::string &src, std::vector<char> *code, std::string *err) {
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "llkv_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *err = "hiprtcCreateProgram failed";
    return LLKV_INTERNAL;
  }
  std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  std::string extra = std::getenv("LLKV_HIP_JIT_DEFINES") ? std::getenv("LLKV_HIP_JIT_DEFINES") : ""; // e.g. "-DLLKV_NT_LOADS=1"
  std::vector<std::string> extra_opts;
  for (size_t b = 0; b < extra.size();) {
    size_t e = extra.find(' ', b);
    if (e == std::string::npos) e = extra.size();
    if (e > b) extra_opts.push_back(extra.substr(b, e - b));
    b = e + 1;
  }
  for (auto &o : extra_opts) opts.push_back(o.c_str());
  hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    *err = "hiprtc compile failed:\n" + log;
    return LLKV_INTERNAL;
  }
  size_t code_size = 0;
  hiprtcGetCodeSize(prog, &code_size);
  code->resize(code_size);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  return LLKV_OK;
}
} // namespace

// A stored code object carries what it was compiled from behind the ELF: [identity text][its length, 8 bytes]["LLKVJIT1"].
// The file NAME is a 64-bit hash of (source, compiler identity); the identity text — kind, plan type string, a second hash
// of the source and the compiler identity — is compared on load, so a colliding or mis-copied file that happens to export
// llkv_jit_a is never run in another plan's place.
static const char kBlobMagic[8] = {'L', 'L', 'K', 'V', 'J', 'I', 'T', '1'};
static uint64_t fnv1a_alt(const std::string &s) { // independent of fnv1a: other basis, the length mixed in
  uint64_t h = 0x84222325cbf29ce4ull ^ (uint64_t)s.size();
  for (unsigned char c : s) h = (h ^ c) * 0x100000001b3ull + 0x9E3779B97F4A7C15ull;
  return h;
}
static std::string blob_identity(JitKind kind, const std::string &type_string, const std::string &src) {
  char h2[32];
  std::snprintf(h2, sizeof h2, "%016llx", (unsigned long long)fnv1a_alt(src));
  return std::string(kind_name(kind)) + "|" + type_string + "\n" + h2 + "\n" + compile_identity();
}
static void blob_append_identity(std::vector<char> *file, const std::string &id) {
  file->insert(file->end(), id.begin(), id.end());
  const uint64_t n = id.size();
  file->insert(file->end(), reinterpret_cast<const char *>(&n), reinterpret_cast<const char *>(&n) + 8);
  file->insert(file->end(), kBlobMagic, kBlobMagic + 8);
}
// strips the trailer; false when it is missing or `expect` (if given) is not what it says; *id_out = the stored text
static bool blob_take_identity(std::vector<char> *file, const std::string *expect, std::string *id_out = nullptr) {
  if (file->size() < 16 || std::memcmp(file->data() + file->size() - 8, kBlobMagic, 8) != 0) return false;
  uint64_t n = 0;
  std::memcpy(&n, file->data() + file->size() - 16, 8);
  if (n > file->size() - 16) return false;
  const std::string id(file->data() + file->size() - 16 - n, (size_t)n);
  if (expect && id != *expect) return false;
  if (id_out) *id_out = id;
  file->resize(file->size() - 16 - (size_t)n);
  return true;
}
// the seed directory is trusted like the library beside it — if nobody but its owner (this user or root) can write to it
// (mode 755 or tighter: a checkout under umask 002 makes it 775 and every plan cold-compiles; LLKV_HIP_TRACE=1 says so once)
static bool trusted_dir(const std::string &dir) {
  struct stat st;
  if (dir.empty() || ::stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  const bool ok = (st.st_uid == ::geteuid() || st.st_uid == 0) && (st.st_mode & 022) == 0;
  if (!ok && std::getenv("LLKV_HIP_TRACE"))
    std::fprintf(stderr, "[llkv_hip] jit: seed directory %s is ignored (owner %u, mode %o: it must belong to this user or root and be writable by its owner only — chmod 755)\n",
                 dir.c_str(), (unsigned)st.st_uid, (unsigned)(st.st_mode & 07777));
  return ok;
}

int jit_compile(JitKind kind, const std::string &type_string, JitKernel *out, std::string *err) {
  std::lock_guard<std::mutex> lk(g_jit_mu);
  const std::string key = std::string(kind_name(kind)) + "|" + type_string;
  auto it = g_jit_cache.find(key);
  if (it != g_jit_cache.end()) { *out = it->second; return LLKV_OK; }

  const std::string src = std::string(kFusedScanSource) + wrapper_source(kind, type_string);
  // cache key: the source AND what compiles it (hiprtc version, architecture, options) — a ROCm upgrade must not
  // resurrect code objects of the previous compiler
  char hex[32];
  std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)fnv1a(src + "\n// " + compile_identity()));
  const std::string dir = cache_dir();
  const bool cacheable = private_dir(dir);
  const std::string path = dir + "/" + hex + ".hsaco";
  const bool two = kind == JitKind::Select || kind == JitKind::Probe || kind == JitKind::Emit || kind == JitKind::Reduce;
  JitKernel k;
  auto load = [&](const std::vector<char> &code, std::string *why) -> bool {
    k = JitKernel{};
    hipError_t e = hipModuleLoadData(&k.module, code.data());
    if (e != hipSuccess) { *why = std::string("hipModuleLoadData: ") + hipGetErrorString(e); return false; }
    e = hipModuleGetFunction(&k.fn, k.module, "llkv_jit_a");
    if (e == hipSuccess && two) e = hipModuleGetFunction(&k.fn2, k.module, "llkv_jit_b");
    if (e != hipSuccess) {
      *why = std::string("hipModuleGetFunction: ") + hipGetErrorString(e);
      (void)hipModuleUnload(k.module);
      return false;
    }
    return true;
  };
  std::vector<char> code;
  bool loaded = false;
  const std::string identity = blob_identity(kind, type_string, src);
  if (cacheable) {
    std::ifstream f(path, std::ios::binary);
    if (f) code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    std::string why;
    if (!code.empty() && !(blob_take_identity(&code, &identity) && (loaded = load(code, &why)))) { // truncated / corrupt / stale / somebody else's blob: drop it and compile again
      (void)::unlink(path.c_str());
      code.clear();
    }
    g_jit_from_cache += loaded ? 1 : 0;
  }
  if (!loaded && !std::getenv("LLKV_HIP_NO_JIT_SEED")) {
    static const std::string seeds = trusted_dir(seed_dir()) ? seed_dir() : std::string();
    if (!seeds.empty()) {
      std::ifstream f(seeds + "/" + hex + ".hsaco", std::ios::binary);
      if (f) code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      std::string why;
      if (!code.empty() && !(blob_take_identity(&code, &identity) && (loaded = load(code, &why)))) code.clear(); // (left alone: the directory is not ours to clean)
      g_jit_from_seed += loaded ? 1 : 0;
    }
  }
  if (!loaded) {
    int rc = compile_to_code(src, &code, err);
    if (rc) { *err = "plan " + type_string + ": " + *err; return rc; }
    if (cacheable) {
      const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
      std::ofstream f(tmp, std::ios::binary);
      std::vector<char> file = code;
      blob_append_identity(&file, identity);
      if (f) { f.write(file.data(), (std::streamsize)file.size()); f.close(); std::rename(tmp.c_str(), path.c_str()); }
    }
    if (!load(code, err)) return LLKV_INTERNAL;
    ++g_jit_compiled;
  }
  g_jit_cache.emplace(key, k);
  *out = k;
  return LLKV_OK;
}

int jit_launch_raw(hipFunction_t fn, uint32_t grid, void *params, size_t /*bytes*/, hipStream_t stream, uint32_t block) {
  if (grid == 0) return LLKV_OK;
  void *args[] = {params};
  hipError_t e = hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, stream, args, nullptr);
  if (e != hipSuccess) return set_error(LLKV_INTERNAL, std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e));
  return LLKV_OK;
}

int jit_launch(const JitKernel &k, const ScanParams &p, hipStream_t stream) {
  ScanParams copy = p;
  return jit_launch_raw(k.fn, p.scan_grid ? p.scan_grid : p.n_tiles, &copy, sizeof copy, stream); // the host sets scan_grid for LDS-accumulator plans only
}

void jit_shutdown() {
  std::lock_guard<std::mutex> lk(g_jit_mu);
  for (auto &kv : g_jit_cache) if (kv.second.module) (void)hipModuleUnload(kv.second.module);
  g_jit_cache.clear();
}

// Where the plan kernels of this process came from so far: hiprtc, the user's cache directory, the seed directory.
extern "C" void llkv_hip_jit_stats(uint64_t *compiled, uint64_t *from_cache, uint64_t *from_seed) {
  std::lock_guard<std::mutex> lk(g_jit_mu);
  if (compiled) *compiled = g_jit_compiled;
  if (from_cache) *from_cache = g_jit_from_cache;
  if (from_seed) *from_seed = g_jit_from_seed;
}

// Does a directory of stored code objects (the seed directory, a cache) hold what hiprtc makes of the tracked kernel source
// today?  Every `every`-th file (by name order) is compiled again from the identity it carries and compared byte for byte.
// No device needed.  Returns 0 and the counts; files without an identity, of another compiler or another source count as bad.
extern "C" int llkv_hip_jit_verify_dir(const char *dir, uint32_t every, uint64_t *checked, uint64_t *bad, char *first_bad, uint64_t first_bad_cap) {
  if (checked) *checked = 0;
  if (bad) *bad = 0;
  if (first_bad && first_bad_cap) first_bad[0] = 0;
  if (!dir) return LLKV_INVALID_ARGUMENT;
  std::vector<std::string> names;
  if (DIR *d = ::opendir(dir)) {
    while (struct dirent *e = ::readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() > 6 && n.compare(n.size() - 6, 6, ".hsaco") == 0) names.push_back(n);
    }
    ::closedir(d);
  } else {
    return LLKV_NOT_FOUND;
  }
  std::sort(names.begin(), names.end());
  if (every == 0) every = 1;
  for (size_t i = 0; i < names.size(); i += every) {
    std::ifstream f(std::string(dir) + "/" + names[i], std::ios::binary);
    std::vector<char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::string id;
    bool ok = blob_take_identity(&file, nullptr, &id);
    if (ok) {
      const size_t bar = id.find('|'), nl = id.find('\n');
      ok = bar != std::string::npos && nl != std::string::npos && bar < nl;
      if (ok) {
        const std::string kind_s = id.substr(0, bar), ts = id.substr(bar + 1, nl - bar - 1);
        int kind = -1;
        for (int k = 0; k <= 8; ++k) if (kind_s == kind_name((JitKind)k)) kind = k;
        ok = kind >= 0;
        if (ok) {
          const std::string src = std::string(kFusedScanSource) + wrapper_source((JitKind)kind, ts);
          char hex[32];
          std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)fnv1a(src + "\n// " + compile_identity()));
          std::vector<char> code;
          std::string err;
          ok = names[i] == std::string(hex) + ".hsaco" && id == blob_identity((JitKind)kind, ts, src) && compile_to_code(src, &code, &err) == LLKV_OK && code == file;
        }
      }
    }
    if (checked) ++*checked;
    if (!ok) {
      if (bad) ++*bad;
      if (first_bad && first_bad_cap && !first_bad[0]) std::snprintf(first_bad, (size_t)first_bad_cap, "%s", names[i].c_str());
    }
  }
  return LLKV_OK;
}

// Refills a directory of code objects from the plans another one names: every file of `from` that carries an identity
// (whatever source or compiler it came from) names a (kind, plan type string); that plan is compiled from the tracked
// kernel source of THIS library and stored in `to` under today's key.  No device needed — after a change to the kernel
// headers the seed directory is rebuilt where the change was made (tools/refresh_jit_seed.sh rebuild) instead of by a run
// of the whole suite on a GPU box.  Files i with i % n_shards == shard (by name order): one process per shard.
extern "C" int llkv_hip_jit_rebuild_dir(const char *from, const char *to, uint32_t shard, uint32_t n_shards, uint64_t *built, uint64_t *failed) {
  if (built) *built = 0;
  if (failed) *failed = 0;
  if (!from || !to || n_shards == 0) return LLKV_INVALID_ARGUMENT;
  std::vector<std::string> names;
  if (DIR *d = ::opendir(from)) {
    while (struct dirent *e = ::readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() > 6 && n.compare(n.size() - 6, 6, ".hsaco") == 0) names.push_back(n);
    }
    ::closedir(d);
  } else {
    return LLKV_NOT_FOUND;
  }
  std::sort(names.begin(), names.end());
  for (size_t i = shard; i < names.size(); i += n_shards) {
    std::ifstream f(std::string(from) + "/" + names[i], std::ios::binary);
    std::vector<char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::string id;
    if (!blob_take_identity(&file, nullptr, &id)) continue; // not one of ours
    const size_t bar = id.find('|'), nl = id.find('\n');
    if (bar == std::string::npos || nl == std::string::npos || bar > nl) continue;
    const std::string kind_s = id.substr(0, bar), ts = id.substr(bar + 1, nl - bar - 1);
    int kind = -1;
    for (int k = 0; k <= 8; ++k) if (kind_s == kind_name((JitKind)k)) kind = k;
    if (kind < 0) continue;
    const std::string src = std::string(kFusedScanSource) + wrapper_source((JitKind)kind, ts);
    char hex[32];
    std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)fnv1a(src + "\n// " + compile_identity()));
    const std::string path = std::string(to) + "/" + hex + ".hsaco";
    struct stat st;
    if (::stat(path.c_str(), &st) == 0) { if (built) ++*built; continue; } // (two stale files may name one plan)
    std::vector<char> code;
    std::string err;
    if (compile_to_code(src, &code, &err) != LLKV_OK) { // a plan the current source no longer accepts: nothing to seed
      if (failed) ++*failed;
      continue;
    }
    blob_append_identity(&code, blob_identity((JitKind)kind, ts, src));
    const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
    std::ofstream o(tmp, std::ios::binary);
    if (o) { o.write(code.data(), (std::streamsize)code.size()); o.close(); std::rename(tmp.c_str(), path.c_str()); }
    if (built) ++*built;
  }
  return LLKV_OK;
}

// The plans a directory of code objects names, one "kind|type string" per line, sorted and unique → `list_path` (a text file that
// can be TRACKED: the seed directory itself is a build artefact, and a clean checkout rebuilds it from this list —
// llkv_hip_jit_build_list, __graft_entry__.build(), tools/refresh_jit_seed.sh list / from-list).
extern "C" int llkv_hip_jit_list_dir(const char *from, const char *list_path, uint64_t *n_plans) {
  if (n_plans) *n_plans = 0;
  if (!from || !list_path) return LLKV_INVALID_ARGUMENT;
  std::vector<std::string> ids;
  if (DIR *d = ::opendir(from)) {
    while (struct dirent *e = ::readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() <= 6 || n.compare(n.size() - 6, 6, ".hsaco") != 0) continue;
      std::ifstream f(std::string(from) + "/" + n, std::ios::binary);
      std::vector<char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      std::string id;
      if (!blob_take_identity(&file, nullptr, &id)) continue;
      const size_t nl = id.find('\n');
      if (nl != std::string::npos && id.find('|') < nl) ids.push_back(id.substr(0, nl));
    }
    ::closedir(d);
  } else {
    return LLKV_NOT_FOUND;
  }
  std::sort(ids.begin(), ids.end());
  ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
  std::ofstream o(list_path);
  if (!o) return LLKV_INTERNAL;
  for (const std::string &id : ids) o << id << "\n";
  if (n_plans) *n_plans = ids.size();
  return LLKV_OK;
}

// Compiles the plans of such a list (lines i with i % n_shards == shard; one process per shard) from the tracked kernel source of
// THIS library into `to`, under today's keys.  No device needed.  `max_seconds` > 0: stop taking new plans after that long (what
// is left compiles at run time, as any unseeded plan does).
extern "C" int llkv_hip_jit_build_list(const char *list_path, const char *to, uint32_t shard, uint32_t n_shards, double max_seconds, uint64_t *built, uint64_t *failed,
                                       uint64_t *skipped) {
  if (built) *built = 0;
  if (failed) *failed = 0;
  if (skipped) *skipped = 0;
  if (!list_path || !to || n_shards == 0) return LLKV_INVALID_ARGUMENT;
  std::ifstream in(list_path);
  if (!in) return LLKV_NOT_FOUND;
  const auto t0 = std::chrono::steady_clock::now();
  std::string line;
  for (size_t i = 0; std::getline(in, line); ++i) {
    if (i % n_shards != shard || line.empty()) continue;
    if (max_seconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_seconds) { if (skipped) ++*skipped; continue; }
    const size_t bar = line.find('|');
    if (bar == std::string::npos) continue;
    const std::string kind_s = line.substr(0, bar), ts = line.substr(bar + 1);
    int kind = -1;
    for (int k = 0; k <= 8; ++k) if (kind_s == kind_name((JitKind)k)) kind = k;
    if (kind < 0) continue;
    const std::string src = std::string(kFusedScanSource) + wrapper_source((JitKind)kind, ts);
    char hex[32];
    std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)fnv1a(src + "\n// " + compile_identity()));
    const std::string path = std::string(to) + "/" + hex + ".hsaco";
    struct stat st;
    if (::stat(path.c_str(), &st) == 0) { if (built) ++*built; continue; }
    std::vector<char> code;
    std::string err;
    if (compile_to_code(src, &code, &err) != LLKV_OK) { if (failed) ++*failed; continue; } // a plan the current source no longer accepts
    blob_append_identity(&code, blob_identity((JitKind)kind, ts, src));
    const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
    std::ofstream o(tmp, std::ios::binary);
    if (o) { o.write(code.data(), (std::streamsize)code.size()); o.close(); std::rename(tmp.c_str(), path.c_str()); }
    if (built) ++*built;
  }
  return LLKV_OK;
}

// Build check (no device needed): compiles one plan of the given kind for gfx950.
// kind: 0 scan, 1 select, 2 project.
extern "C" int llkv_hip_jit_compile_only(const char *type_string, char *log_out, uint64_t log_cap) {
  int kind = 0;
  std::string ts = type_string;
  if (ts.rfind("keybits:", 0) == 0) { kind = 7; ts = ts.substr(8); } // an EmitPlan compiled as the key-bits scan
  else if (ts.rfind("part:", 0) == 0) { kind = 8; ts = ts.substr(5); } // a shared-image lowering compiled as the partitioned GROUP BY's scatter
  else if (ts.rfind("SelPlan<", 0) == 0) kind = 1;
  else if (ts.rfind("ProjPlan<", 0) == 0) kind = 2;
  else if (ts.rfind("ProbePlan<", 0) == 0) kind = 3;
  else if (ts.rfind("EmitPlan<", 0) == 0) kind = 4;
  else if (ts.rfind("ReducePlan<", 0) == 0) kind = 5;
  else if (ts.rfind("Plan<", 0) == 0 && ((ts.size() > 3 && ts.compare(ts.size() - 3, 3, ",2>") == 0) ||
                                         (ts.size() > 5 && ts.compare(ts.size() - 5, 3, ",2,") == 0))) kind = 6; // shared-image GROUP BY [, passes]
  const std::string src = std::string(kFusedScanSource) + wrapper_source((JitKind)kind, ts);
  std::vector<char> code;
  std::string err;
  int rc = compile_to_code(src, &code, &err);
  if (log_out && log_cap) {
    size_t m = std::min<size_t>(err.size(), log_cap - 1);
    std::memcpy(log_out, err.data(), m);
    log_out[m] = 0;
  }
  return rc;
}

} // namespace llkv
