// api.hip — C-ABI entry points (include/rptree_hip.h): contexts, datasets, topology,
// forest accessors and the thin wrappers around the kernels in project/split/knn.hip.
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>

#include "common.h"

namespace rpt {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int32_t fail(int32_t code, const std::string& msg) {
  g_err = msg;
  return code;
}

// ---- caching device allocator ------------------------------------------------------------
namespace {
struct Block {
  void* p;
  size_t bytes;
};
std::mutex g_pool_mu;
std::map<int, std::vector<Block>> g_pool_free;            // device -> cached blocks
std::map<void*, std::pair<int, size_t>> g_pool_live;      // ptr -> (device, bytes)
struct Pending {
  void* p;
  size_t bytes;
  int dev;
  hipStream_t stream;
};
std::vector<Pending> g_pool_pending;                      // freed, stream not yet synchronised
thread_local hipStream_t tl_stream = nullptr;
size_t round_bytes(size_t b) {
  const size_t g = b >= ((size_t)1 << 20) ? ((size_t)2 << 20) : 256;
  return (b + g - 1) / g * g;
}
}  // namespace

static hipError_t dev_alloc_impl(void** p, size_t bytes);
// allocator debugging aids, read from the environment ONCE per process (library load):
// RPT_NO_POOL = plain hipMalloc/hipFree, RPT_POOL_POISON=<byte> = fill every block handed out
// (finds reads of uninitialised memory)
static const bool g_no_pool = getenv("RPT_NO_POOL") != nullptr;
static const int g_poison = getenv("RPT_POOL_POISON") ? atoi(getenv("RPT_POOL_POISON")) : -1;
hipError_t dev_alloc(void** p, size_t bytes) {
  if (g_no_pool) return hipMalloc(p, bytes ? bytes : 1);
  const hipError_t e = dev_alloc_impl(p, bytes);
  if (e == hipSuccess && g_poison >= 0) {
    (void)hipDeviceSynchronize();
    (void)hipMemset(*p, g_poison, bytes ? bytes : 1);
    (void)hipDeviceSynchronize();
  }
  return e;
}
static hipError_t dev_alloc_impl(void** p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t want = round_bytes(bytes ? bytes : 1);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    std::vector<Block>& fl = g_pool_free[dev];
    int best = -1;
    for (int i = 0; i < (int)fl.size(); ++i)
      if (fl[i].bytes >= want && fl[i].bytes <= want + want / 4 &&
          (best < 0 || fl[i].bytes < fl[best].bytes))
        best = i;
    if (best >= 0) {
      *p = fl[best].p;
      g_pool_live[*p] = {dev, fl[best].bytes};
      fl.erase(fl.begin() + best);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) {  // give the cache back to the driver and retry once
    dev_trim();
    e = hipMalloc(p, want);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[*p] = {dev, want};
  }
  return e;
}

void dev_free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool_live.find(p);
  if (it == g_pool_live.end()) {
    (void)hipFree(p);
    return;
  }
  g_pool_pending.push_back(Pending{p, it->second.second, it->second.first, tl_stream});
  g_pool_live.erase(it);
}

void dev_set_stream(hipStream_t s) { tl_stream = s; }

hipError_t stream_sync(hipStream_t s) {
  const hipError_t e = hipStreamSynchronize(s);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_pool_pending.size();) {
    if (g_pool_pending[i].stream == s) {
      g_pool_free[g_pool_pending[i].dev].push_back(Block{g_pool_pending[i].p, g_pool_pending[i].bytes});
      g_pool_pending[i] = g_pool_pending.back();
      g_pool_pending.pop_back();
    } else {
      ++i;
    }
  }
  return e;
}

constexpr size_t kPinBytes = 8u << 20;

void* pin_alloc(rpt_ctx* ctx, size_t bytes) {
  bytes = (bytes + 63) & ~(size_t)63;
  if (bytes > kPinBytes) return nullptr;
  if (!ctx->pin) {
    if (hipHostMalloc((void**)&ctx->pin, kPinBytes, hipHostMallocDefault) != hipSuccess) {
      ctx->pin = nullptr;
      return nullptr;
    }
    ctx->pin_cap = kPinBytes;
    ctx->pin_off = 0;
  }
  if (ctx->pin_off + bytes > ctx->pin_cap) (void)ctx_sync(ctx);  // in-flight copies drain first
  void* r = ctx->pin + ctx->pin_off;
  ctx->pin_off += bytes;
  return r;
}

int32_t upload_async(rpt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (bytes == 0) return RPT_OK;
  void* stage = pin_alloc(ctx, bytes);
  if (!stage) {
    RPT_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    RPT_HIP(stream_sync(ctx->stream));
    return RPT_OK;
  }
  std::memcpy(stage, src_host, bytes);
  RPT_HIP(hipMemcpyAsync(dst_dev, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
  return RPT_OK;
}

hipError_t ctx_sync(rpt_ctx* ctx) {
  const hipError_t e = stream_sync(ctx->stream);
  ctx->pin_off = 0;
  return e;
}

// Returns the cache of the CURRENT device to the driver: after hipDeviceSynchronize every stream
// of that device has drained, so its cached blocks and its blocks waiting for a stream
// synchronisation are all idle.  Other devices (other contexts, possibly driven by other host
// threads: rpt_comm_init) are not touched — their streams were not synchronised here.
void dev_trim() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = 0; i < g_pool_pending.size();) {
    if (g_pool_pending[i].dev == dev) {
      (void)hipFree(g_pool_pending[i].p);
      g_pool_pending[i] = g_pool_pending.back();
      g_pool_pending.pop_back();
    } else {
      ++i;
    }
  }
  auto it = g_pool_free.find(dev);
  if (it != g_pool_free.end()) {
    for (Block& b : it->second) (void)hipFree(b.p);
    it->second.clear();
  }
}

void enumerate_topology(int64_t N, int32_t L, int32_t min_leaf, std::vector<Node>& out) {
  // iterative DFS, pre-order (left before right): Internal.hs:289 leaf test, :495,503 halves
  struct Item { int32_t level; int64_t heap, off, n; };
  std::vector<Item> stack;
  stack.push_back({0, 0, 0, N});
  while (!stack.empty()) {
    Item it = stack.back();
    stack.pop_back();
    bool leaf = is_leaf(it.level, it.n, L, min_leaf);
    out.push_back({it.level, it.heap, it.off, it.n, leaf});
    if (!leaf) {
      int64_t nh = it.n / 2;
      stack.push_back({it.level + 1, 2 * it.heap + 2, it.off + nh, it.n - nh});
      stack.push_back({it.level + 1, 2 * it.heap + 1, it.off, nh});
    }
  }
}

}  // namespace rpt

using namespace rpt;

static void prof_resolve(rpt_ctx* ctx);

// ---- context options ---------------------------------------------------------------------
namespace {
struct OptDesc {
  const char* name;
  int64_t rpt_options::*field;
};
const OptDesc kOptions[] = {
    {"no_stream", &rpt_options::no_stream},
    {"stream_maxnodes", &rpt_options::stream_maxnodes},
    {"stream_minper", &rpt_options::stream_minper},
    {"no_wmid", &rpt_options::no_wmid},
    {"no_midselect", &rpt_options::no_midselect},
    {"stream_big_node", &rpt_options::stream_big_node},
    {"no_wsub", &rpt_options::no_wsub},
    {"no_wsort", &rpt_options::no_wsort},
    {"no_wpack", &rpt_options::no_wpack},
    {"no_csub", &rpt_options::no_csub},
    {"no_codes", &rpt_options::no_codes},
    {"no_pcodes", &rpt_options::no_pcodes},
    {"proj_narrow", &rpt_options::proj_narrow},
    {"proj_bf16_f32", &rpt_options::proj_bf16_f32},
    {"proj_bf16_terms", &rpt_options::proj_bf16_terms},
    {"proj_bf16_codes", &rpt_options::proj_bf16_codes},
    {"proj_csr_nodense", &rpt_options::proj_csr_nodense},
    {"knn_wave", &rpt_options::knn_wave},
    {"knn_kp", &rpt_options::knn_kp},
    {"knn_no_pre32", &rpt_options::knn_no_pre32},
    {"knn_no_pre16", &rpt_options::knn_no_pre16},
    {"knn_kp16", &rpt_options::knn_kp16},
    {"knn_no_pre8", &rpt_options::knn_no_pre8},
    {"knn_kp8", &rpt_options::knn_kp8},
    {"knn_csr_pre32", &rpt_options::knn_csr_pre32},
    {"knn_general", &rpt_options::knn_general},
    {"knn_shard_old", &rpt_options::knn_shard_old},
    {"comm_force_exchange", &rpt_options::comm_force_exchange},
    {"comm_inject_failure", &rpt_options::comm_inject_failure},
    {"comm_timeout_ms", &rpt_options::comm_timeout_ms},
    {"comm_stall_test", &rpt_options::comm_stall_test},
    {"tune0", &rpt_options::tune0},
    {"tune1", &rpt_options::tune1},
    {"tune2", &rpt_options::tune2},
    {"tune3", &rpt_options::tune3},
    {"debug_host", &rpt_options::debug_host},
    {"debug_stamps", &rpt_options::debug_stamps},
};
const OptDesc* find_option(const char* name) {
  if (!name) return nullptr;
  for (const OptDesc& o : kOptions)
    if (std::strcmp(o.name, name) == 0) return &o;
  return nullptr;
}
// RPT_<NAME> in the environment seeds an option when a context is created (a variable that is
// set but holds no number means 1); nothing reads the environment after that.
void options_from_env(rpt_options& opt) {
  for (const OptDesc& o : kOptions) {
    std::string env = "RPT_";
    for (const char* c = o.name; *c; ++c) env += (char)std::toupper((unsigned char)*c);
    if (const char* v = getenv(env.c_str())) {
      char* end = nullptr;
      const long long x = std::strtoll(v, &end, 10);
      opt.*(o.field) = (end && end != v) ? (int64_t)x : 1;
    }
  }
}

// No exception leaves the library (the header's promise): the host-side planners use
// std::vector / std::string / std::map, whose failures would otherwise terminate the caller.
template <class F>
int32_t guarded(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    try {
      return fail(RPT_E_NOMEM, "out of host memory");
    } catch (...) {
      return RPT_E_NOMEM;
    }
  } catch (const std::exception& e) {
    try {
      return fail(RPT_E_INTERNAL, std::string("internal error: ") + e.what());
    } catch (...) {
      return RPT_E_INTERNAL;
    }
  } catch (...) {
    return RPT_E_INTERNAL;
  }
}
}  // namespace

extern "C" {

int32_t rpt_abi_version(void) { return RPT_ABI_VERSION; }
const char* rpt_last_error(void) { return g_err.c_str(); }

int32_t rpt_device_count(int32_t* count) {
  return guarded([&]() -> int32_t {
    RPT_ARG(count, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
      *count = 0;
      return fail(RPT_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return RPT_OK;
  });
}

int32_t rpt_ctx_create(int32_t device, rpt_ctx** out) {
  return guarded([&]() -> int32_t {
    RPT_ARG(out, "out is NULL");
    *out = nullptr;
    int c = 0;
    RPT_HIP(hipGetDeviceCount(&c));
    if (c <= 0) return fail(RPT_E_HIP, "no HIP device available (there is no CPU fallback)");
    RPT_ARG(device >= 0 && device < c, "device index out of range");
    RPT_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RPT_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(RPT_E_UNSUPPORTED,
                  std::string("this library is built for gfx950 only, device is ") +
                      prop.gcnArchName);
    rpt_ctx* ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return fail(RPT_E_NOMEM, "out of host memory");
    ctx->device = device;
    options_from_env(ctx->opt);
    ctx->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(RPT_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = ctx;
    return RPT_OK;
  });
}

int32_t rpt_ctx_destroy(rpt_ctx* ctx) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    if (!ctx) return RPT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
      (void)stream_sync(ctx->stream);
      prof_resolve(ctx);
      for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
      ctx->prof_events.clear();
      (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    dev_trim();
    delete ctx;
    return RPT_OK;
  });
}

int32_t rpt_ctx_sync(rpt_ctx* ctx) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx, "ctx is NULL");
    RPT_HIP(hipSetDevice(ctx->device));
    RPT_HIP(stream_sync(ctx->stream));
    return RPT_OK;
  });
}

int32_t rpt_ctx_trim(rpt_ctx* ctx) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx, "ctx is NULL");
    RPT_HIP(hipSetDevice(ctx->device));
    RPT_HIP(stream_sync(ctx->stream));
    dev_trim();
    return RPT_OK;
  });
}

int32_t rpt_ctx_stream(rpt_ctx* ctx, void** hip_stream) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && hip_stream, "NULL argument");
    *hip_stream = (void*)ctx->stream;
    return RPT_OK;
  });
}

int32_t rpt_ctx_set_option(rpt_ctx* ctx, const char* name, int64_t value) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && name, "NULL argument");
    const OptDesc* o = find_option(name);
    if (!o) return fail(RPT_E_ARG, std::string("unknown option: ") + name);
    ctx->opt.*(o->field) = value;
    return RPT_OK;
  });
}

int32_t rpt_ctx_get_option(rpt_ctx* ctx, const char* name, int64_t* value) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && name && value, "NULL argument");
    const OptDesc* o = find_option(name);
    if (!o) return fail(RPT_E_ARG, std::string("unknown option: ") + name);
    *value = ctx->opt.*(o->field);
    return RPT_OK;
  });
}

// ---- kernel timing ----------------------------------------------------------------------
static void prof_resolve(rpt_ctx* ctx) {
  for (rpt_prof_span& sp : ctx->spans) {
    float ms = 0.f;
    if (hipEventSynchronize(sp.b) == hipSuccess &&
        hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      ctx->prof_ms[sp.which] += (double)ms;
      ctx->prof_n[sp.which] += 1;
      if (sp.which == RPT_PROF_PROJECT_WIDE) {  // the wide launches are projection launches too
        ctx->prof_ms[RPT_PROF_PROJECT] += (double)ms;
        ctx->prof_n[RPT_PROF_PROJECT] += 1;
      }
    }
    if (hipEventQuery(sp.b) == hipSuccess) {  // both have completed: free for the next span
      ctx->prof_events.push_back(sp.a);
      ctx->prof_events.push_back(sp.b);
    } else {
      (void)hipGetLastError();
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
  }
  ctx->spans.clear();
}

int32_t rpt_prof_enable(rpt_ctx* ctx, int32_t on) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx, "ctx is NULL");
    ctx->prof = on != 0;
    return RPT_OK;
  });
}

int32_t rpt_prof_reset(rpt_ctx* ctx) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx, "ctx is NULL");
    RPT_HIP(hipSetDevice(ctx->device));
    prof_resolve(ctx);
    for (int i = 0; i < RPT_PROF_CLASSES; ++i) {
      ctx->prof_ms[i] = 0;
      ctx->prof_n[i] = 0;
    }
    return RPT_OK;
  });
}

int32_t rpt_prof_get(rpt_ctx* ctx, int32_t which, double* total_ms, int64_t* launches) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && total_ms && launches, "NULL argument");
    RPT_ARG(which >= 0 && which < RPT_PROF_CLASSES, "unknown kernel class");
    RPT_HIP(hipSetDevice(ctx->device));
    RPT_HIP(stream_sync(ctx->stream));
    prof_resolve(ctx);
    *total_ms = ctx->prof_ms[which];
    *launches = ctx->prof_n[which];
    return RPT_OK;
  });
}

// ---- datasets -------------------------------------------------------------------------
static int32_t check_dtype(int32_t dt) {
  RPT_ARG(dt == RPT_F64 || dt == RPT_F32 || dt == RPT_BF16, "unknown dtype");
  return RPT_OK;
}

int32_t rpt_dataset_dense_host(rpt_ctx* ctx, const void* X_host, int64_t n, int32_t d,
                               int32_t dtype, rpt_dataset** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && out, "NULL argument");
    *out = nullptr;
    RPT_TRY(check_dtype(dtype));
    RPT_ARG(n >= 0 && d >= 1, "n must be >= 0 and d >= 1");
    RPT_ARG(n < (int64_t)0x7fffffff, "n must fit int32 point ids");
    RPT_ARG(n == 0 || X_host, "X_host is NULL");
    RPT_HIP(hipSetDevice(ctx->device));
    rpt_dataset* ds = new (std::nothrow) rpt_dataset();
    if (!ds) return fail(RPT_E_NOMEM, "out of host memory");
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    ds->dtype = dtype;
    ds->owns = true;
    size_t bytes = (size_t)n * d * dtype_size(dtype);
    hipError_t e = dev_alloc(&ds->X, bytes ? bytes : 16);
    if (e != hipSuccess) {
      delete ds;
      return fail(RPT_E_NOMEM, std::string("hipMalloc dataset: ") + hipGetErrorString(e));
    }
    if (bytes) {
      e = hipMemcpyAsync(ds->X, X_host, bytes, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = stream_sync(ctx->stream);
      if (e != hipSuccess) {
        dev_free(ds->X);
        delete ds;
        return fail(RPT_E_HIP, std::string("H2D copy: ") + hipGetErrorString(e));
      }
    }
    *out = ds;
    return RPT_OK;
  });
}

int32_t rpt_dataset_dense_dev(rpt_ctx* ctx, const void* X_dev, int64_t n, int32_t d,
                              int32_t dtype, rpt_dataset** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && out, "NULL argument");
    *out = nullptr;
    RPT_TRY(check_dtype(dtype));
    RPT_ARG(n >= 0 && d >= 1, "n must be >= 0 and d >= 1");
    RPT_ARG(n < (int64_t)0x7fffffff, "n must fit int32 point ids");
    RPT_ARG(n == 0 || X_dev, "X_dev is NULL");
    RPT_ARG(((uintptr_t)X_dev & 15) == 0, "X_dev must be 16-byte aligned");
    rpt_dataset* ds = new (std::nothrow) rpt_dataset();
    if (!ds) return fail(RPT_E_NOMEM, "out of host memory");
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    ds->dtype = dtype;
    ds->owns = false;
    ds->X = const_cast<void*>(X_dev);
    *out = ds;
    return RPT_OK;
  });
}

int32_t rpt_dataset_csr_host(rpt_ctx* ctx, const int64_t* rowptr_host, const int32_t* col_host,
                             const void* val_host, int64_t n, int32_t d, int32_t dtype,
                             rpt_dataset** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && out, "NULL argument");
    *out = nullptr;
    RPT_TRY(check_dtype(dtype));
    RPT_ARG(dtype != RPT_BF16, "CSR datasets are f64 or f32");
    RPT_ARG(n >= 0 && d >= 1 && rowptr_host, "bad CSR arguments");
    RPT_ARG(n < (int64_t)0x7fffffff, "n must fit int32 point ids");
    RPT_ARG(rowptr_host[0] == 0, "rowptr[0] must be 0");
    int64_t nnz = rowptr_host[n];
    RPT_ARG(nnz >= 0, "rowptr[n] negative");
    for (int64_t i = 0; i < n; ++i)
      RPT_ARG(rowptr_host[i + 1] >= rowptr_host[i], "rowptr must be non-decreasing");
    RPT_ARG(nnz == 0 || (col_host && val_host), "col/val NULL");
    // SVector invariants (Internal.hs:99-105) are unchecked in the reference; the kernels
    // index a dense hyperplane by col, so col < d is validated here to keep HBM accesses in
    // bounds.
    for (int64_t j = 0; j < nnz; ++j)
      RPT_ARG(col_host[j] >= 0 && col_host[j] < d, "CSR column index out of range");
    RPT_HIP(hipSetDevice(ctx->device));
    rpt_dataset* ds = new (std::nothrow) rpt_dataset();
    if (!ds) return fail(RPT_E_NOMEM, "out of host memory");
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    ds->dtype = dtype;
    ds->csr = true;
    ds->owns = true;
    ds->nnz = nnz;
    size_t vb = (size_t)nnz * dtype_size(dtype);
    hipError_t e = dev_alloc((void**)&ds->rowptr, (size_t)(n + 1) * 8);
    if (e == hipSuccess) e = dev_alloc((void**)&ds->col, nnz ? (size_t)nnz * 4 : 16);
    if (e == hipSuccess) e = dev_alloc(&ds->val, vb ? vb : 16);
    if (e == hipSuccess)
      e = hipMemcpyAsync(ds->rowptr, rowptr_host, (size_t)(n + 1) * 8, hipMemcpyHostToDevice,
                         ctx->stream);
    if (e == hipSuccess && nnz)
      e = hipMemcpyAsync(ds->col, col_host, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && nnz)
      e = hipMemcpyAsync(ds->val, val_host, vb, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = stream_sync(ctx->stream);
    if (e != hipSuccess) {
      rpt_dataset_free(ds);
      return fail(RPT_E_HIP, std::string("CSR upload: ") + hipGetErrorString(e));
    }
    *out = ds;
    return RPT_OK;
  });
}

int32_t rpt_dataset_csr_dev(rpt_ctx* ctx, const int64_t* rowptr_dev, const int32_t* col_dev,
                            const void* val_dev, int64_t n, int32_t d, int32_t dtype, int64_t nnz,
                            rpt_dataset** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && out, "NULL argument");
    *out = nullptr;
    RPT_TRY(check_dtype(dtype));
    RPT_ARG(dtype != RPT_BF16, "CSR datasets are f64 or f32");
    RPT_ARG(n >= 0 && d >= 1 && nnz >= 0 && rowptr_dev, "bad CSR arguments");
    RPT_ARG(n < (int64_t)0x7fffffff, "n must fit int32 point ids");
    RPT_ARG(nnz == 0 || (col_dev && val_dev), "col/val NULL");
    // borrowed device arrays are NOT validated (that would be a pass over them): rowptr must be
    // non-decreasing from 0 to nnz and every column index in [0, d), as rpt_dataset_csr_host checks
    rpt_dataset* ds = new (std::nothrow) rpt_dataset();
    if (!ds) return fail(RPT_E_NOMEM, "out of host memory");
    ds->ctx = ctx;
    ds->n = n;
    ds->d = d;
    ds->dtype = dtype;
    ds->csr = true;
    ds->owns = false;
    ds->nnz = nnz;
    ds->rowptr = const_cast<int64_t*>(rowptr_dev);
    ds->col = const_cast<int32_t*>(col_dev);
    ds->val = const_cast<void*>(val_dev);
    *out = ds;
    return RPT_OK;
  });
}

int32_t rpt_dataset_free(rpt_dataset* ds) {
  return guarded([&]() -> int32_t {
    if (ds) dev_set_stream(ds->ctx->stream);
    if (!ds) return RPT_OK;
    if (ds->shadow32) dev_free(ds->shadow32);
    if (ds->shadow16) dev_free(ds->shadow16);
    if (ds->shadow8) dev_free(ds->shadow8);
    if (ds->shadow_col16) dev_free(ds->shadow_col16);
    if (ds->shadow_ell) dev_free(ds->shadow_ell);
    if (ds->csr_split) dev_free(ds->csr_split);
    if (ds->csr_dense) dev_free(ds->csr_dense);
    if (ds->owns) {
      (void)hipSetDevice(ds->ctx->device);
      (void)stream_sync(ds->ctx->stream);
      if (ds->X) dev_free(ds->X);
      if (ds->rowptr) dev_free(ds->rowptr);
      if (ds->col) dev_free(ds->col);
      if (ds->val) dev_free(ds->val);
    }
    delete ds;
    return RPT_OK;
  });
}

int32_t rpt_dataset_info(const rpt_dataset* ds, int64_t* n, int32_t* d, int32_t* dtype,
                         int32_t* is_csr, int64_t* nnz) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ds, "ds is NULL");
    if (n) *n = ds->n;
    if (d) *d = ds->d;
    if (dtype) *dtype = ds->dtype;
    if (is_csr) *is_csr = ds->csr ? 1 : 0;
    if (nnz) *nnz = ds->csr ? ds->nnz : ds->n * ds->d;
    return RPT_OK;
  });
}

// ---- topology -------------------------------------------------------------------------
int32_t rpt_topology(int64_t n, int32_t max_depth, int32_t min_leaf, int64_t* out,
                     int64_t cap_records, int64_t* n_records) {
  return guarded([&]() -> int32_t {
    RPT_ARG(n >= 0 && max_depth >= 0 && max_depth <= 30, "bad topology arguments");
    RPT_ARG(n_records, "n_records is NULL");
    std::vector<Node> nodes;
    enumerate_topology(n, max_depth, min_leaf, nodes);
    *n_records = (int64_t)nodes.size();
    if (out) {
      for (int64_t i = 0; i < (int64_t)nodes.size() && i < cap_records; ++i) {
        out[5 * i + 0] = nodes[i].level;
        out[5 * i + 1] = nodes[i].heap;
        out[5 * i + 2] = nodes[i].off;
        out[5 * i + 3] = nodes[i].n;
        out[5 * i + 4] = nodes[i].leaf ? 1 : 0;
      }
    }
    return RPT_OK;
  });
}

// ---- projection -----------------------------------------------------------------------
static int32_t upload_R(rpt_ctx* ctx, const double* R_host, size_t count, DevBuf<double>& buf) {
  RPT_TRY(buf.alloc(count));
  return upload_async(ctx, buf.p, R_host, count * 8);
}

int32_t rpt_project_dev(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t C,
                        int32_t mode, void* P_dev) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && ds && R_host && P_dev, "NULL argument");
    RPT_ARG(C >= 1, "C must be >= 1");
    RPT_HIP(hipSetDevice(ctx->device));
    DevBuf<double> Rd;
    RPT_TRY(upload_R(ctx, R_host, (size_t)C * ds->d, Rd));
    RPT_TRY(project_columns(ctx, ds, Rd.p, C, mode, P_dev));
    RPT_HIP(stream_sync(ctx->stream));  // Rd is released on return
    return RPT_OK;
  });
}

int32_t rpt_project_host(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t C,
                         int32_t mode, void* P_host) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && ds && R_host && P_host, "NULL argument");
    RPT_ARG(C >= 1, "C must be >= 1");
    RPT_HIP(hipSetDevice(ctx->device));
    size_t esz = dtype_size(proj_dtype(ds->dtype));
    DevBuf<char> P;
    RPT_TRY(P.alloc((size_t)C * ds->n * esz));
    RPT_TRY(rpt_project_dev(ctx, ds, R_host, C, mode, P.p));
    if (ds->n)
      RPT_HIP(hipMemcpy(P_host, P.p, (size_t)C * ds->n * esz, hipMemcpyDeviceToHost));
    return RPT_OK;
  });
}

// ---- forest ---------------------------------------------------------------------------
static int32_t forest_alloc(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                            int32_t L, int32_t min_leaf, rpt_forest** out) {
  RPT_ARG(ctx && ds && R_host && out, "NULL argument");
  RPT_ARG(T >= 1, "T must be >= 1");
  RPT_ARG(L >= 0 && L <= 30, "maxDepth must be in [0,30]");
  RPT_ARG(min_leaf >= 0, "minLeaf must be >= 0");
  RPT_ARG(ds->ctx == ctx, "dataset belongs to another context");
  RPT_ARG((int64_t)T * ds->n < ((int64_t)1 << 40), "forest too large");
  RPT_HIP(hipSetDevice(ctx->device));
  rpt_forest* f = new (std::nothrow) rpt_forest();
  if (!f) return fail(RPT_E_NOMEM, "out of host memory");
  f->ctx = ctx;
  f->n = ds->n;
  f->d = ds->d;
  f->T = T;
  f->L = L;
  f->min_leaf = min_leaf;
  f->pdtype = proj_dtype(ds->dtype);
  f->nodes = ((int64_t)1 << L) - 1;
  int32_t s = f->perm.alloc((size_t)T * f->n);
  if (s == RPT_OK) s = f->thr.alloc((size_t)T * f->nodes);
  if (s == RPT_OK) s = f->mglo.alloc((size_t)T * f->nodes);
  if (s == RPT_OK) s = f->mghi.alloc((size_t)T * f->nodes);
  if (s == RPT_OK) s = upload_R(ctx, R_host, (size_t)T * L * f->d, f->R);
  if (s != RPT_OK) {
    delete f;
    return s;
  }
  *out = f;
  return RPT_OK;
}

int32_t rpt_forest_build(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                         int32_t L, int32_t min_leaf, int32_t flags, rpt_forest** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(out, "out is NULL");
    *out = nullptr;
    rpt_forest* f = nullptr;
    RPT_TRY(forest_alloc(ctx, ds, R_host, T, L, min_leaf, &f));
    int32_t s = build_forest(ctx, ds, f, flags);
    if (s != RPT_OK) {
      delete f;
      return s;
    }
    *out = f;
    return RPT_OK;
  });
}

int32_t rpt_forest_stream_build(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                                int32_t L, int32_t min_leaf, int64_t chunk, int32_t flags,
                                rpt_forest** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(out, "out is NULL");
    *out = nullptr;
    RPT_ARG(ctx && ds && R_host, "NULL argument");
    RPT_ARG(T >= 1, "T must be >= 1");
    RPT_ARG(L >= 0 && L <= 24, "maxDepth of a streamed forest must be in [0,24]");
    RPT_ARG(min_leaf >= 0, "minLeaf must be >= 0");
    RPT_ARG(chunk >= 1, "chunk size must be >= 1");
    RPT_ARG(ds->ctx == ctx, "dataset belongs to another context");
    RPT_HIP(hipSetDevice(ctx->device));
    rpt_forest* f = new (std::nothrow) rpt_forest();
    if (!f) return fail(RPT_E_NOMEM, "out of host memory");
    f->ctx = ctx;
    f->n = ds->n;
    f->d = ds->d;
    f->T = T;
    f->L = L;
    f->min_leaf = min_leaf;
    f->pdtype = proj_dtype(ds->dtype);
    f->nodes = ((int64_t)1 << (L + 1)) - 1;
    int32_t s = f->perm.alloc((size_t)T * f->n);
    if (s == RPT_OK) s = f->thr.alloc((size_t)T * f->nodes);
    if (s == RPT_OK) s = f->mglo.alloc((size_t)T * f->nodes);
    if (s == RPT_OK) s = f->mghi.alloc((size_t)T * f->nodes);
    if (s == RPT_OK) s = upload_R(ctx, R_host, (size_t)T * L * f->d, f->R);
    if (s == RPT_OK) s = stream_build_forest(ctx, ds, f, chunk, flags);
    if (s != RPT_OK) {
      delete f;
      return s;
    }
    *out = f;
    return RPT_OK;
  });
}

int32_t rpt_forest_get_topology(rpt_forest* f, int64_t* slots, int8_t* kind_host,
                                int64_t* leaf_off_host, int64_t* leaf_len_host, int64_t* held,
                                int64_t* dropped) {
  return guarded([&]() -> int32_t {
    RPT_ARG(f, "forest is NULL");
    RPT_ARG(f->xtopo, "this forest has the implicit batch topology (rpt_topology enumerates it)");
    if (slots) *slots = f->nodes;
    if (kind_host) std::memcpy(kind_host, f->xkind_h.data(), (size_t)f->nodes);
    if (leaf_off_host) std::memcpy(leaf_off_host, f->xoff_h.data(), (size_t)f->nodes * 8);
    if (leaf_len_host) std::memcpy(leaf_len_host, f->xlen_h.data(), (size_t)f->nodes * 8);
    if (held) *held = f->held;
    if (dropped) *dropped = f->dropped;
    return RPT_OK;
  });
}

int32_t rpt_forest_import(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_host, int32_t T,
                          int32_t L, int32_t min_leaf, const int32_t* perm_host,
                          const double* thr_host, const double* mglo_host,
                          const double* mghi_host, rpt_forest** out) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(out, "out is NULL");
    *out = nullptr;
    RPT_ARG(perm_host && thr_host && mglo_host && mghi_host, "NULL argument");
    rpt_forest* f = nullptr;
    RPT_TRY(forest_alloc(ctx, ds, R_host, T, L, min_leaf, &f));
    for (int64_t i = 0; i < (int64_t)T * f->n; ++i)
      if (perm_host[i] < 0 || perm_host[i] >= f->n) {
        delete f;
        return fail(RPT_E_ARG, "perm entry out of range");
      }
    hipError_t e = hipSuccess;
    if (f->n) e = hipMemcpy(f->perm.p, perm_host, (size_t)T * f->n * 4, hipMemcpyHostToDevice);
    size_t nb = (size_t)T * f->nodes * 8;
    if (e == hipSuccess && nb) e = hipMemcpy(f->thr.p, thr_host, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess && nb) e = hipMemcpy(f->mglo.p, mglo_host, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess && nb) e = hipMemcpy(f->mghi.p, mghi_host, nb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      delete f;
      return fail(RPT_E_HIP, std::string("forest import: ") + hipGetErrorString(e));
    }
    *out = f;
    return RPT_OK;
  });
}

int32_t rpt_forest_free(rpt_forest* f) {
  return guarded([&]() -> int32_t {
    if (f) dev_set_stream(f->ctx->stream);
    if (!f) return RPT_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)stream_sync(f->ctx->stream);
    delete f;
    return RPT_OK;
  });
}

int32_t rpt_forest_info(const rpt_forest* f, int64_t* n, int32_t* d, int32_t* T, int32_t* L,
                        int32_t* min_leaf) {
  return guarded([&]() -> int32_t {
    RPT_ARG(f, "forest is NULL");
    if (n) *n = f->n;
    if (d) *d = f->d;
    if (T) *T = f->T;
    if (L) *L = f->L;
    if (min_leaf) *min_leaf = f->min_leaf;
    return RPT_OK;
  });
}

int32_t rpt_forest_get_perm(rpt_forest* f, int32_t* perm_host) {
  return guarded([&]() -> int32_t {
    if (f) dev_set_stream(f->ctx->stream);
    RPT_ARG(f && perm_host, "NULL argument");
    RPT_HIP(hipSetDevice(f->ctx->device));
    RPT_HIP(stream_sync(f->ctx->stream));
    if (f->n)
      RPT_HIP(hipMemcpy(perm_host, f->perm.p, (size_t)f->T * f->n * 4, hipMemcpyDeviceToHost));
    return RPT_OK;
  });
}

int32_t rpt_forest_get_nodes(rpt_forest* f, double* thr_host, double* mglo_host,
                             double* mghi_host) {
  return guarded([&]() -> int32_t {
    if (f) dev_set_stream(f->ctx->stream);
    RPT_ARG(f && thr_host && mglo_host && mghi_host, "NULL argument");
    RPT_HIP(hipSetDevice(f->ctx->device));
    RPT_HIP(stream_sync(f->ctx->stream));
    size_t nb = (size_t)f->T * f->nodes * 8;
    if (nb) {
      RPT_HIP(hipMemcpy(thr_host, f->thr.p, nb, hipMemcpyDeviceToHost));
      RPT_HIP(hipMemcpy(mglo_host, f->mglo.p, nb, hipMemcpyDeviceToHost));
      RPT_HIP(hipMemcpy(mghi_host, f->mghi.p, nb, hipMemcpyDeviceToHost));
    }
    return RPT_OK;
  });
}

int32_t rpt_forest_get_proj(rpt_forest* f, void* proj_host) {
  return guarded([&]() -> int32_t {
    if (f) dev_set_stream(f->ctx->stream);
    RPT_ARG(f && proj_host, "NULL argument");
    RPT_ARG(f->proj.p, "this forest holds no projections (imported forest)");
    RPT_HIP(hipSetDevice(f->ctx->device));
    RPT_HIP(stream_sync(f->ctx->stream));
    size_t nb = (size_t)f->T * f->L * f->n * dtype_size(f->pdtype);
    if (nb) RPT_HIP(hipMemcpy(proj_host, f->proj.p, nb, hipMemcpyDeviceToHost));
    return RPT_OK;
  });
}

int32_t rpt_forest_stats(rpt_forest* f, int64_t* tie_nodes, int64_t* big_mid_nodes) {
  return guarded([&]() -> int32_t {
    RPT_ARG(f, "forest is NULL");
    if (tie_nodes) *tie_nodes = f->tie_nodes;
    if (big_mid_nodes) *big_mid_nodes = f->big_mid_nodes;
    return RPT_OK;
  });
}

int32_t rpt_forest_get_mode(const rpt_forest* f, int32_t* mode) {
  return guarded([&]() -> int32_t {
    RPT_ARG(f && mode, "NULL argument");
    *mode = f->mode;
    return RPT_OK;
  });
}

int32_t rpt_forest_set_mode(rpt_forest* f, int32_t mode) {
  return guarded([&]() -> int32_t {
    RPT_ARG(f, "forest is NULL");
    RPT_ARG(mode == RPT_PROJ_AUTO || mode == RPT_PROJ_EXACT || mode == RPT_PROJ_MFMA,
            "unknown projection mode");
    f->mode = mode;
    return RPT_OK;
  });
}

int32_t rpt_split_segments(rpt_ctx* ctx, const double* key_host, int64_t n,
                           int32_t* perm_io_host, const int64_t* seg_off_host,
                           const int64_t* seg_len_host, int32_t S, double* thr_mg_host) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && key_host && perm_io_host && seg_off_host && seg_len_host && thr_mg_host,
            "NULL argument");
    RPT_ARG(n >= 1 && S >= 1, "n and S must be >= 1");
    RPT_HIP(hipSetDevice(ctx->device));
    return split_segments(ctx, key_host, n, perm_io_host, seg_off_host, seg_len_host, S,
                          thr_mg_host);
  });
}

// ---- queries --------------------------------------------------------------------------
static int32_t check_query(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* q) {
  RPT_ARG(ctx && f && q, "NULL argument");
  RPT_ARG(f->ctx == ctx && q->ctx == ctx, "handles belong to another context");
  RPT_ARG(q->d == f->d, "query dimension differs from the forest's");
  RPT_HIP(hipSetDevice(ctx->device));
  return RPT_OK;
}

int32_t rpt_candidates(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* queries,
                       int64_t* off_host, int32_t* ids_host, int64_t cap, int64_t* total) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_TRY(check_query(ctx, f, queries));
    RPT_ARG(total, "total is NULL");
    return candidates(ctx, f, queries, off_host, ids_host, cap, total);
  });
}

int32_t rpt_knnh_host(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                      const rpt_dataset* queries, int32_t k, int64_t* off_host, int32_t* ids_host,
                      double* dist_host, int64_t cap, int64_t* total) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_TRY(check_query(ctx, f, queries));
    RPT_ARG(data && data->ctx == ctx, "bad data handle");
    RPT_ARG(data->n == f->n && data->d == f->d, "data shape differs from the forest's");
    RPT_ARG(data->csr == queries->csr, "data and queries must both be dense or both CSR");
    RPT_ARG(k >= 1, "k must be >= 1");
    RPT_ARG(total, "total is NULL");
    return knn_h(ctx, f, data, queries, k, off_host, ids_host, dist_host, cap, total);
  });
}

int32_t rpt_knn_dev(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                    const rpt_dataset* queries, int32_t k, int32_t flags, int32_t* ids_dev,
                    double* dist_dev, int32_t* count_dev) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_TRY(check_query(ctx, f, queries));
    RPT_ARG(data && data->ctx == ctx, "bad data handle");
    RPT_ARG(data->n == f->n && data->d == f->d, "data shape differs from the forest's");
    RPT_ARG(data->csr == queries->csr, "data and queries must both be dense or both CSR");
    RPT_ARG(k >= 1 && k <= 1024, "k must be in [1,1024]");
    RPT_ARG(flags >= 0 && (flags & ~0x1ffff03) == 0 && (flags & 3) != 3, "unknown knn flags");
    RPT_ARG(ids_dev && dist_dev && count_dev, "NULL output");
    return knn_dev(ctx, f, data, queries, k, flags, ids_dev, dist_dev, count_dev);
  });
}

int32_t rpt_knn_host(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data,
                     const rpt_dataset* queries, int32_t k, int32_t flags, int32_t* ids_host,
                     double* dist_host, int32_t* count_host) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_TRY(check_query(ctx, f, queries));
    RPT_ARG(ids_host && dist_host && count_host, "NULL output");
    RPT_ARG(k >= 1 && k <= 1024, "k must be in [1,1024]");
    RPT_ARG(flags >= 0 && (flags & ~0x1ffff03) == 0 && (flags & 3) != 3, "unknown knn flags");
    int64_t nq = queries->n;
    DevBuf<int32_t> ids, cnt;
    DevBuf<double> dist;
    RPT_TRY(ids.alloc((size_t)nq * k));
    RPT_TRY(dist.alloc((size_t)nq * k));
    RPT_TRY(cnt.alloc((size_t)nq));
    RPT_TRY(rpt_knn_dev(ctx, f, data, queries, k, flags, ids.p, dist.p, cnt.p));
    RPT_HIP(stream_sync(ctx->stream));
    if (nq) {
      RPT_HIP(hipMemcpy(ids_host, ids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
      RPT_HIP(hipMemcpy(dist_host, dist.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
      RPT_HIP(hipMemcpy(count_host, cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    }
    return RPT_OK;
  });
}

int32_t rpt_knn_last_candidates(rpt_ctx* ctx, int64_t* total) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && total, "NULL argument");
    *total = ctx->last_candidates;
    return RPT_OK;
  });
}

int32_t rpt_knn_last_uncertified(rpt_ctx* ctx, int64_t* total) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && total, "NULL argument");
    *total = ctx->last_uncertified;
    return RPT_OK;
  });
}

int32_t rpt_knn_last_retries(rpt_ctx* ctx, int64_t* total) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && total, "NULL argument");
    *total = ctx->last_retries;
    return RPT_OK;
  });
}

int32_t rpt_build_last_handed_back(rpt_ctx* ctx, int64_t* nodes, int64_t* inconsistent) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && nodes && inconsistent, "NULL argument");
    *nodes = ctx->last_csub_redo;
    *inconsistent = ctx->last_csub_bad;
    return RPT_OK;
  });
}

int32_t rpt_knn_last_tier(rpt_ctx* ctx, int32_t* tier) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && tier, "NULL argument");
    *tier = ctx->last_tier;
    return RPT_OK;
  });
}

int32_t rpt_knn_merge_dev(rpt_ctx* ctx, const int32_t* ids_dev, const double* dist_dev,
                          const int32_t* count_dev, int32_t G, int64_t nq, int32_t k,
                          int32_t flags, int32_t* out_ids_dev, double* out_dist_dev,
                          int32_t* out_count_dev) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && ids_dev && dist_dev && count_dev && out_ids_dev && out_dist_dev &&
                out_count_dev,
            "NULL argument");
    RPT_ARG(G >= 1 && nq >= 0 && k >= 1 && k <= 1024, "bad merge arguments");
    RPT_HIP(hipSetDevice(ctx->device));
    return knn_merge_dev(ctx, ids_dev, dist_dev, count_dev, 0, G, nq, k, flags, out_ids_dev,
                         out_dist_dev, out_count_dev);
  });
}

int32_t rpt_knn_record_layout(int64_t nq, int32_t k, int64_t* bytes, int64_t* off_dist,
                              int64_t* off_ids, int64_t* off_count) {
  return guarded([&]() -> int32_t {
    RPT_ARG(bytes && off_dist && off_ids && off_count, "NULL argument");
    RPT_ARG(nq >= 0 && k >= 1, "bad record arguments");
    *off_dist = 0;
    *off_ids = nq * k * 8;
    *off_count = nq * k * 12;
    // + the shard's status word (int32 behind the counts, 0 = ok), rounded to 16 bytes
    *bytes = (nq * k * 12 + nq * 4 + 4 + 15) & ~(int64_t)15;
    return RPT_OK;
  });
}

int32_t rpt_knn_merge_records_dev(rpt_ctx* ctx, const void* records_dev, int64_t record_bytes,
                                  int32_t G, int64_t nq, int32_t k, int32_t flags,
                                  int32_t* out_ids_dev, double* out_dist_dev,
                                  int32_t* out_count_dev) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && records_dev && out_ids_dev && out_dist_dev && out_count_dev, "NULL argument");
    RPT_ARG(G >= 1 && nq >= 0 && k >= 1 && k <= 1024, "bad merge arguments");
    int64_t bytes, od, oi, oc;
    RPT_TRY(rpt_knn_record_layout(nq, k, &bytes, &od, &oi, &oc));
    RPT_ARG(record_bytes >= bytes && record_bytes % 8 == 0,
            "record_bytes smaller than rpt_knn_record_layout's size or not a multiple of 8");
    RPT_HIP(hipSetDevice(ctx->device));
    const char* base = static_cast<const char*>(records_dev);
    return knn_merge_dev(ctx, reinterpret_cast<const int32_t*>(base + oi),
                         reinterpret_cast<const double*>(base + od),
                         reinterpret_cast<const int32_t*>(base + oc), record_bytes, G, nq, k, flags,
                         out_ids_dev, out_dist_dev, out_count_dev);
  });
}

int32_t rpt_brute_knn_host(rpt_ctx* ctx, const rpt_dataset* data, const rpt_dataset* queries,
                           int32_t k, int32_t* ids_host, double* dist_host) {
  return guarded([&]() -> int32_t {
    if (ctx) dev_set_stream(ctx->stream);
    RPT_ARG(ctx && data && queries && ids_host && dist_host, "NULL argument");
    RPT_ARG(data->ctx == ctx && queries->ctx == ctx, "handles belong to another context");
    RPT_ARG(!data->csr && !queries->csr, "brute-force kNN supports dense data only");
    RPT_ARG(data->d == queries->d && data->dtype == queries->dtype, "shape/dtype mismatch");
    RPT_ARG(k >= 1 && k <= 1024, "k must be in [1,1024]");
    RPT_HIP(hipSetDevice(ctx->device));
    return brute_knn(ctx, data, queries, k, ids_host, dist_host);
  });
}

}  // extern "C"
