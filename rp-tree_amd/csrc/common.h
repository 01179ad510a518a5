// common.h — internal declarations shared by the HIP translation units of librptree_hip.so.
// gfx950 (MI355X) only: wave64, no portability layers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rptree_hip.h"

namespace rpt {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const std::string& msg);
int32_t fail(int32_t code, const std::string& msg);

#define RPT_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return ::rpt::fail(RPT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));  \
  } while (0)

#define RPT_TRY(expr)             \
  do {                            \
    int32_t s__ = (expr);         \
    if (s__ != RPT_OK) return s__; \
  } while (0)

#define RPT_ARG(cond, msg) \
  do {                     \
    if (!(cond)) return ::rpt::fail(RPT_E_ARG, msg); \
  } while (0)

// ---- topology (Internal.hs:289,495,503) ---------------------------------------------
struct Node {
  int32_t level;
  int64_t heap;
  int64_t off;
  int64_t n;
  bool leaf;
};
inline bool is_leaf(int32_t level, int64_t n, int32_t L, int32_t min_leaf) {
  return level >= L || n <= (int64_t)min_leaf;
}
void enumerate_topology(int64_t N, int32_t L, int32_t min_leaf, std::vector<Node>& out);

// device-side segment descriptor (one per node of a launch; identical for every tree)
struct Seg {
  int64_t off;    // offset of the segment inside a tree's perm row
  int32_t n;      // segment length
  int32_t heap;   // heap index of the node (-1: none)
};

inline size_t dtype_size(int32_t dt) { return dt == RPT_F64 ? 8 : dt == RPT_F32 ? 4 : 2; }
// compute/projection type: double for f64 data, float otherwise
inline int32_t proj_dtype(int32_t dt) { return dt == RPT_F64 ? RPT_F64 : RPT_F32; }

// Caching device allocator (api.hip).  Multi-GB hipMalloc/hipFree per forest build cost
// milliseconds and occasionally ~100 ms (driver map/unmap); freed blocks are therefore kept
// per device and handed out again (best fit within 25 %).  All work of a context is ordered
// on its stream and every entry point that releases buffers synchronises that stream first,
// so reuse is safe.  rpt_ctx_trim / rpt_ctx_destroy return the cache to the driver.
// A freed block becomes reusable only after the stream that was current when it was freed has
// been synchronised (stream_sync): kernels still in flight may be reading it.
hipError_t dev_alloc(void** p, size_t bytes);
void dev_free(void* p);
void dev_trim();
void dev_set_stream(hipStream_t s);          // stream of the calling thread's current entry point
hipError_t stream_sync(hipStream_t s);       // hipStreamSynchronize + release blocks freed on s

// "done once per DEVICE" flag for hipFuncSetAttribute calls: function attributes belong to the
// device's copy of the code object, and one process may drive several devices from several host
// threads (rpt_comm_init).  first(dev) is true exactly once per device (devices 0..63).
struct DeviceOnce {
  std::mutex mu;
  std::atomic<unsigned long long> done{0};
  // runs f() (an int32_t status) once per device; the device's bit is set only after f succeeded,
  // under the mutex, so a failed attempt is retried by the next call and a second host thread on
  // the same device cannot pass before the first one's call has returned
  template <class F>
  int32_t run(int dev, F&& f) {
    const unsigned long long bit = 1ULL << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (done.load(std::memory_order_relaxed) & bit) return 0;
    const int32_t s = f();
    if (s == 0) done.fetch_or(bit, std::memory_order_release);
    return s;
  }
};

// simple owned device buffer
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t count = 0;
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  void release() {
    if (p) dev_free(p);
    p = nullptr;
    count = 0;
  }
  int32_t alloc(size_t n) {
    release();
    if (n == 0) n = 1;
    hipError_t e = dev_alloc((void**)&p, n * sizeof(T));
    if (e != hipSuccess)
      return fail(RPT_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    count = n;
    return RPT_OK;
  }
  int32_t ensure(size_t n) { return (n <= count && p) ? RPT_OK : alloc(n); }
};

}  // namespace rpt

// kernel classes timed by rpt_prof_* (HIP events on the ctx stream)
enum { RPT_PROF_PROJECT = 0, RPT_PROF_SPLIT = 1, RPT_PROF_KNN_PLAN = 2, RPT_PROF_KNN_TOPK = 3,
       RPT_PROF_PROJECT_WIDE = 4, RPT_PROF_CLASSES = 5 };

struct rpt_prof_span {
  hipEvent_t a, b;
  int which;
};

// Algorithm switches of a context (rpt_ctx_set_option / rpt_ctx_get_option).  They exist for
// the parity tests of the fallback paths and for A/B timing; every default is the tuned path.
// rpt_ctx_create initialises them ONCE from the environment (RPT_<NAME>, upper case); no entry
// point reads the environment afterwards.
struct rpt_options {
  int64_t no_stream = 0;        // split: skip the streaming regime (general path from level 0)
  int64_t stream_maxnodes = 0;  // split: nodes per tree up to which levels are streamed (0 = auto)
  int64_t stream_minper = 32768;// split: points per histogram block
  int64_t no_wmid = 0;          // split: pivot bins never take the one-wave bitonic path
  int64_t no_midselect = 0;     // split: large pivot bins are sorted, not split by selection
  int64_t stream_big_node = (int64_t)1 << 21;  // split: nodes above this get > 4096 value bins
  int64_t no_wsub = 0;          // split: block-level subtree kernel instead of the wave kernel
  int64_t no_wsort = 0;         // split: the wave kernel selects by histograms (round 2) instead of sorting
  int64_t no_wpack = 0;         // split: the sorting wave kernel gathers a key per level, no packed images
  int64_t no_csub = 0;          // split: nodes of 1025 .. 8192 points take the general / LDS-subtree kernels
                                // instead of the block kernel that selects on packed 16-bit codes
  int64_t no_codes = 0;         // split: stream on the keys themselves, no 16-bit codes
  int64_t no_pcodes = 0;        // split: no codes for projection kernels without a code epilogue
  int64_t proj_narrow = 0;      // projection: 32 hyperplanes per pass only
  int64_t proj_bf16_f32 = 0;    // projection: bf16 rows through the f32-MFMA kernels
  int64_t proj_bf16_codes = 0;  // projection: bf16 rows get their codes from proj_bf16x3's epilogue, not from pcode_kernel
  int64_t proj_bf16_terms = 0;  // projection: 3 = bf16 rows against THREE bf16 terms of a hyperplane (default two: 2^-17 |x||r|);
                                // 8 = two terms on the eight-wave workgroup shape (A/B)
  int64_t proj_csr_nodense = 0; // projection: RPT_PROJ_MFMA on CSR rows stays on the segmented CSR kernel (one FMA per
                                // term) instead of the dense-ified bf16 matrix-pipe formulation
  int64_t knn_wave = -1;        // kNN: -1 auto, 0 workgroup-per-query, 1 wave-per-query kernel
  int64_t knn_kp = 0;           // kNN: entries the f32 prefilter keeps (0 = k + max(6, k/2))
  int64_t knn_no_pre32 = 0;     // kNN: no f32 prefilter (all-f64 distances)
  int64_t knn_no_pre16 = 0;     // kNN: the prefilter ranks on the f32 shadow, never on the f16 one
  int64_t knn_kp16 = 0;         // kNN: entries the f16 prefilter keeps (0 = k + max(8, k / 2))
  int64_t knn_no_pre8 = 0;      // kNN: never rank on the int8 shadow (the half shadow is the first tier)
  int64_t knn_kp8 = 0;          // kNN: entries the int8 prefilter keeps (0 = k + max(48, k), capped by the kernel variant)
  int64_t knn_csr_pre32 = 0;    // kNN: rank CSR f64 rows on their (u16 column, f32 value) shadow
  int64_t knn_general = 0;      // kNN: unfused general path
  int64_t knn_shard_old = 0;    // kNN: small shards keep the round-3 one-wave kernel (in-kernel traversal, fixed k')
  int64_t comm_force_exchange = 0;  // sharded kNN: a one-rank communicator runs record -> all-gather -> merge too
  int64_t comm_inject_failure = 0;  // sharded kNN (test hook): this device's shard reports a failure
  int64_t comm_timeout_ms = 0;      // sharded kNN: rpt_comm_sync's deadline for an exchange in flight (0 = 120 000)
  int64_t comm_stall_test = 0;      // sharded kNN (test hook): rpt_comm_sync treats a pending exchange as timed out
                                    // (the abort path on a one-GPU box)
  int64_t tune0 = 0, tune1 = 0, tune2 = 0, tune3 = 0;  // experiment hooks (0 = the built-in choice)
  int64_t debug_host = 0;       // stderr: host-side phase times of a build
  int64_t debug_stamps = 0;     // device time stamps of the wave kernel
};

struct rpt_ctx {
  rpt_options opt;
  int32_t device = 0;
  hipStream_t stream = nullptr;
  // pinned bump arena for small asynchronous host<->device transfers (api.hip: pin_alloc)
  char* pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;
  int64_t last_uncertified = 0;  // queries of the last kNN call re-run with all-f64 distances
  int64_t last_candidates = 0;
  int64_t last_retries = 0;      // queries of the last fused kNN call that took the in-kernel wider second attempt
  int32_t last_tier = 0;  // ranking tier of the last fused kNN call: 0 exact, 1 f32 shadow, 2 half, 3 int8
  // last build: nodes csub_kernel handed back to the general kernels (pivot codes shared by more
  // points than its pool), and how many of those for a histogram that contradicted the node
  // sizes (a defect: always 0)
  int64_t last_csub_redo = 0, last_csub_bad = 0;
  int32_t n_cu = 256;
  bool prof = false;
  std::vector<rpt_prof_span> spans;
  std::vector<hipEvent_t> prof_events;  // resolved spans' events, reused (creating a pair per span
                                        // cost ~5 us: 0.07 ms of a profiled C2 build)
  double prof_ms[RPT_PROF_CLASSES] = {0, 0, 0, 0, 0};
  int64_t prof_n[RPT_PROF_CLASSES] = {0, 0, 0, 0, 0};
};

namespace rpt {
// RAII span: records an event pair around the launches issued in its scope when profiling
// is enabled (rpt_prof_enable); resolved lazily by rpt_prof_get.
struct ProfScope {
  rpt_ctx* ctx;
  rpt_prof_span sp;
  bool on;
  ProfScope(rpt_ctx* c, int which) : ctx(c), on(c->prof) {
    if (!on) return;
    sp.which = which;
    auto take = [&](hipEvent_t& e) {
      if (!ctx->prof_events.empty()) {
        e = ctx->prof_events.back();
        ctx->prof_events.pop_back();
        return true;
      }
      return hipEventCreate(&e) == hipSuccess;
    };
    if (!take(sp.a)) {
      on = false;
      return;
    }
    if (!take(sp.b)) {
      ctx->prof_events.push_back(sp.a);
      on = false;
      return;
    }
    (void)hipEventRecord(sp.a, ctx->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(sp.b, ctx->stream);
    ctx->spans.push_back(sp);
  }
};
}  // namespace rpt

struct rpt_dataset {
  rpt_ctx* ctx = nullptr;
  int64_t n = 0;
  int32_t d = 0;
  int32_t dtype = RPT_F64;
  bool csr = false;
  bool owns = false;
  void* X = nullptr;  // dense: [n][d]
  int64_t* rowptr = nullptr;
  int32_t* col = nullptr;
  void* val = nullptr;
  int64_t nnz = 0;
  // lazily built by the first kNN call on dense f64 data: f32 copy of X (the fused kernel ranks
  // candidates on it before it computes exact distances of the survivors) and the largest row norm
  mutable float* shadow32 = nullptr;        // dense: X as f32; CSR: val as f32
  // round 3: X as IEEE half (dense f64 data whose elements fit its range): the first-tier ranking
  // shadow, a quarter of the f64 bytes; shadow16_state: 0 = not tried, 1 = there, -1 = unusable
  mutable uint16_t* shadow16 = nullptr;
  mutable int shadow16_state = 0;
  // int8 shadow (round 3): c = rint(x / s8_scale) clamped to [-127, 127], one global scale, rows of
  // d bytes (d % 16 == 0) — an eighth of the f64 bytes, 1 M x 128 rows fit the 256 MB Infinity
  // Cache; s8_emax = max over the rows of |x - s c| (what the certificate charges a dropped row)
  mutable int8_t* shadow8 = nullptr;
  mutable int shadow8_state = 0;  // as shadow16_state
  mutable double s8_scale = 0.0, s8_emax = 0.0;
  mutable uint16_t* shadow_col16 = nullptr;  // CSR (d <= 65536): col as u16
  mutable int64_t max_rowlen = 0;            // CSR: the longest row
  // round 3: CSR f64 rows again as a fixed-width table of (u16 column | IEEE-half value << 16)
  // slots, ell_w slots per row (the longest row rounded up to 4; absent slots are 0 = column 0,
  // value 0, which adds nothing to a distance): the first-tier ranking shadow of SVector data, 4
  // instead of 12 bytes per nonzero and no rowptr step in the walk; ell_state as shadow16_state
  mutable uint32_t* shadow_ell = nullptr;
  mutable int ell_w = 0;
  mutable int ell_state = 0;
  mutable double max_norm = -1.0;
  // lazily built by the first projection of a CSR dataset whose hyperplane tile does not fit LDS
  // whole: index of every row's first nonzero with column >= csr_split_k (project.hip)
  mutable int64_t* csr_split = nullptr;
  mutable int csr_split_k = -1;
  // round 4, tolerance-mode projection of SVector rows on the matrix pipe (project.hip
  // launch_csr_dense_mfma): the rows dense-ified as TWO bf16 terms x = x_hi + x_lo (16 significant
  // bits), [2][n][d] bf16, built by the first RPT_PROJ_MFMA projection of a CSR dataset with
  // d % 8 == 0 that fits; csr_dense_state: 0 untried, 1 built, -1 unavailable
  mutable uint16_t* csr_dense = nullptr;
  mutable int csr_dense_state = 0;
};

struct rpt_forest {
  rpt_ctx* ctx = nullptr;
  int64_t n = 0;
  int32_t d = 0, T = 0, L = 0, min_leaf = 0;
  int32_t pdtype = RPT_F64;  // type of proj
  int32_t mode = RPT_PROJ_AUTO;  // projection mode used by the build (queries reuse it)
  // set when a query batch had more than a quarter of its f32-prefilter cuts uncertified (many
  // equal distances: e.g. the queries are data points, found once per tree): later batches on this
  // forest go straight to the all-f64 kernel
  bool prefilter_off = false;
  bool pre8_off = false;   // ... and one more: the int8 shadow failed too many cuts here
  bool pre16_off = false;  // ... the same one tier up: the f16 shadow failed too many cuts here
  double kp8_boost = 1.0;  // int8 tier: factor on the kept entries, raised when a batch retried often
  int64_t nodes = 0;         // 2^L - 1
  rpt::DevBuf<int32_t> perm;  // [T][N] final leaf-ordered permutation
  rpt::DevBuf<double> thr, mglo, mghi;  // [T][nodes]
  rpt::DevBuf<char> proj;               // [T][L][N] in pdtype
  rpt::DevBuf<double> R;                // [T][L][d] hyperplanes (device copy)
  int64_t tie_nodes = 0, big_mid_nodes = 0;
  // Explicit topology (forests built by the streaming insert, rpt_forest_stream_build): the shape
  // of a streamed tree depends on the chunk sizes, so it is stored: heap slots 0 .. 2^(L+1)-2
  // (`nodes` of them), kind 0 = absent, 1 = Bin, 2 = Tip; a Tip's points are
  // perm[t][xoff[h] .. xoff[h] + xlen[h]).  The same for every tree (it is a function of
  // (N, chunk, minLeaf, maxDepth) alone), which is why it is stored once.
  bool xtopo = false;
  std::vector<int8_t> xkind_h;
  std::vector<int64_t> xoff_h, xlen_h;
  rpt::DevBuf<int8_t> xkind;
  rpt::DevBuf<int64_t> xoff, xlen;
  int64_t held = 0, dropped = 0;  // points stored per tree / lost to Internal.hs:277
};

namespace rpt {

// ---- pinned staging (api.hip) ------------------------------------------------------------
// pin_alloc: `bytes` of pinned host memory that stays valid until the next ctx_sync (or until
// the arena wraps, which synchronises the stream first).  nullptr if bytes exceeds the arena.
void* pin_alloc(rpt_ctx* ctx, size_t bytes);
// H2D copy ordered on the ctx stream that does NOT wait: the source is staged in the arena
// (falls back to a synchronous copy for sources larger than the arena).
int32_t upload_async(rpt_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
// stream_sync + recycle the arena
hipError_t ctx_sync(rpt_ctx* ctx);

// ---- projection (project.hip) -----------------------------------------------------------
// P_dev[C][n] (compute type) = R_dev[C][d] applied to every row of ds.  R_dev is a device
// copy of the dense-ified hyperplanes (double).
struct CodeOut;  // codes.h: optional 16-bit codes next to the projections
bool project_writes_codes(const rpt_ctx* ctx, const rpt_dataset* ds, int32_t mode);
// co != null: the kernels that support it also write codes (codes.h) and set *codes_written;
// the others (CSR, rows that are not 16-byte granular, bf16 on the bf16 pipe) leave it false.
int32_t project_columns(rpt_ctx* ctx, const rpt_dataset* ds, const double* R_dev, int32_t C,
                        int32_t mode, void* P_dev, const CodeOut* co = nullptr,
                        bool* codes_written = nullptr);

// ---- split / build (split.hip) ----------------------------------------------------------
int32_t build_forest(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int32_t mode);
// streaming insert of the dataset in chunks of `chunk` points (Conduit.hs:147-176 over
// Internal.hs:245-297); f->nodes = 2^(L+1)-1, node arrays and perm allocated by the caller
int32_t stream_build_forest(rpt_ctx* ctx, const rpt_dataset* ds, rpt_forest* f, int64_t chunk,
                            int32_t mode);
int32_t split_segments(rpt_ctx* ctx, const double* key_host, int64_t n, int32_t* perm_io_host,
                       const int64_t* seg_off, const int64_t* seg_len, int32_t S,
                       double* thr_mg_host);

// ---- queries (knn.hip) ------------------------------------------------------------------
int32_t candidates(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* q, int64_t* off_host,
                   int32_t* ids_host, int64_t cap, int64_t* total);
int32_t knn_dev(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data, const rpt_dataset* q,
                int32_t k, int32_t flags, int32_t* ids_dev, double* dist_dev, int32_t* count_dev);
int32_t knn_merge_dev(rpt_ctx* ctx, const int32_t* ids_dev, const double* dist_dev,
                      const int32_t* count_dev, int64_t shard_stride, int32_t G, int64_t nq,
                      int32_t k, int32_t flags, int32_t* out_ids, double* out_dist,
                      int32_t* out_count);
int32_t brute_knn(rpt_ctx* ctx, const rpt_dataset* data, const rpt_dataset* q, int32_t k,
                  int32_t* ids_host, double* dist_host);
int32_t knn_h(rpt_ctx* ctx, rpt_forest* f, const rpt_dataset* data, const rpt_dataset* q,
              int32_t k, int64_t* off_host, int32_t* ids_host, double* dist_host, int64_t cap,
              int64_t* total);

}  // namespace rpt
