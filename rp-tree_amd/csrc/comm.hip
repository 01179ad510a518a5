// comm.hip — the multi-GPU entry points of the C ABI (include/rptree_hip.h, "multi-GPU"):
// tree shards per device, the query exchange as ONE RCCL all-gather over xGMI, the k-way merge.
//
// Reference contract: the trees of a forest are independent (`createMulti` maps `create` over
// the IntMap, Internal.hs:234-240) and `knn` only concatenates the per-tree candidates in key
// order before one stable sort (RPTree.hs:174-176).  So rank r of G builds the contiguous tree
// block [r*T/G, (r+1)*T/G) with no communication, answers a query batch from its own trees, and
// the global answer is the stable merge of the G local top-k lists in rank order: top-k of a
// union is a subset of the union of the per-shard top-ks, and (distance, shard, rank in shard)
// is the reference's (distance, candidate position) order.
//
// Two ways to form the communicator, same data path:
//   rpt_comm_init(n)            one process drives n devices: ncclCommInitAll, one rpt_ctx and one
//                               worker thread per device (a build synchronises its stream once,
//                               so devices must not share a host thread)
//   rpt_comm_init_rank(ctx,..)  one process per device (launchers such as torch.distributed.run):
//                               ncclCommInitRank with an id made by rpt_comm_unique_id on rank 0
// The collective is enqueued on the ctx streams, behind the kernels that fill the exchange
// records and ahead of the merge.  The shard's query kernels synchronise their stream once (they
// read back the overflow / uncertified counters, knn.hip knn_dev); the collective, the merge and
// the status check that follow are asynchronous: rpt_comm_sync is the second and last host wait.
//
// Failure protocol (one process per GPU: the peers have no way to see a local error code).  Every
// record carries a status word behind its counts (rpt_knn_record_layout).  A rank whose query
// kernels fail STILL enqueues the all-gather, with its status word poisoned, and returns its
// error; after the merge every rank scans the gathered status words: if one is set, every count of
// the answer becomes -1 and the next rpt_comm_sync (rpt_knn_sharded calls it) fails with
// RPT_E_INTERNAL naming the rank.  The failed word accumulates (atomicMax) over all the exchanges
// since the last rpt_comm_sync, which reads and resets it: a peer's failure in ANY batch enqueued
// before the sync is reported.
// A rank that cannot even hold its record (allocation failure), whose collective cannot be
// enqueued, or that returns before the collective for any other reason aborts its OWN communicator
// (ncclCommAbort).  That alone does not release the peers: an RCCL collective whose remote rank has
// gone keeps spinning.  So rpt_comm_sync never blocks in hipStreamSynchronize while a collective
// may be in flight: it POLLS hipStreamQuery together with ncclCommGetAsyncError on every local
// communicator, under a deadline (ctx option comm_timeout_ms, default 120 000; RCCL_/NCCL_ timeouts
// are not relied on); an asynchronous RCCL error or the deadline aborts the local communicator too,
// marks it dead and returns RPT_E_INTERNAL — every rank gets out, the communicator is dead
// everywhere.  This path has run with ONE rank only (comm_force_exchange + comm_inject_failure /
// comm_stall_test on the one-GPU box); a two-process test is what is still missing.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "common.h"

using namespace rpt;

namespace {

int32_t nccl_fail(const char* what, ncclResult_t r) {
  return fail(RPT_E_HIP, std::string(what) + ": " + ncclGetErrorString(r));
}
#define RPT_NCCL(expr)                                  \
  do {                                                  \
    ncclResult_t r__ = (expr);                          \
    if (r__ != ncclSuccess) return nccl_fail(#expr, r__); \
  } while (0)

// One persistent host thread per local device (single-process mode).
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int32_t()> job;
  bool has_job = false, done = false, quit = false;
  int32_t status = RPT_OK;
  std::string err;

  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<int32_t()> j = std::move(job);
      has_job = false;
      lk.unlock();
      int32_t s;
      std::string e;
      try {
        s = j();
        if (s != RPT_OK) e = rpt_last_error();
      } catch (const std::bad_alloc&) {
        s = RPT_E_NOMEM;
        e = "out of host memory";
      } catch (...) {
        s = RPT_E_INTERNAL;
        e = "internal error in a device worker";
      }
      lk.lock();
      status = s;
      err = std::move(e);
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<int32_t()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int32_t wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return status;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

// exchange buffers of one device for one (nq, k) shape
struct Exchange {
  int64_t nq = -1;
  int32_t k = -1;
  int64_t bytes = 0, off_dist = 0, off_ids = 0, off_count = 0;
  int64_t off_status = 0;
  DevBuf<char> record;    // this shard's result (rpt_knn_record_layout)
  DevBuf<char> gathered;  // [nranks][bytes]
  DevBuf<int32_t> failed; // device: 1 + the first rank whose status word is set, 0 = none
  int32_t* failed_host = nullptr;  // pinned copy, valid after the stream is synchronised
  bool poisoned = false;  // the record's status word is currently non-zero
  bool check = false;     // an exchange ran since the last rpt_comm_sync
};

}  // namespace

struct rpt_comm {
  int32_t nranks = 0, nlocal = 0, first_rank = 0;
  bool owns_ctx = false;
  bool dead = false;  // aborted after an unrecoverable local failure (see the failure protocol)
  std::vector<rpt_ctx*> ctx;
  std::vector<ncclComm_t> comm;
  std::vector<Worker*> workers;  // nlocal > 1 only
  std::vector<std::unique_ptr<Exchange>> ex;  // one per local device
};

struct rpt_sharded_forest {
  rpt_comm* comm = nullptr;
  int32_t T = 0;
  std::vector<rpt_forest*> local;
  std::vector<int32_t> first_tree, n_trees;
};

namespace {

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

// Runs job(g) for every local device g: inline for one device, on the device workers otherwise
// (all posted, then all awaited).  Returns the first failure, its message on this thread.
int32_t run_all(rpt_comm* c, const std::function<int32_t(int)>& job) {
  if (c->nlocal == 1) return job(0);
  for (int g = 0; g < c->nlocal; ++g) c->workers[g]->post([&job, g] { return job(g); });
  int32_t first = RPT_OK;
  std::string msg;
  for (int g = 0; g < c->nlocal; ++g) {
    const int32_t s = c->workers[g]->wait();
    if (s != RPT_OK && first == RPT_OK) {
      first = s;
      msg = "device " + std::to_string(c->ctx[g]->device) + ": " + c->workers[g]->err;
    }
  }
  if (first != RPT_OK) set_error(msg);
  return first;
}

void tree_block(int32_t T, int32_t G, int32_t r, int32_t* lo, int32_t* hi) {
  *lo = (int32_t)((int64_t)r * T / G);
  *hi = (int32_t)((int64_t)(r + 1) * T / G);
}

int32_t ensure_exchange(rpt_comm* c, int g, int64_t nq, int32_t k) {
  Exchange& e = *c->ex[(size_t)g];
  if (e.nq == nq && e.k == k) return RPT_OK;
  e.nq = -1;
  RPT_TRY(rpt_knn_record_layout(nq, k, &e.bytes, &e.off_dist, &e.off_ids, &e.off_count));
  e.off_status = e.off_count + nq * 4;  // the int32 behind the counts
  RPT_TRY(e.record.alloc((size_t)e.bytes));
  RPT_TRY(e.gathered.alloc((size_t)e.bytes * c->nranks));
  if (!e.failed.p) {
    RPT_TRY(e.failed.alloc(1));
    RPT_HIP(hipMemsetAsync(e.failed.p, 0, 4, c->ctx[g]->stream));
  }
  if (!e.failed_host) {
    RPT_HIP(hipHostMalloc((void**)&e.failed_host, 64, hipHostMallocDefault));
    *e.failed_host = 0;
  }
  // the status word and the tail padding of a record are gathered too: defined bytes, once
  RPT_HIP(hipMemsetAsync(e.record.p, 0, (size_t)e.bytes, c->ctx[g]->stream));
  e.poisoned = false;
  e.nq = nq;
  e.k = k;
  return RPT_OK;
}

// after the merge: any status word set among the G gathered records -> every count of the answer
// becomes -1 and *failed = max(*failed, 1 + that rank)
__global__ void exchange_status_kernel(const char* __restrict__ gathered, int64_t record_bytes,
                                       int64_t off_status, int G, int64_t nq,
                                       int32_t* __restrict__ out_count,
                                       int32_t* __restrict__ failed) {
  int bad = 0;
  for (int g = G - 1; g >= 0; --g)
    if (*reinterpret_cast<const int32_t*>(gathered + g * record_bytes + off_status) != 0) bad = g + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && bad) atomicMax(failed, bad);  // accumulates until rpt_comm_sync resets it
  if (bad && i < nq) out_count[i] = -1;
}

// unrecoverable local failure: the peers must not wait for a collective this rank will not join
void abort_comm(rpt_comm* c) {
  for (ncclComm_t& x : c->comm)
    if (x) {
      (void)ncclCommAbort(x);
      x = nullptr;
    }
  c->dead = true;
}

void destroy_comm(rpt_comm* c) {
  if (!c) return;
  for (Worker* w : c->workers) {
    if (w) {
      w->stop();
      delete w;
    }
  }
  for (int g = 0; g < (int)c->ctx.size(); ++g) {
    if (!c->ctx[g]) continue;
    (void)hipSetDevice(c->ctx[g]->device);
    dev_set_stream(c->ctx[g]->stream);
    (void)stream_sync(c->ctx[g]->stream);
    if (g < (int)c->ex.size() && c->ex[(size_t)g]) {
      c->ex[(size_t)g]->record.release();
      c->ex[(size_t)g]->gathered.release();
      c->ex[(size_t)g]->failed.release();
      if (c->ex[(size_t)g]->failed_host) (void)hipHostFree(c->ex[(size_t)g]->failed_host);
      c->ex[(size_t)g]->failed_host = nullptr;
    }
    if (g < (int)c->comm.size() && c->comm[g]) (void)ncclCommDestroy(c->comm[g]);
  }
  if (c->owns_ctx)
    for (rpt_ctx* x : c->ctx)
      if (x) (void)rpt_ctx_destroy(x);
  delete c;
}

}  // namespace

extern "C" {

int32_t rpt_comm_init(int32_t n_gpus, rpt_comm** out) {
  return guarded([&]() -> int32_t {
    RPT_ARG(out, "out is NULL");
    *out = nullptr;
    int have = 0;
    RPT_HIP(hipGetDeviceCount(&have));
    RPT_ARG(n_gpus >= 1, "n_gpus must be >= 1");
    if (n_gpus > have)
      return fail(RPT_E_ARG, "rpt_comm_init(" + std::to_string(n_gpus) + "): only " +
                                 std::to_string(have) + " HIP device(s) visible");
    rpt_comm* c = new rpt_comm();
    c->nranks = c->nlocal = n_gpus;
    c->first_rank = 0;
    c->owns_ctx = true;
    c->ctx.assign((size_t)n_gpus, nullptr);
    c->comm.assign((size_t)n_gpus, nullptr);
    for (int g = 0; g < n_gpus; ++g) c->ex.emplace_back(new Exchange());
    for (int g = 0; g < n_gpus; ++g) {
      const int32_t s = rpt_ctx_create(g, &c->ctx[g]);
      if (s != RPT_OK) {
        destroy_comm(c);
        return s;
      }
    }
    std::vector<int> devs((size_t)n_gpus);
    for (int g = 0; g < n_gpus; ++g) devs[g] = g;
    const ncclResult_t r = ncclCommInitAll(c->comm.data(), n_gpus, devs.data());
    if (r != ncclSuccess) {
      for (ncclComm_t& x : c->comm) x = nullptr;
      destroy_comm(c);
      return nccl_fail("ncclCommInitAll", r);
    }
    if (n_gpus > 1)
      for (int g = 0; g < n_gpus; ++g) {
        Worker* w = new Worker();
        c->workers.push_back(w);
        w->th = std::thread([w] { w->loop(); });
      }
    *out = c;
    return RPT_OK;
  });
}

int32_t rpt_comm_unique_id(void* uid_out) {
  return guarded([&]() -> int32_t {
    RPT_ARG(uid_out, "uid_out is NULL");
    static_assert(sizeof(ncclUniqueId) == RPT_COMM_UID_BYTES, "RCCL unique id size");
    ncclUniqueId id;
    RPT_NCCL(ncclGetUniqueId(&id));
    std::memcpy(uid_out, &id, sizeof(id));
    return RPT_OK;
  });
}

int32_t rpt_comm_init_rank(rpt_ctx* ctx, int32_t nranks, int32_t rank, const void* uid,
                           rpt_comm** out) {
  return guarded([&]() -> int32_t {
    RPT_ARG(ctx && uid && out, "NULL argument");
    *out = nullptr;
    RPT_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "rank must be in [0, nranks)");
    RPT_HIP(hipSetDevice(ctx->device));
    rpt_comm* c = new rpt_comm();
    c->nranks = nranks;
    c->nlocal = 1;
    c->first_rank = rank;
    c->owns_ctx = false;
    c->ctx.assign(1, ctx);
    c->comm.assign(1, nullptr);
    c->ex.emplace_back(new Exchange());
    ncclUniqueId id;
    std::memcpy(&id, uid, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c->comm[0], nranks, id, rank);
    if (r != ncclSuccess) {
      c->comm[0] = nullptr;
      destroy_comm(c);
      return nccl_fail("ncclCommInitRank", r);
    }
    *out = c;
    return RPT_OK;
  });
}

int32_t rpt_comm_destroy(rpt_comm* comm) {
  return guarded([&]() -> int32_t {
    destroy_comm(comm);
    return RPT_OK;
  });
}

int32_t rpt_comm_info(const rpt_comm* comm, int32_t* nranks, int32_t* nlocal,
                      int32_t* first_rank) {
  RPT_ARG(comm, "comm is NULL");
  if (nranks) *nranks = comm->nranks;
  if (nlocal) *nlocal = comm->nlocal;
  if (first_rank) *first_rank = comm->first_rank;
  return RPT_OK;
}

int32_t rpt_comm_ctx(rpt_comm* comm, int32_t local_index, rpt_ctx** ctx) {
  RPT_ARG(comm && ctx, "NULL argument");
  RPT_ARG(local_index >= 0 && local_index < comm->nlocal, "local_index out of range");
  *ctx = comm->ctx[(size_t)local_index];
  return RPT_OK;
}

int32_t rpt_comm_sync(rpt_comm* comm) {
  return guarded([&]() -> int32_t {
    RPT_ARG(comm, "comm is NULL");
    // an exchange may be in flight: poll instead of blocking (failure protocol, top of this file)
    bool pending = false;
    for (auto& e : comm->ex) pending = pending || (e && e->check);
    if (pending && !comm->dead) {
      int64_t limit_ms = comm->ctx[0]->opt.comm_timeout_ms;
      if (limit_ms <= 0) limit_ms = 120000;
      if (comm->ctx[0]->opt.comm_stall_test) {  // test hook: the deadline has passed, whatever the streams say
        abort_comm(comm);
        return fail(RPT_E_INTERNAL, "rpt_comm_sync: the exchange did not complete within its deadline "
                                    "(comm_stall_test); communicator aborted");
      }
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<char> drained(comm->ctx.size(), 0);
      size_t left = comm->ctx.size();
      int spins = 0;
      while (left) {
        for (size_t g = 0; g < comm->ctx.size(); ++g) {
          if (drained[g]) continue;
          (void)hipSetDevice(comm->ctx[g]->device);
          const hipError_t q = hipStreamQuery(comm->ctx[g]->stream);
          if (q == hipSuccess) {
            drained[g] = 1;
            --left;
          } else if (q != hipErrorNotReady) {
            abort_comm(comm);
            return fail(RPT_E_HIP, std::string("rpt_comm_sync: ") + hipGetErrorString(q) +
                                       " (communicator aborted)");
          }
          if (g < comm->comm.size() && comm->comm[g]) {
            ncclResult_t ae = ncclSuccess;
            const ncclResult_t r = ncclCommGetAsyncError(comm->comm[g], &ae);
            if (r != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress)) {
              abort_comm(comm);
              return fail(RPT_E_INTERNAL, std::string("rpt_comm_sync: RCCL reports ") +
                                              ncclGetErrorString(r != ncclSuccess ? r : ae) +
                                              " on the exchange (a peer has gone?); communicator aborted");
            }
          }
        }
        if (!left) break;
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                            std::chrono::steady_clock::now() - t0).count();
        if (ms > limit_ms) {
          abort_comm(comm);
          return fail(RPT_E_INTERNAL, "rpt_comm_sync: the exchange did not complete within " +
                                          std::to_string(limit_ms) + " ms (a peer that never joined "
                                          "the all-gather?); communicator aborted");
        }
        // the first few thousand turns spin (a batch is a fraction of a millisecond), then back off
        if (++spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      (void)hipGetLastError();  // hipErrorNotReady of the polls must not surface at a later launch check
    }
    for (rpt_ctx* x : comm->ctx) RPT_TRY(rpt_ctx_sync(x));  // (returns at once after the poll; recycles the allocator's blocks)
    // the status words of the exchanges since the last sync (failure protocol, top of this file)
    int32_t bad = 0;
    for (size_t g = 0; g < comm->ex.size(); ++g) {
      Exchange* e = comm->ex[g].get();
      if (e && e->check) {
        e->check = false;
        if (e->failed_host && *e->failed_host) {
          if (!bad) bad = *e->failed_host;
          *e->failed_host = 0;
          (void)hipSetDevice(comm->ctx[g]->device);
          RPT_HIP(hipMemsetAsync(e->failed.p, 0, 4, comm->ctx[g]->stream));
        }
      }
    }
    if (bad)
      return fail(RPT_E_INTERNAL, "sharded kNN: rank " + std::to_string(bad - 1) +
                                      " failed to answer a batch (its own call returned the "
                                      "error); the merged answer is invalid, every count is -1");
    return RPT_OK;
  });
}

int32_t rpt_forest_build_sharded(rpt_comm* comm, const rpt_dataset* const* ds,
                                 const double* R_host, int32_t T, int32_t L, int32_t min_leaf,
                                 int32_t flags, rpt_sharded_forest** out) {
  return guarded([&]() -> int32_t {
    RPT_ARG(comm && ds && R_host && out, "NULL argument");
    *out = nullptr;
    RPT_ARG(T >= comm->nranks, "need at least one tree per rank");
    for (int g = 0; g < comm->nlocal; ++g) {
      RPT_ARG(ds[g], "dataset handle is NULL");
      RPT_ARG(ds[g]->ctx == comm->ctx[(size_t)g],
              "ds[g] must live on the communicator's g-th device (rpt_comm_ctx)");
      RPT_ARG(ds[g]->n == ds[0]->n && ds[g]->d == ds[0]->d && ds[g]->dtype == ds[0]->dtype,
              "the replicas of the point set differ in shape or type");
    }
    rpt_sharded_forest* sf = new rpt_sharded_forest();
    sf->comm = comm;
    sf->T = T;
    sf->local.assign((size_t)comm->nlocal, nullptr);
    sf->first_tree.assign((size_t)comm->nlocal, 0);
    sf->n_trees.assign((size_t)comm->nlocal, 0);
    const int64_t per_tree = (int64_t)L * ds[0]->d;
    const int32_t s = run_all(comm, [&](int g) -> int32_t {
      int32_t lo, hi;
      tree_block(T, comm->nranks, comm->first_rank + g, &lo, &hi);
      sf->first_tree[(size_t)g] = lo;
      sf->n_trees[(size_t)g] = hi - lo;
      // createMulti: every tree is built from the same points and its own L vectors
      return rpt_forest_build(comm->ctx[(size_t)g], ds[g], R_host + (int64_t)lo * per_tree, hi - lo,
                              L, min_leaf, flags, &sf->local[(size_t)g]);
    });
    if (s != RPT_OK) {
      for (rpt_forest* f : sf->local)
        if (f) (void)rpt_forest_free(f);
      delete sf;
      return s;
    }
    *out = sf;
    return RPT_OK;
  });
}

int32_t rpt_sharded_forest_free(rpt_sharded_forest* sf) {
  return guarded([&]() -> int32_t {
    if (!sf) return RPT_OK;
    for (rpt_forest* f : sf->local)
      if (f) (void)rpt_forest_free(f);
    delete sf;
    return RPT_OK;
  });
}

int32_t rpt_sharded_forest_local(rpt_sharded_forest* sf, int32_t local_index, rpt_forest** f,
                                 int32_t* first_tree, int32_t* n_trees) {
  RPT_ARG(sf, "forest is NULL");
  RPT_ARG(local_index >= 0 && local_index < (int32_t)sf->local.size(), "local_index out of range");
  if (f) *f = sf->local[(size_t)local_index];
  if (first_tree) *first_tree = sf->first_tree[(size_t)local_index];
  if (n_trees) *n_trees = sf->n_trees[(size_t)local_index];
  return RPT_OK;
}

int32_t rpt_knn_sharded_dev(rpt_comm* comm, rpt_sharded_forest* sf,
                            const rpt_dataset* const* data, const rpt_dataset* const* queries,
                            int32_t k, int32_t flags, int32_t* const* ids_dev,
                            double* const* dist_dev, int32_t* const* count_dev) {
  return guarded([&]() -> int32_t {
    RPT_ARG(comm && sf && data && queries && ids_dev && dist_dev && count_dev, "NULL argument");
    RPT_ARG(sf->comm == comm, "forest belongs to another communicator");
    RPT_ARG(k >= 1 && k <= 1024, "k must be in [1,1024]");
    RPT_ARG(flags >= 0 && flags <= 2, "unknown knn flags (voting is per device: rpt_knn_*)");
    const int64_t nq = queries[0] ? queries[0]->n : -1;
    for (int g = 0; g < comm->nlocal; ++g) {
      RPT_ARG(data[g] && queries[g] && ids_dev[g] && dist_dev[g] && count_dev[g],
              "NULL per-device argument");
      RPT_ARG(queries[g]->n == nq, "the replicas of the query batch differ in size");
    }
    if (comm->dead)
      return fail(RPT_E_INTERNAL, "the communicator was aborted after a local failure");
    if (nq == 0) return RPT_OK;
    // a one-rank communicator has nothing to exchange — unless comm_force_exchange asks for the
    // full data path (record -> ncclAllGather -> merge), which is how the exchange is exercised
    // on a one-GPU box
    const bool exchange = comm->nranks > 1 || comm->ctx[0]->opt.comm_force_exchange != 0;
    // (1) every device answers the batch from its own trees into its exchange record
    std::vector<int32_t> st((size_t)comm->nlocal, RPT_OK), have((size_t)comm->nlocal, 0);
    std::vector<std::string> msg((size_t)comm->nlocal);
    (void)run_all(comm, [&](int g) -> int32_t {
      auto body = [&]() -> int32_t {
        rpt_ctx* ctx = comm->ctx[(size_t)g];
        RPT_HIP(hipSetDevice(ctx->device));
        dev_set_stream(ctx->stream);
        RPT_TRY(ensure_exchange(comm, g, nq, k));
        have[(size_t)g] = 1;
        Exchange& e = *comm->ex[(size_t)g];
        if (e.poisoned) {
          RPT_HIP(hipMemsetAsync(e.record.p + e.off_status, 0, 4, ctx->stream));
          e.poisoned = false;
        }
        int32_t* rid = reinterpret_cast<int32_t*>(e.record.p + e.off_ids);
        double* rdist = reinterpret_cast<double*>(e.record.p + e.off_dist);
        int32_t* rcnt = reinterpret_cast<int32_t*>(e.record.p + e.off_count);
        if (ctx->opt.comm_inject_failure)  // test hook of the failure protocol
          return fail(RPT_E_INTERNAL, "comm_inject_failure: this rank pretends its query kernels failed");
        RPT_TRY(rpt_knn_dev(ctx, sf->local[(size_t)g], data[g], queries[g], k, flags, rid, rdist, rcnt));
        if (!exchange) {  // the record is the answer
          RPT_HIP(hipMemcpyAsync(ids_dev[g], rid, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, ctx->stream));
          RPT_HIP(hipMemcpyAsync(dist_dev[g], rdist, (size_t)nq * k * 8, hipMemcpyDeviceToDevice, ctx->stream));
          RPT_HIP(hipMemcpyAsync(count_dev[g], rcnt, (size_t)nq * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return RPT_OK;
      };
      st[(size_t)g] = body();
      if (st[(size_t)g] != RPT_OK) msg[(size_t)g] = rpt_last_error();
      return RPT_OK;
    });
    int32_t first = RPT_OK;
    std::string first_msg;
    for (int g = 0; g < comm->nlocal; ++g)
      if (st[(size_t)g] != RPT_OK && first == RPT_OK) {
        first = st[(size_t)g];
        first_msg = "device " + std::to_string(comm->ctx[(size_t)g]->device) + ": " + msg[(size_t)g];
      }
    if (!exchange) return first == RPT_OK ? RPT_OK : fail(first, first_msg);
    // a failed device still joins the collective, with its record's status word poisoned; one
    // that holds no record cannot: the communicator is aborted so that no peer waits for it
    for (int g = 0; g < comm->nlocal; ++g) {
      if (st[(size_t)g] == RPT_OK) continue;
      bool ok = have[(size_t)g] != 0;
      if (ok) {
        Exchange& e = *comm->ex[(size_t)g];
        (void)hipSetDevice(comm->ctx[(size_t)g]->device);
        (void)hipGetLastError();
        ok = hipMemsetAsync(e.record.p + e.off_status, 0xff, 4, comm->ctx[(size_t)g]->stream) == hipSuccess;
        e.poisoned = true;
      }
      if (!ok) {
        abort_comm(comm);
        return fail(first, first_msg + " (no exchange record: communicator aborted)");
      }
    }
    // (2) ONE all-gather of the records, enqueued behind the kernels on every ctx stream
    ncclResult_t nr = ncclSuccess;
    if (comm->nlocal > 1) nr = ncclGroupStart();
    for (int g = 0; g < comm->nlocal && nr == ncclSuccess; ++g) {
      Exchange& e = *comm->ex[(size_t)g];
      nr = ncclAllGather(e.record.p, e.gathered.p, (size_t)e.bytes, ncclUint8,
                         comm->comm[(size_t)g], comm->ctx[(size_t)g]->stream);
    }
    if (comm->nlocal > 1) {
      const ncclResult_t ge = ncclGroupEnd();
      if (nr == ncclSuccess) nr = ge;
    }
    if (nr != ncclSuccess) {  // some devices may have joined, others not: nothing to salvage
      abort_comm(comm);
      return nccl_fail("ncclAllGather (communicator aborted)", nr);
    }
    // (3) k-way merge in (distance, shard, rank) order on every device, behind the collective,
    // then the scan of the gathered status words
    for (int g = 0; g < comm->nlocal; ++g) {
      Exchange& e = *comm->ex[(size_t)g];
      rpt_ctx* ctx = comm->ctx[(size_t)g];
      RPT_TRY(rpt_knn_merge_records_dev(ctx, e.gathered.p, e.bytes, comm->nranks, nq, k, flags,
                                        ids_dev[g], dist_dev[g], count_dev[g]));
      RPT_HIP(hipSetDevice(ctx->device));
      hipLaunchKernelGGL(exchange_status_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0,
                         ctx->stream, e.gathered.p, e.bytes, e.off_status, comm->nranks, nq,
                         count_dev[g], e.failed.p);
      RPT_HIP(hipGetLastError());
      RPT_HIP(hipMemcpyAsync(e.failed_host, e.failed.p, 4, hipMemcpyDeviceToHost, ctx->stream));
      e.check = true;
    }
    return first == RPT_OK ? RPT_OK : fail(first, first_msg);
  });
}

int32_t rpt_knn_sharded(rpt_comm* comm, rpt_sharded_forest* sf, const rpt_dataset* const* data,
                        const rpt_dataset* const* queries, int32_t k, int32_t flags,
                        int32_t* ids_host, double* dist_host, int32_t* count_host) {
  return guarded([&]() -> int32_t {
    RPT_ARG(comm && queries && queries[0] && ids_host && dist_host && count_host, "NULL argument");
    RPT_ARG(k >= 1 && k <= 1024, "k must be in [1,1024]");
    const int64_t nq = queries[0]->n;
    const int G = comm->nlocal;
    std::vector<DevBuf<int32_t>> ids((size_t)G), cnt((size_t)G);
    std::vector<DevBuf<double>> dist((size_t)G);
    std::vector<int32_t*> pi((size_t)G), pc((size_t)G);
    std::vector<double*> pd((size_t)G);
    for (int g = 0; g < G; ++g) {
      RPT_HIP(hipSetDevice(comm->ctx[(size_t)g]->device));
      dev_set_stream(comm->ctx[(size_t)g]->stream);
      RPT_TRY(ids[(size_t)g].alloc((size_t)nq * k));
      RPT_TRY(dist[(size_t)g].alloc((size_t)nq * k));
      RPT_TRY(cnt[(size_t)g].alloc((size_t)nq));
      pi[(size_t)g] = ids[(size_t)g].p;
      pd[(size_t)g] = dist[(size_t)g].p;
      pc[(size_t)g] = cnt[(size_t)g].p;
    }
    int32_t s = rpt_knn_sharded_dev(comm, sf, data, queries, k, flags, pi.data(), pd.data(), pc.data());
    if (s == RPT_OK) {
      s = rpt_comm_sync(comm);  // also reports a PEER's failure (status words of the records)
    } else if (!comm->dead) {   // the collective this rank joined with a poisoned record drains
      const std::string keep = rpt_last_error();
      (void)rpt_comm_sync(comm);
      set_error(keep);
    }
    if (s == RPT_OK && nq) {  // every device holds the same merged answer: read device 0's
      RPT_HIP(hipSetDevice(comm->ctx[0]->device));
      RPT_HIP(hipMemcpy(ids_host, pi[0], (size_t)nq * k * 4, hipMemcpyDeviceToHost));
      RPT_HIP(hipMemcpy(dist_host, pd[0], (size_t)nq * k * 8, hipMemcpyDeviceToHost));
      RPT_HIP(hipMemcpy(count_host, pc[0], (size_t)nq * 4, hipMemcpyDeviceToHost));
    }
    for (int g = 0; g < G; ++g) {  // buffers go back to the allocator of THEIR device's stream
      (void)hipSetDevice(comm->ctx[(size_t)g]->device);
      dev_set_stream(comm->ctx[(size_t)g]->stream);
      ids[(size_t)g].release();
      dist[(size_t)g].release();
      cnt[(size_t)g].release();
    }
    return s;
  });
}

}  // extern "C"
