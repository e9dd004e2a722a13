// C-ABI layer of libcora_hip.so (see include/cora_hip.h).  Owns the handle,
// device memory and stream; every compute entry point ends in a HIP kernel of
// kernels.hip -- there is no CPU fallback.
//
// ONE translation unit in ten pieces (round 6: the file had grown to 3 400 lines).  This file holds the handle (cora_ctx), the
// error / device macros and the helpers every part uses; the entry points live in capi/*.inc, included at the end in this order:
//   handle.inc          creation of (partitioned) handles, destruction, rank state, row maps, statistics
//   resident.inc        device vectors, the current point, the trust-region trial / accept pair, products, projections
//   preconditioner.inc  Jacobi / Cholesky set-up, a host factor installed as a device solve plan, the staged solve
//   products.inc        implicit formulation, exchange of a partitioned product and its overlap, distributed long rows
//   solver_ops.inc      auxiliary factors, formulation switch, retraction, vector updates, inner products
//   stpcg.inc           the device-resident Steihaug-Toint PCG (every form of the iteration), injected communication
//   blocks.inc          row moves, STPCG measurement hooks, LOBPCG's block algebra, timers
//   host_pointer.inc    the host-pointer operator API (one entry per reference method), host-side debug hooks
//   comm.inc            native communication: RCCL, in-process and device-side (p2p.h) transports
#include <hip/hip_runtime.h>
// RCCL's types and the few enumerators used, declared here (NCCL's public ABI: they have not changed since 2.0): the
// entry points are resolved with dlsym at run time (native communication, end of file), so neither the build nor a
// single-GPU user needs RCCL's headers or library.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <future>
#include <condition_variable>
#include <map>
#include <mutex>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <deque>
#include <atomic>
#include <vector>

#include "../../include/cora_hip.h"
#include "cora_internal.h"
#include "kernels.h"
#include "p2p.h"
#include "parallel.h"

using namespace cora;

namespace {
thread_local std::string g_create_error;
constexpr int kScratchSlots = 9;
}  // namespace

constexpr size_t kProfMarks = 8;  // events per profiled STPCG iteration (cora_debug_profile_stpcg)
struct cora_native_comm;
static void native_comm_destroy(cora_native_comm *nc);
static double *native_scalars(cora_native_comm *nc);                       // 8 device doubles of the sharded STPCG
static int native_allreduce_dev(cora_native_comm *nc, double *d, int n);   // sum over the ranks, in place, on the stream
static int native_exchange_on(cora_native_comm *nc, double *dX, int ld, hipStream_t st);  // the exchange, ordered on st
// the exchange of a PRODUCT, in two halves around the launch of the distributed long rows' chunks: pack the exported rows
// (+ zeroed slots for the long rows' partial sums, returned in *slots) | all-gather of rows and slots in ONE collective,
// rows scattered into dX, slots summed in rank order into the owners' rows of `out` (+ their kappa slots)
static int native_product_pack(cora_native_comm *nc, const double *dX, int ld, hipStream_t st, double **slots);
static int native_product_gather(cora_native_comm *nc, double *dX, int ld, hipStream_t st, double *out, double *kappa,
                                 hipEvent_t after_collective = nullptr);
static const std::string &native_error(const cora_native_comm *nc);
static int native_allgather_rows(cora_native_comm *nc, double *dX, int ld, int64_t row0, int64_t nrows);  // one piece of every shard, packed
struct cora_ctx {
  HostFormat F;
  cora_native_comm *native_comm = nullptr;  // owned: the library's own communication (cora_comm_create_*)
  cora::P2PState *p2p_pending = nullptr;    // a mailbox exported by cora_comm_p2p_handle, not yet connected
  int device = -1;
  bool has_device = false;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int p = 0, ld = 0;

  SliceDesc *d_slices = nullptr;
  SliceDesc *d_slices_pf = nullptr;  // HostFormat::slices_pose_first (empty: nullptr)
  // partitioned handles: the slices that read only rows of this rank's own shard ("interior") and the ones that read a
  // row another rank owns ("boundary"), both in chain order.  With the library's own communication the interior slices
  // run while the exchange of the operand is still under way on comm_stream (exchange_and_product)
  SliceDesc *d_slices_int = nullptr, *d_slices_bnd = nullptr;
  int n_slices_int = 0, n_slices_bnd = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_operand = nullptr, ev_exchanged = nullptr;
  int overlap_exchange = 1;  // cora_comm_overlap_enable: 0 never, 1 when the interior part is worth a launch of its own, 2 always
  double *d_sval = nullptr;
  double *d_head_val = nullptr;  // HostFormat::head_val
  int32_t *d_scol = nullptr;
  int32_t *d_perm = nullptr;
  LongChunk *d_chunks = nullptr;
  int32_t *d_chunk_order = nullptr;
  double *d_lval = nullptr;
  int32_t *d_lcol = nullptr;
  double *d_partials = nullptr;
  unsigned *d_tickets = nullptr;
  int32_t *d_api2int = nullptr;
  double *d_diag_inv = nullptr;  // 1/diag(Q), local rows
  double *d_lam_st = nullptr, *d_lam_ob = nullptr;
  // partitioned handles: distributed long rows (HostFormat::long_rows): partial-sum slots, rows, owners
  double *d_long_out = nullptr;
  int32_t *d_long_rows = nullptr, *d_long_owner = nullptr;

  // sparse Cholesky factors resident on the device (level-scheduled triangular solves):
  // the preconditioner's (Q + lambda I)[0:m] and, for the translation-implicit formulation,
  // the translation Laplacian Q33[0:nt-1]
  struct DevStage {
    RowOpDev fwd_a{}, fwd_b{}, bwd_a{}, bwd_b{};
    BlockOpDev blocks{};
    SubOpDev sub{};
    bool has_fwd_a = false, has_bwd_a = false, dense = false, is_sub = false, aux_sum = false;
  };
  struct DevFactor {
    TriPlan plan;  // host copy is dropped after upload (only the counts are kept)
    std::vector<DevStage> stages;
    // the plan's arrays live in a few large device chunks handed out front to back (60 arrays per factor: one hipMalloc
    // / hipFree each cost more than the copies); a re-installed factor writes over the chunks of the one before
    std::vector<void *> allocs;
    std::vector<size_t> chunk_bytes;
    size_t chunk_at = 0, chunk_used = 0;
    int aux_rows = 0;  // two-stage plans: rows appended to the work vector
    bool fuse_ok = false;  // substitution blocks whose tiles hold every pose's rotation rows at consecutive positions:
                           // the STPCG passes can be fused into the sweeps (SubFuse, kernels.h)
    bool ready = false;
    int64_t entries[6] = {0, 0, 0, 0, 0, 0};  // cora_precond_entries (counted at install, before the host copy is dropped)
    unsigned long long generation = 0;  // counts installs: a captured STPCG graph carries the plan's arrays and sizes
  };
  DevFactor precond_f, implicit_f, aux_f;  // aux_f: the caller's own factor (cora_aux_set_cholesky)
  bool implicit = false;  // Formulation::Implicit active

  bool have_point = false;
  double *d_Y = nullptr, *d_G = nullptr, *d_rgrad = nullptr;
  // cora_tnt_trial_dev leaves Q X of its trial point here; cora_tnt_accept_dev of the same point takes it as the new
  // point's Euclidean gradient (the two buffers change places) instead of forming the product again
  double *d_G_trial = nullptr;
  const double *trial_x = nullptr;  // the trial point d_G_trial belongs to (nullptr: none)
  unsigned long long stpcg_pending_seq = 0;  // a neutral iteration of the last inner solve may still be in flight (stpcg_run)
  double f = 0.0;
  int precond = CORA_PRECOND_NONE;

  double *scratch[kScratchSlots] = {nullptr};
  size_t scratch_bytes[kScratchSlots] = {0};
  double *d_stage = nullptr;
  size_t stage_bytes = 0;
  std::future<void> deferred_free;  // a solve plan's host arrays being freed (install_factor)
  // two pinned buffers of kPinChunk bytes and their events: big downloads are pipelined through them (DMA into one
  // while the host copies the other out) instead of hipMemcpy's own staging of pageable memory
  char *h_pin[2] = {nullptr, nullptr};
  hipEvent_t ev_pin[2] = {nullptr, nullptr};
  double *d_red = nullptr;      // reduction partials
  size_t red_doubles = 0;
  double *d_scalars = nullptr;  // 8 doubles
  StpcgState *d_stpcg = nullptr, *h_stpcg = nullptr;  // device-resident STPCG scalars and their pinned mirror
  unsigned long long dot_seq = 0;  // h_scalars[7] carries the sequence number of the last finished reduction
  unsigned *d_ticket = nullptr;  // last-block ticket of the inner-product kernels (zero between launches)
  double *h_scalars = nullptr;  // pinned, 8 doubles
  double *h_gram = nullptr;     // pinned, 16 x 24 x 24 doubles: where the Gram products' reduction writes its results
  int *d_flag = nullptr;
  int *h_flag = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // measurement hook (cora_debug_profile_stpcg): event pairs around the Hessian-vector product of every
  // device-resident STPCG iteration
  // injected communication of a partitioned handle (cora_set_comm)
  cora_exchange_fn comm_exchange = nullptr;
  cora_allreduce_fn comm_allreduce = nullptr;
  cora_allgather_fn comm_allgather = nullptr;
  void *comm_user = nullptr;
  bool comm_required = false;  // cora_require_comm: a collective step without communication is an error, not a no-op
  bool local_products = false;  // cora_debug_local_products: products skip every collective step (kernel timing only)
  // cora_debug_product_phases: events at the phase boundaries of the one-collective product (serial order)
  std::vector<hipEvent_t> phase_events;
  bool phase_timing = false;
  int prof_stpcg = 0;  // 0 off | 1 events around the product of every STPCG iteration | 2 around every launch of it
  std::vector<hipEvent_t> prof_events;  // kProfMarks per iteration
  double prof_hvp_us = 0.0;
  int prof_hvp_count = 0;
  double prof_phase_us[7] = {0, 0, 0, 0, 0, 0, 0};  // mean time between marks k and k + 1 (-1: not recorded); [6]: two marks in a row
  bool prof_kappa_folded = false;  // the profiled iterations had no kappa launch (SubFuse::n_kappa)
  int stpcg_path = 0;  // iteration form of the last cora_stpcg_dev: 0 unfused, 1 fused vector passes, 2 sweep-fused
  // A batch of device-resident STPCG iterations as a hipGraph: the launches of an iteration have the same arguments
  // every time (the scalars live in device memory, the sequence number the host waits for is a device counter), so a
  // batch is captured once and replayed -- replayed launches follow each other 1.3 us closer than launches enqueued
  // one by one (tools/launch_lab.hip) and cost the host one call instead of twenty.  Kept while its key -- every
  // pointer and size the captured launches carry -- stays the same, i.e. normally for a whole TNT call and beyond.
  hipGraphExec_t stpcg_graph = nullptr;
  std::vector<uintptr_t> stpcg_graph_key;
  unsigned long long *d_seq_counter = nullptr;  // the device's copy of dot_seq (kernels.h, DotArgs::seq_counter)
  long stpcg_graph_replays = 0, stpcg_graph_captures = 0;
  std::vector<std::pair<double *, size_t>> user_allocs;  // live vectors of cora_dev_alloc (pointer, bytes)
  std::vector<std::pair<double *, size_t>> pool;         // released ones, kept for the next request of the same size
  std::string err;
};

struct cora_ctx;
static int apply_product(cora_ctx *c, const double *dX, int ld, int epi, double *dOut);  // formulation-aware
static int launch_product(cora_ctx *c, SpmmArgs A, int ld, int epi, bool finish = true);  // one SpMM launch (+ the distributed long rows)
static int finish_long_rows(cora_ctx *c, const SpmmArgs &A, int ld, int epi);
static int exchange_and_product(cora_ctx *c, SpmmArgs A, int ld, int epi);  // exchange of the operand's remote rows + the product
static int product_kappa_slots(const cora_ctx *c, const SpmmArgs &A);

namespace {

int fail(cora_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIP_TRY(c, expr)                                                              \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess)                                                            \
      return fail((c), CORA_ERR_HIP,                                                  \
                  std::string(#expr) + ": " + hipGetErrorString(e__));                \
  } while (0)

#define NEED_DEVICE(c)                                                                \
  do {                                                                                \
    if (!(c)) return CORA_ERR_ARG;                                                    \
    if (!(c)->has_device)                                                             \
      return fail((c), CORA_ERR_HIP, "no HIP device bound to this handle (plan-only)"); \
    HIP_TRY((c), hipSetDevice((c)->device));                                          \
  } while (0)

#define NEED_RANK(c)                                                                  \
  do {                                                                                \
    if ((c)->p <= 0) return fail((c), CORA_ERR_NOT_READY, "cora_set_rank not called"); \
  } while (0)

// A kept trial product (cora_tnt_trial_dev) belongs to the CONTENTS of a vector, not to its address: every entry point that
// can write a caller's vector, or hand its address out again, says so here (round-5 advice: an accept after such a write
// silently took a stale Euclidean gradient).
inline void wrote(cora_ctx *c, const void *p) {
  if (c && p && p == c->trial_x) c->trial_x = nullptr;
}

template <typename T>
hipError_t to_device(T **dptr, const std::vector<T> &v) {
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  hipError_t e = hipMalloc(reinterpret_cast<void **>(dptr), bytes);
  if (e != hipSuccess) return e;
  if (!v.empty()) e = hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

size_t vec_bytes(const cora_ctx *c, int ld) {
  return static_cast<size_t>(c->F.L.rows) * ld * sizeof(double);
}

int get_scratch(cora_ctx *c, int slot, int ld, double **out, int64_t extra_rows = 0) {
  const size_t need = vec_bytes(c, ld) + static_cast<size_t>(extra_rows) * ld * sizeof(double);
  if (c->scratch_bytes[slot] < need) {
    if (c->scratch[slot]) (void)hipFree(c->scratch[slot]);
    c->scratch[slot] = nullptr;
    c->scratch_bytes[slot] = 0;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->scratch[slot]), need));
    c->scratch_bytes[slot] = need;
  }
  *out = c->scratch[slot];
  return CORA_OK;
}

// collective steps of a partitioned handle; no-ops on a single-GPU one
// (a partitioned handle WITHOUT communication keeps the round-1 contract: the caller keeps the rows the products
// read current and adds up the per-rank partial results itself)
int comm_missing(cora_ctx *c) {
  if (!c->comm_required) return CORA_OK;
  return fail(c, CORA_ERR_NOT_READY,
              "partitioned handle without communication: install it with cora_set_comm or cora_comm_create_* (again after "
              "every rebuild of the handle, e.g. Problem::updateProblemData or a second setPartition)");
}
int comm_exchange(cora_ctx *c, const double *dX, int ld) {
  if (c->F.L.world == 1) return CORA_OK;
  if (!c->comm_exchange) return comm_missing(c);
  if (c->comm_exchange(c->comm_user, const_cast<double *>(dX), ld)) return fail(c, CORA_ERR_HIP, "exchange step failed");
  return CORA_OK;
}
int comm_allreduce(cora_ctx *c, double *vals, int n) {
  if (c->F.L.world == 1) return CORA_OK;
  if (!c->comm_allreduce) return comm_missing(c);
  if (c->comm_allreduce(c->comm_user, vals, n)) return fail(c, CORA_ERR_HIP, "all-reduce step failed");
  return CORA_OK;
}
int comm_allgather(cora_ctx *c, const double *dX, int ld) {
  if (c->F.L.world == 1) return CORA_OK;
  if (!c->comm_allgather) return comm_missing(c);
  if (c->comm_allgather(c->comm_user, const_cast<double *>(dX), ld)) return fail(c, CORA_ERR_HIP, "all-gather step failed");
  return CORA_OK;
}

RowArgs row_args(const cora_ctx *c) {
  const Layout &L = c->F.L;
  RowArgs R;
  R.d = L.d;
  R.nl_poses = L.nl_poses;
  R.nl_ranges = L.nl_ranges;
  R.nl_trans = L.nl_trans;
  R.rot_base = static_cast<size_t>(L.rot_base);
  R.rng_base = static_cast<size_t>(L.rng_base);
  R.trn_base = static_cast<size_t>(L.trn_base);
  R.base = static_cast<size_t>(L.base);
  return R;
}

SpmmArgs spmm_args(const cora_ctx *c, const double *X, double *out) {
  SpmmArgs A;
  A.slices = c->d_slices;
  A.slices_pose_first = c->d_slices_pf;
  A.n_slices = static_cast<int>(c->F.slices.size());
  A.n_chunks = static_cast<int>(c->F.chunks.size());
  A.sval = c->d_sval;
  A.head_val = c->d_head_val;
  A.scol = c->d_scol;
  A.perm = c->d_perm;
  A.chunks = c->d_chunks;
  A.chunk_order = c->d_chunk_order;
  A.lval = c->d_lval;
  A.lcol = c->d_lcol;
  A.partials = c->d_partials;
  A.tickets = c->d_tickets;
  A.n_long_rows = c->F.n_long_rows;
  A.X = X;
  A.out = out;
  A.Y = c->d_Y;
  A.lam_st = c->d_lam_st;
  A.lam_ob = c->d_lam_ob;
  const Layout &L = c->F.L;
  A.win_rot_lo = static_cast<int32_t>(L.rot_base);
  A.win_rot_hi = static_cast<int32_t>(L.rot_base + static_cast<int64_t>(L.nl_poses) * L.d);
  A.win_trn_lo = static_cast<int32_t>(L.trn_base);
  A.win_trn_hi = static_cast<int32_t>(L.trn_base + L.nl_trans);
  A.n_local_poses = L.nl_poses;
  return A;
}

int ensure_red(cora_ctx *c, size_t doubles) {
  if (c->red_doubles < doubles) {
    if (c->d_red) (void)hipFree(c->d_red);
    c->d_red = nullptr;
    c->red_doubles = 0;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_red), doubles * sizeof(double)));
    c->red_doubles = doubles;
  }
  return CORA_OK;
}

// host col-major (N x k, ld) -> resident vector (rows x ld_for(k)), zero padded
int upload_impl(cora_ctx *c, const double *host, int ldh, int k, double *dptr) {
  const int64_t N = c->F.L.N;
  if (k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_SHAPE, "column count must be in [1, 24]");
  if (ldh < N) return fail(c, CORA_ERR_SHAPE, "leading dimension smaller than N");
  if (!host || !dptr) return fail(c, CORA_ERR_ARG, "null pointer");
  const int ld = ld_for(k);
  const size_t need = static_cast<size_t>(N) * k * sizeof(double);
  if (c->stage_bytes < need) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_stage), need));
    c->stage_bytes = need;
  }
  HIP_TRY(c, hipMemcpy2DAsync(c->d_stage, N * sizeof(double), host, static_cast<size_t>(ldh) * sizeof(double),
                              N * sizeof(double), k, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(dptr, 0, vec_bytes(c, ld), c->stream));
  HIP_TRY(c, launch_upload(N, k, ld, c->d_stage, c->d_api2int, dptr, c->stream));
  return CORA_OK;
}

int download_impl(cora_ctx *c, const double *dptr, int k, double *host, int ldh) {
  const int64_t N = c->F.L.N;
  if (k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_SHAPE, "column count must be in [1, 24]");
  if (ldh < N) return fail(c, CORA_ERR_SHAPE, "leading dimension smaller than N");
  if (!host || !dptr) return fail(c, CORA_ERR_ARG, "null pointer");
  const int ld = ld_for(k);
  const size_t need = static_cast<size_t>(N) * k * sizeof(double);
  if (c->stage_bytes < need) {
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_stage), need));
    c->stage_bytes = need;
  }
  {
    const int rc = comm_allgather(c, dptr, ld);
    if (rc) return rc;
  }
  HIP_TRY(c, launch_download(N, k, ld, dptr, c->d_api2int, c->d_stage, c->stream));
  constexpr size_t kPinChunk = size_t(8) << 20;
  if (ldh == N && need >= 4 * kPinChunk) {
    // One flat block of `need` bytes.  hipMemcpy into pageable memory stages through the runtime's own bounce buffer
    // (measured 4.4 GB/s: 0.1 s for the 430 MB Ritz block of a 10^6-pose certification); here the DMA engine fills one
    // pinned chunk while host threads copy the previous one out.
    for (int i = 0; i < 2; ++i) {
      if (!c->h_pin[i]) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_pin[i]), kPinChunk));
      if (!c->ev_pin[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pin[i], hipEventDisableTiming));
    }
    const char *src = reinterpret_cast<const char *>(c->d_stage);
    char *dst = reinterpret_cast<char *>(host);
    const size_t nchunks = (need + kPinChunk - 1) / kPinChunk;
    auto bytes_of = [&](size_t ch) { return std::min(kPinChunk, need - ch * kPinChunk); };
    auto start = [&](size_t ch) -> hipError_t {
      const hipError_t e = hipMemcpyAsync(c->h_pin[ch & 1], src + ch * kPinChunk, bytes_of(ch), hipMemcpyDeviceToHost, c->stream);
      return e != hipSuccess ? e : hipEventRecord(c->ev_pin[ch & 1], c->stream);
    };
    HIP_TRY(c, start(0));
    const unsigned nth = std::min(4u, std::max(1u, std::thread::hardware_concurrency()));
    for (size_t ch = 0; ch < nchunks; ++ch) {
      HIP_TRY(c, hipEventSynchronize(c->ev_pin[ch & 1]));
      if (ch + 1 < nchunks) HIP_TRY(c, start(ch + 1));  // (the other buffer: its host copy finished in the round before)
      const size_t nb = bytes_of(ch);
      const char *from = c->h_pin[ch & 1];
      char *to = dst + ch * kPinChunk;
      cora::parallel_parts(nth, [&](unsigned t) {
        const size_t a = nb * t / nth, b = nb * (t + 1) / nth;
        std::memcpy(to + a, from + a, b - a);
      });
    }
    return CORA_OK;
  }
  HIP_TRY(c, hipMemcpy2DAsync(host, static_cast<size_t>(ldh) * sizeof(double), c->d_stage, N * sizeof(double),
                              N * sizeof(double), k, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CORA_OK;
}

// Lambda, grad and f from (Y, G) already resident in d_Y / d_G.
// wait == false (one GPU): nothing is waited for -- f arrives in h_scalars[4] (pinned) before whatever the caller enqueues
// next on the stream finishes, and the caller stores it in c->f after its own wait.
int point_finish(cora_ctx *c, bool wait = true) {
  const RowArgs R = row_args(c);
  const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
  const int nb = static_cast<int>((units + 255) / 256);
  int rc = ensure_red(c, static_cast<size_t>(std::max(nb, 1)) * 4);
  if (rc) return rc;
  int nblocks = 0;
  HIP_TRY(c, launch_point_finish(R, c->ld, c->d_Y, c->d_G, c->d_rgrad, c->d_lam_st, c->d_lam_ob, c->d_red,
                                 &nblocks, c->stream));
  if (!wait) {
    c->h_scalars[4] = 0.0;
    if (nblocks > 0) HIP_TRY(c, launch_reduce_partials(c->d_red, nblocks, 1, c->h_scalars + 4, c->stream));
    c->have_point = true;
    return CORA_OK;
  }
  if (nblocks > 0) {
    HIP_TRY(c, launch_reduce_partials(c->d_red, nblocks, 1, c->h_scalars, c->stream));  // pinned host memory
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->f = c->h_scalars[0];
  } else {
    c->f = 0.0;
  }
  if ((rc = comm_allreduce(c, &c->f, 1))) return rc;
  c->have_point = true;
  return CORA_OK;
}

// Wait for the inner-product kernel that was just launched: its last block writes the results and then
// the sequence number into pinned memory, which the host polls -- a few microseconds less than a stream
// synchronisation, twice per STPCG iteration.  Falls back to the stream if the number never arrives.
int wait_dots(cora_ctx *c, unsigned long long seq) {
  volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(c->h_scalars + 7);
  for (long spin = 0; spin < 20000000L; ++spin)
    if (*flag >= seq) return CORA_OK;  // (numbers only grow on a handle; a later reduction may already have finished)
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return *flag >= seq ? CORA_OK : fail(c, CORA_ERR_HIP, "inner-product kernel did not complete");
}

int set_point_dev_impl(cora_ctx *c, const double *dY) {
  c->trial_x = nullptr;
  if (dY != c->d_Y)
    HIP_TRY(c, hipMemcpyAsync(c->d_Y, dY, vec_bytes(c, c->ld), hipMemcpyDeviceToDevice, c->stream));
  const int rc = apply_product(c, c->d_Y, c->ld, EPI_NONE, c->d_G);
  if (rc) return rc;
  return point_finish(c);
}

void free_rank_state(cora_ctx *c) {
  for (double **p : {&c->d_Y, &c->d_G, &c->d_rgrad, &c->d_G_trial}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  c->trial_x = nullptr;
  c->have_point = false;
}

}  // namespace

extern "C" {
#include "capi/handle.inc"
#include "capi/resident.inc"
#include "capi/preconditioner.inc"
#include "capi/products.inc"
#include "capi/solver_ops.inc"
#include "capi/stpcg.inc"
#include "capi/blocks.inc"
#include "capi/host_pointer.inc"

}  // extern "C"

#include "capi/comm.inc"
