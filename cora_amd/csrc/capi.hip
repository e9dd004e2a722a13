// C-ABI layer of libcora_hip.so (see include/cora_hip.h).  Owns the handle,
// device memory and stream; every compute entry point ends in a HIP kernel of
// kernels.hip -- there is no CPU fallback.
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

const char *cora_last_error(const cora_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int cora_ld_for(int k) { return ld_for(k); }

int cora_ctx_create_part(int device, int d, int n_poses, int n_ranges, int n_trans, const int32_t *rowptr,
                         const int32_t *colidx, const double *vals, int rank, int world, cora_ctx **out) {
  return cora_ctx_create_part_opts(device, d, n_poses, n_ranges, n_trans, rowptr, colidx, vals, rank, world, 0u, out);
}

int cora_ctx_create_part_opts(int device, int d, int n_poses, int n_ranges, int n_trans, const int32_t *rowptr,
                              const int32_t *colidx, const double *vals, int rank, int world, unsigned flags,
                              cora_ctx **out) {
  if (!out) return fail(nullptr, CORA_ERR_ARG, "out is null");
  *out = nullptr;
  if (!rowptr) return fail(nullptr, CORA_ERR_ARG, "null CSR pointer");
  {  // an empty Q (variables without a single measurement) has no index / value arrays to point at
    const int64_t N = static_cast<int64_t>(d) * n_poses + n_ranges + n_trans;
    static const int32_t no_col = 0;
    static const double no_val = 0.0;
    if (N > 0 && rowptr[N] == 0) {
      if (!colidx) colidx = &no_col;
      if (!vals) vals = &no_val;
    }
  }
  if (!colidx || !vals) return fail(nullptr, CORA_ERR_ARG, "null CSR pointer");
  cora_ctx *c = new (std::nothrow) cora_ctx();
  if (!c) return fail(nullptr, CORA_ERR_NOMEM, "out of host memory");
  try {
    build_format(d, n_poses, n_ranges, n_trans, rowptr, colidx, vals, rank, world, c->F,
                 (flags & CORA_PART_WHOLE_LONG_ROWS) == 0);
  } catch (const std::exception &e) {
    const std::string msg = e.what();
    delete c;
    return fail(nullptr, CORA_ERR_SHAPE, msg);
  }
  c->device = device;
  if (device < 0) {  // plan-only handle: format inspection / host tests
    *out = c;
    return CORA_OK;
  }
#define CREATE_TRY(expr)                                                        \
  do {                                                                          \
    hipError_t e__ = (expr);                                                    \
    if (e__ != hipSuccess) {                                                    \
      const std::string m = std::string(#expr) + ": " + hipGetErrorString(e__); \
      cora_ctx_destroy(c);                                                      \
      return fail(nullptr, CORA_ERR_HIP, m);                                    \
    }                                                                           \
  } while (0)
  CREATE_TRY(hipSetDevice(device));
  c->has_device = true;
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  const HostFormat &F = c->F;
  CREATE_TRY(to_device(&c->d_slices, F.slices));
  if (!F.slices_pose_first.empty()) CREATE_TRY(to_device(&c->d_slices_pf, F.slices_pose_first));
  CREATE_TRY(to_device(&c->d_head_val, F.head_val));
  if (F.L.world > 1) {
    // a slice is "boundary" when one of its stored columns is a row outside this rank's shard (padding entries carry
    // local columns; what a chain slice does not store belongs to local poses by construction: format_build.cpp puts
    // the couplings of a shard's first pose to the shard before into the general slots and the tail)
    std::vector<SliceDesc> in, bd;
    const int64_t lo = F.L.base, hi = F.L.base + F.L.shard_rows;
    std::vector<int32_t> cols;
    for (const SliceDesc &sd : F.slices) {
      bool remote = false;
      cols.clear();
      slice_columns(F, sd, cols);  // (the implied columns of a chain slice are local rows)
      if (sd.type & kSliceChainFlag) {
        // a chain slice's tail holds its poses' range measurements -- landmark rows, mostly another rank's.  Only the
        // general slots decide where the slice runs; a slice that could run ahead of the exchange but for its tail is
        // split: everything except the remote pairs of the tails with the interior slices, the remote pairs (added to
        // the translation rows) with the boundary slices.
        const size_t ngen = static_cast<size_t>(sd.width) * kWave;
        for (size_t q = 0; q < ngen && !remote; ++q) remote = cols[q] < lo || cols[q] >= hi;
        bool remote_tail = false;
        for (size_t q = ngen; q < cols.size() && !remote_tail; ++q) remote_tail = cols[q] < lo || cols[q] >= hi;
        if (!remote && remote_tail) {
          SliceDesc a = sd, b = sd;
          a.nrows |= kSliceSkipRemoteTail;
          b.nrows |= kSliceRemoteTailOnly;
          in.push_back(a);
          bd.push_back(b);
          continue;
        }
        (remote ? bd : in).push_back(sd);
        continue;
      }
      for (size_t q = 0; q < cols.size() && !remote; ++q) remote = cols[q] < lo || cols[q] >= hi;
      (remote ? bd : in).push_back(sd);
    }
    c->n_slices_int = static_cast<int>(in.size());
    c->n_slices_bnd = static_cast<int>(bd.size());
    CREATE_TRY(to_device(&c->d_slices_int, in));
    CREATE_TRY(to_device(&c->d_slices_bnd, bd));
    {
      // highest priority: the exchange's small kernels take the wavefront slots the product frees first, instead of
      // queueing behind a launch that fills the GPU
      int least = 0, greatest = 0;
      CREATE_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      CREATE_TRY(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, greatest));
    }
    CREATE_TRY(hipEventCreateWithFlags(&c->ev_operand, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&c->ev_exchanged, hipEventDisableTiming));
  }
  if (!F.long_rows.empty()) {
    CREATE_TRY(to_device(&c->d_long_rows, F.long_rows));
    CREATE_TRY(to_device(&c->d_long_owner, F.long_owner));
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_long_out), F.long_rows.size() * kMaxLD * sizeof(double)));
  }
  CREATE_TRY(to_device(&c->d_sval, F.sval));
  CREATE_TRY(to_device(&c->d_scol, F.scol));
  CREATE_TRY(to_device(&c->d_perm, F.perm));
  CREATE_TRY(to_device(&c->d_chunks, F.chunks));
  CREATE_TRY(to_device(&c->d_chunk_order, F.chunk_order));
  CREATE_TRY(to_device(&c->d_lval, F.lval));
  CREATE_TRY(to_device(&c->d_lcol, F.lcol));
  CREATE_TRY(to_device(&c->d_api2int, F.api2int));
  {
    std::vector<double> dinv(F.diag.size());
    for (size_t i = 0; i < dinv.size(); ++i) dinv[i] = 1.0 / F.diag[i];
    CREATE_TRY(to_device(&c->d_diag_inv, dinv));
  }
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_partials),
                       std::max<size_t>(F.chunks.size(), 1) * kMaxLD * sizeof(double)));
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_tickets),
                       std::max<size_t>(F.n_long_rows, 1) * sizeof(unsigned)));
  CREATE_TRY(hipMemset(c->d_tickets, 0, std::max<size_t>(F.n_long_rows, 1) * sizeof(unsigned)));
  // (+ 2 doubles: the pose slices' cooperative epilogue reads the blocks in pairs of doubles, kernels.hip)
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_lam_st),
                       (static_cast<size_t>(F.L.nl_poses) * d * d + 2) * sizeof(double)));
  CREATE_TRY(hipMemset(c->d_lam_st, 0, (static_cast<size_t>(F.L.nl_poses) * d * d + 2) * sizeof(double)));
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_lam_ob),
                       std::max<size_t>(F.L.nl_ranges, 1) * sizeof(double)));
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_scalars), 8 * sizeof(double)));
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_ticket), 4 * sizeof(unsigned)));  // [0] inner products, [1] kappa (k_spmm)
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_stpcg), sizeof(StpcgState)));
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_seq_counter), sizeof(unsigned long long)));
  CREATE_TRY(hipMemset(c->d_seq_counter, 0, sizeof(unsigned long long)));
  CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_stpcg), 2 * sizeof(StpcgState)));
  CREATE_TRY(hipMemset(c->d_ticket, 0, 4 * sizeof(unsigned)));
  CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_scalars), 8 * sizeof(double)));
  std::memset(c->h_scalars, 0, 8 * sizeof(double));
  CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_flag), sizeof(int)));
  CREATE_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_flag), sizeof(int)));
  CREATE_TRY(hipEventCreate(&c->ev0));
  CREATE_TRY(hipEventCreate(&c->ev1));
#undef CREATE_TRY
  *out = c;
  return CORA_OK;
}

int cora_ctx_create(int device, int d, int n_poses, int n_ranges, int n_trans, const int32_t *rowptr,
                    const int32_t *colidx, const double *vals, cora_ctx **out) {
  return cora_ctx_create_part(device, d, n_poses, n_ranges, n_trans, rowptr, colidx, vals, 0, 1, out);
}

void cora_ctx_destroy(cora_ctx *c) {
  if (!c) return;
  if (c->has_device) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    native_comm_destroy(c->native_comm);
    c->native_comm = nullptr;
    if (c->p2p_pending) cora::p2p_destroy(c->p2p_pending);
    c->p2p_pending = nullptr;
    free_rank_state(c);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->ev_operand) (void)hipEventDestroy(c->ev_operand);
    if (c->ev_exchanged) (void)hipEventDestroy(c->ev_exchanged);
    void *ptrs[] = {c->d_slices_int, c->d_slices_bnd, c->d_slices, c->d_slices_pf, c->d_head_val, c->d_long_out, c->d_long_rows, c->d_long_owner, c->d_sval, c->d_scol, c->d_perm, c->d_chunks, c->d_chunk_order, c->d_lval, c->d_lcol,
                    c->d_partials, c->d_tickets, c->d_api2int, c->d_diag_inv, c->d_lam_st, c->d_lam_ob,
                    c->d_stage, c->d_red, c->d_scalars, c->d_flag, c->d_ticket, c->d_stpcg, c->d_seq_counter};
    if (c->stpcg_graph) (void)hipGraphExecDestroy(c->stpcg_graph);
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
    for (int i = 0; i < kScratchSlots; ++i)
      if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    for (auto &p : c->user_allocs)
      if (p.first) (void)hipFree(p.first);
    for (auto &p : c->pool)
      if (p.first) (void)hipFree(p.first);
    for (auto *f : {&c->precond_f, &c->implicit_f, &c->aux_f})
      for (void *p : f->allocs)
        if (p) (void)hipFree(p);
    for (int i = 0; i < 2; ++i) {
      if (c->h_pin[i]) (void)hipHostFree(c->h_pin[i]);
      if (c->ev_pin[i]) (void)hipEventDestroy(c->ev_pin[i]);
    }
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->h_gram) (void)hipHostFree(c->h_gram);
    if (c->h_stpcg) (void)hipHostFree(c->h_stpcg);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->phase_events) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

int cora_set_rank(cora_ctx *c, int p) {
  if (!c) return CORA_ERR_ARG;
  if (p < c->F.L.d || p > kMaxLD)
    return fail(c, CORA_ERR_SHAPE, "relaxation rank must satisfy d <= p <= 24");
  if (p == c->p) return CORA_OK;
  c->p = p;
  c->ld = ld_for(p);
  c->have_point = false;
  if (c->has_device) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_rank_state(c);
    for (double **q : {&c->d_Y, &c->d_G, &c->d_rgrad}) {
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(q), vec_bytes(c, c->ld)));
      HIP_TRY(c, hipMemsetAsync(*q, 0, vec_bytes(c, c->ld), c->stream));
    }
  }
  return CORA_OK;
}

int cora_get_rank(const cora_ctx *c) { return c ? c->p : 0; }

int cora_set_stream(cora_ctx *c, void *hip_stream) {
  NEED_DEVICE(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = static_cast<hipStream_t>(hip_stream);
  c->own_stream = false;
  return CORA_OK;
}

int cora_ld(const cora_ctx *c) { return c ? c->ld : 0; }
int64_t cora_rows(const cora_ctx *c) { return c ? c->F.L.rows : 0; }
int64_t cora_shard_rows(const cora_ctx *c) { return c ? c->F.L.shard_rows : 0; }
int64_t cora_shard_begin(const cora_ctx *c) { return c ? c->F.L.base : 0; }
int64_t cora_nnz(const cora_ctx *c) { return c ? c->F.nnz_global : 0; }
int64_t cora_dim(const cora_ctx *c) { return c ? c->F.L.N : 0; }

int cora_row_map(const cora_ctx *c, int32_t *api_to_internal) {
  if (!c || !api_to_internal) return CORA_ERR_ARG;
  std::memcpy(api_to_internal, c->F.api2int.data(), c->F.api2int.size() * sizeof(int32_t));
  return CORA_OK;
}

int cora_remote_rows(const cora_ctx *c, int32_t *rows, int64_t *count) {
  if (!c || !count) return CORA_ERR_ARG;
  const Layout &L = c->F.L;
  const int64_t lo = L.base, hi = L.base + L.shard_rows;
  std::vector<char> seen(static_cast<size_t>(L.rows), 0);
  {
    std::vector<int32_t> cols;  // (scol also holds the chain slices' tail descriptors: only real columns count)
    for (const SliceDesc &sd : c->F.slices) slice_columns(c->F, sd, cols);
    for (int32_t col : cols) seen[col] = 1;  // padded slots repeat a real column of their lane
  }
  for (int32_t col : c->F.lcol) seen[col] = 1;
  int64_t n = 0;
  for (int64_t r = 0; r < L.rows; ++r)
    if (seen[r] && (r < lo || r >= hi)) {
      if (rows) rows[n] = static_cast<int32_t>(r);
      ++n;
    }
  *count = n;
  return CORA_OK;
}

int cora_long_rows(const cora_ctx *c, int32_t *api_rows, int64_t *count) {
  if (!c || !count) return CORA_ERR_ARG;
  const int64_t n = static_cast<int64_t>(c->F.long_rows.size());
  if (api_rows)
    for (int64_t j = 0; j < n; ++j) api_rows[j] = c->F.int2api[c->F.long_rows[static_cast<size_t>(j)]];
  *count = n;
  return CORA_OK;
}

int cora_precond_stats(const cora_ctx *c, int64_t s[4]) {
  if (!c || !s) return CORA_ERR_ARG;
  const auto &f = c->precond_f;
  s[0] = f.ready ? static_cast<int64_t>(f.plan.stages.size()) : 0;
  s[1] = f.ready ? f.plan.nnzW : 0;
  s[2] = f.ready ? f.plan.nnzL : 0;
  s[3] = f.ready && !f.plan.stages.empty() ? f.plan.stages.back().rows : 0;
  return CORA_OK;
}

// entries the solve plan of the preconditioner stores (bench.py's algorithmic bytes of the last stage; padding of the
// substitution blocks): [0] last stage, forward product | [1] last stage, backward product | [2], [3] entry slots of the
// substitution blocks' forward / backward sweep (null padding included) | [4] substitution blocks | [5] aux rows
int cora_precond_entries(const cora_ctx *c, int64_t s[6]) {
  if (!c || !s) return CORA_ERR_ARG;
  for (int k = 0; k < 6; ++k) s[k] = c->precond_f.ready ? c->precond_f.entries[k] : 0;
  return CORA_OK;
}

int cora_format_stats(const cora_ctx *c, int64_t s[8]) {
  if (!c || !s) return CORA_ERR_ARG;
  s[0] = static_cast<int64_t>(c->F.slices.size());
  s[1] = c->F.padded_nnz;
  s[2] = c->F.long_nnz;
  s[3] = c->F.n_long_rows;
  s[4] = static_cast<int64_t>(c->F.chunks.size());
  s[5] = c->F.L.local_rows;
  s[6] = c->F.nnz_local;
  s[7] = c->F.max_width;
  return CORA_OK;
}

int cora_format_bytes(const cora_ctx *c, int64_t b[4]) {
  if (!c || !b) return CORA_ERR_ARG;
  const HostFormat &F = c->F;
  b[0] = static_cast<int64_t>((F.sval.size() + F.lval.size()) * sizeof(double));
  b[1] = static_cast<int64_t>((F.scol.size() + F.lcol.size()) * sizeof(int32_t));
  b[2] = static_cast<int64_t>(F.slices.size() * sizeof(SliceDesc) + F.chunks.size() * sizeof(LongChunk) +
                              (F.perm.size() + F.chunk_order.size()) * sizeof(int32_t) + F.head_val.size() * sizeof(double));
  b[3] = b[0] + b[1] + b[2];
  return CORA_OK;
}

// ------------------------------------------------------------ resident API

// Resident vectors come from a small per-handle pool: TNT, the saddle escape and LOBPCG allocate and release 4-10
// vectors per call, three to six calls per staircase level, and every hipMalloc / hipFree is a device-wide
// synchronisation (a failed certification at 10^5 poses spent more time in them than in its eigensolver iterations).
// A released vector is kept (up to kPoolMax of them) and handed to the next request of the same size, zeroed on the
// handle's stream like a fresh one; everything returns to the driver with the handle.
int cora_dev_alloc(cora_ctx *c, int k, double **dptr) {
  NEED_DEVICE(c);
  if (!dptr || k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_ARG, "bad arguments");
  const size_t bytes = vec_bytes(c, ld_for(k));
  *dptr = nullptr;
  for (size_t i = 0; i < c->pool.size(); ++i)
    if (c->pool[i].second == bytes) {
      *dptr = c->pool[i].first;
      c->pool.erase(c->pool.begin() + static_cast<std::ptrdiff_t>(i));
      break;
    }
  if (!*dptr) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(dptr), bytes));
  wrote(c, *dptr);  // the allocator may return an address that was freed while a trial product of it was kept
  HIP_TRY(c, hipMemsetAsync(*dptr, 0, bytes, c->stream));
  for (auto &p : c->user_allocs)
    if (!p.first) {
      p = {*dptr, bytes};
      return CORA_OK;
    }
  c->user_allocs.push_back({*dptr, bytes});
  return CORA_OK;
}

int cora_dev_free(cora_ctx *c, double *dptr) {
  NEED_DEVICE(c);
  wrote(c, dptr);
  constexpr size_t kPoolMax = 16;
  for (auto &p : c->user_allocs)
    if (p.first == dptr && dptr) {
      if (c->pool.size() < kPoolMax) {
        c->pool.push_back(p);  // work already enqueued on the handle's stream is ordered before any reuse
      } else {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipFree(dptr);
      }
      p = {nullptr, 0};
      return CORA_OK;
    }
  return fail(c, CORA_ERR_ARG, "pointer was not allocated by cora_dev_alloc");
}

int cora_upload(cora_ctx *c, const double *host, int ld, int k, double *dptr) {
  NEED_DEVICE(c);
  wrote(c, dptr);
  int rc = upload_impl(c, host, ld, k, dptr);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host buffer may be pageable
  return CORA_OK;
}

int cora_download(cora_ctx *c, const double *dptr, int k, double *host, int ld) {
  NEED_DEVICE(c);
  return download_impl(c, dptr, k, host, ld);
}

int cora_set_point_dev(cora_ctx *c, const double *dY) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (!dY) return fail(c, CORA_ERR_ARG, "null pointer");
  return set_point_dev_impl(c, dY);
}

int cora_set_point(cora_ctx *c, const double *Y, int ldy) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  int rc = upload_impl(c, Y, ldy, c->p, c->d_Y);
  if (rc) return rc;
  return set_point_dev_impl(c, c->d_Y);
}

int cora_objective_dev(cora_ctx *c, const double *dY, double *f) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (!dY || !f) return fail(c, CORA_ERR_ARG, "null pointer");
  double *dG;
  int rc = get_scratch(c, 5, c->ld, &dG);
  if (rc) return rc;
  if ((rc = apply_product(c, dY, c->ld, EPI_NONE, dG))) return rc;
  double v = 0.0;
  const double *a[1] = {dY};
  const double *b[1] = {dG};
  rc = cora_dots_dev(c, 1, a, b, &v);
  if (rc) return rc;
  *f = 0.5 * v;
  return CORA_OK;
}

// One trust-region trial step in ONE wait (Optimization::Riemannian::TNT's outer iteration between two inner solves, as
// called from src/CORA.cpp:139-140): H s for the model decrease, the retraction, Q X of the trial point for its cost, and
// the four inner products in one reduction.  The same kernels, in the same order per vector, as cora_hvp_dev +
// cora_dots_dev + cora_retract_dev + cora_objective_dev: the values are the same bits, two waits and a launch fewer.
int cora_tnt_trial_dev(cora_ctx *c, const double *dS, double *dHs, double *dXprop, double out[4]) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  if (!dS || !dHs || !dXprop || !out || dHs == dS || dXprop == dS || dXprop == dHs) return fail(c, CORA_ERR_ARG, "bad arguments");
  c->trial_x = nullptr;
  int rc;
  if (!c->d_G_trial) {
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_G_trial), vec_bytes(c, c->ld)));
    HIP_TRY(c, hipMemsetAsync(c->d_G_trial, 0, vec_bytes(c, c->ld), c->stream));
  }
  if ((rc = apply_product(c, dS, c->ld, EPI_HVP, dHs))) return rc;
  HIP_TRY(c, launch_project_manifold(row_args(c), c->ld, c->d_Y, dS, 1.0, dXprop, c->stream));
  if ((rc = apply_product(c, dXprop, c->ld, EPI_NONE, c->d_G_trial))) return rc;
  const double *A[4] = {c->d_rgrad, dS, dS, dXprop};
  const double *B[4] = {dS, dHs, dS, c->d_G_trial};
  if ((rc = cora_dots_dev(c, 4, A, B, out))) return rc;
  out[3] *= 0.5;
  c->trial_x = dXprop;
  return CORA_OK;
}

// The accepted trial point becomes the current point, and the preconditioned gradient with the norms TNT's stopping
// tests need comes back in the same wait: out = f, <g, g>, <P g, P g>, <g, P g>.  After cora_tnt_trial_dev of the same
// vector (one GPU) the product Q X is not formed again.  Otherwise: cora_set_point_dev + cora_precondition_projected_dev
// + cora_dots_dev, call by call.
int cora_tnt_accept_dev(cora_ctx *c, const double *dX, double *dPg, double out[4]) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (!dX || !dPg || !out || dX == dPg) return fail(c, CORA_ERR_ARG, "bad arguments");
  int rc;
  const bool fast = c->F.L.world == 1 && c->trial_x == dX && c->d_G_trial && !std::getenv("CORA_NO_TNT_FUSE");
  if (fast) {
    c->trial_x = nullptr;
    if (dX != c->d_Y) HIP_TRY(c, hipMemcpyAsync(c->d_Y, dX, vec_bytes(c, c->ld), hipMemcpyDeviceToDevice, c->stream));
    std::swap(c->d_G, c->d_G_trial);
    if ((rc = point_finish(c, false))) return rc;
  } else if ((rc = set_point_dev_impl(c, dX))) {
    return rc;
  }
  if ((rc = cora_precondition_projected_dev(c, c->d_rgrad, dPg))) return rc;
  const double *A[3] = {c->d_rgrad, dPg, c->d_rgrad};
  const double *B[3] = {c->d_rgrad, dPg, dPg};
  if ((rc = cora_dots_dev(c, 3, A, B, out + 1))) return rc;
  if (fast) c->f = c->h_scalars[4];
  out[0] = c->f;
  return CORA_OK;
}

int cora_point_cost(cora_ctx *c, double *f) {
  if (!c || !f) return CORA_ERR_ARG;
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  *f = c->f;
  return CORA_OK;
}

const double *cora_point_Y_dev(const cora_ctx *c) { return (c && c->have_point) ? c->d_Y : nullptr; }
const double *cora_point_egrad_dev(const cora_ctx *c) { return (c && c->have_point) ? c->d_G : nullptr; }
const double *cora_point_rgrad_dev(const cora_ctx *c) { return (c && c->have_point) ? c->d_rgrad : nullptr; }

int cora_spmm_dev(cora_ctx *c, const double *dX, int k, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  if (!dX || !dOut || k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_ARG, "bad arguments");
  return apply_product(c, dX, ld_for(k), EPI_NONE, dOut);
}

int cora_hvp_dev(cora_ctx *c, const double *dX, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  NEED_RANK(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  if (!dX || !dOut) return fail(c, CORA_ERR_ARG, "null pointer");
  return apply_product(c, dX, c->ld, EPI_HVP, dOut);
}

int cora_certificate_product_dev(cora_ctx *c, const double *dX, int k, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  if (!dX || !dOut || k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_ARG, "bad arguments");
  return exchange_and_product(c, spmm_args(c, dX, dOut), ld_for(k), EPI_S);
}

int cora_tangent_space_projection_dev(cora_ctx *c, const double *dV, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  NEED_RANK(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  HIP_TRY(c, launch_tangent_project(row_args(c), c->ld, c->d_Y, dV, nullptr, dOut, c->stream));
  return CORA_OK;
}

int cora_precond_setup(cora_ctx *c, int kind) {
  if (!c) return CORA_ERR_ARG;
  if (kind == CORA_PRECOND_NONE || kind == CORA_PRECOND_JACOBI) {
    if (kind == CORA_PRECOND_JACOBI)
      for (double v : c->F.diag)
        if (!(v != 0.0)) return fail(c, CORA_ERR_NAN, "zero on the diagonal of Q: Jacobi preconditioner undefined");
    c->precond = kind;
    return CORA_OK;
  }
  if (kind == CORA_PRECOND_BLOCK_CHOLESKY || kind == CORA_PRECOND_REGULARIZED_CHOLESKY) {
    if (!c->precond_f.ready)
      return fail(c, CORA_ERR_NOT_READY,
                  "Cholesky preconditioners need a factor installed with cora_precond_set_cholesky");
    c->precond = kind;
    return CORA_OK;
  }
  return fail(c, CORA_ERR_ARG, "unknown preconditioner kind");
}

// Builds the level schedule of a factor and uploads it.  row_of[i] = internal row of permuted variable i.
static int install_factor(cora_ctx *c, cora_ctx::DevFactor &f, int m, const int32_t *Lp, const int32_t *Li,
                          const double *Lx, const std::vector<int32_t> &row_of, int32_t zero_row,
                          const std::vector<int32_t> *group = nullptr) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "  [install] %-26s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  f.chunk_at = 0;
  f.chunk_used = 0;
  f.stages.clear();
  f.ready = false;
  f.aux_rows = 0;
  f.fuse_ok = false;
  try {
    build_tri_plan(m, Lp, Li, Lx, row_of, zero_row, f.plan, group, static_cast<int32_t>(c->F.L.rows));
  } catch (const std::exception &e) {
    return fail(c, CORA_ERR_ARG, e.what());
  }
  tick("plan (host)");
  auto up = [&](auto **dst, const auto &vec) -> hipError_t {
    using T = typename std::remove_reference<decltype(vec)>::type::value_type;
    const size_t bytes = (std::max<size_t>(vec.size(), 1) * sizeof(T) + 255) & ~static_cast<size_t>(255);
    while (f.chunk_at < f.allocs.size() && f.chunk_used + bytes > f.chunk_bytes[f.chunk_at]) {
      ++f.chunk_at;
      f.chunk_used = 0;
    }
    if (f.chunk_at == f.allocs.size()) {
      const size_t cb = std::max<size_t>(bytes, static_cast<size_t>(64) << 20);
      void *q = nullptr;
      const hipError_t e = hipMalloc(&q, cb);
      if (e != hipSuccess) return e;
      f.allocs.push_back(q);
      f.chunk_bytes.push_back(cb);
      f.chunk_used = 0;
    }
    T *p = reinterpret_cast<T *>(static_cast<char *>(f.allocs[f.chunk_at]) + f.chunk_used);
    f.chunk_used += bytes;
    *dst = p;
    if (vec.empty()) return hipSuccess;
    return hipMemcpy(p, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice);  // (vec may be a temporary)
  };
  auto up_op = [&](RowOpDev &D, RowOpHost &H) -> hipError_t {
    hipError_t e;
    D.n8 = H.n8;
    D.n64 = H.n64;
    D.nlong = static_cast<int>(H.long_out.size());
    D.nchunks = static_cast<int>(H.chunk_begin.size());
    if ((e = up(&D.out_row, H.out_row)) != hipSuccess) return e;
    if ((e = up(&D.begin, H.begin)) != hipSuccess) return e;
    if ((e = up(&D.end, H.end)) != hipSuccess) return e;
    if ((e = up(&D.long_out, H.long_out)) != hipSuccess) return e;
    if ((e = up(&D.long_chunk_ptr, H.long_chunk_ptr)) != hipSuccess) return e;
    if ((e = up(&D.chunk_begin, H.chunk_begin)) != hipSuccess) return e;
    if ((e = up(&D.chunk_end, H.chunk_end)) != hipSuccess) return e;
    if ((e = up(&D.col, H.col)) != hipSuccess) return e;
    if ((e = up(&D.val, H.val)) != hipSuccess) return e;
    std::vector<int32_t> chunk_row(static_cast<size_t>(D.nchunks), 0);
    for (int r = 0; r < D.nlong; ++r)
      for (int32_t ch = H.long_chunk_ptr[r]; ch < H.long_chunk_ptr[r + 1]; ++ch) chunk_row[ch] = r;
    if ((e = up(&D.chunk_row, chunk_row)) != hipSuccess) return e;
    const std::vector<unsigned> tick(static_cast<size_t>(std::max(D.nlong, 1)), 0u);
    const unsigned *tp = nullptr;
    if ((e = up(&tp, tick)) != hipSuccess) return e;
    D.tickets = const_cast<unsigned *>(tp);
    const std::vector<double> part(static_cast<size_t>(std::max(D.nchunks, 1)) * kMaxLD, 0.0);
    const double *pp = nullptr;
    if ((e = up(&pp, part)) != hipSuccess) return e;
    D.partial = const_cast<double *>(pp);
    H = RowOpHost();  // the host copy is not needed any more
    return hipSuccess;
  };
  const size_t K = f.plan.stages.size();
  f.stages.resize(K);
  std::shared_ptr<SubBlockOpHost> dead_sub;
  for (int64_t &e : f.entries) e = 0;
  if (K > 0) {
    const TriStage &top = f.plan.stages.back();
    f.entries[0] = static_cast<int64_t>(top.fwd_b.val.size());
    f.entries[1] = static_cast<int64_t>(top.bwd_b.val.size());
    if (f.plan.stages[0].sub) {
      const SubBlockOpHost &o = f.plan.stages[0].sub_op;
      f.entries[2] = static_cast<int64_t>(o.f_val.size());
      f.entries[3] = static_cast<int64_t>(o.b_val.size());
      f.entries[4] = static_cast<int64_t>(o.nrows.size());
      f.entries[5] = o.n_aux;
    }
  }
  for (size_t k = 0; k < K; ++k) {
    TriStage &S = f.plan.stages[k];
    cora_ctx::DevStage &D = f.stages[k];
    D.has_fwd_a = k > 0;
    D.has_bwd_a = k + 1 < K;
    D.dense = S.dense;
    if (S.sub) {
      SubBlockOpHost &H = S.sub_op;
      D.is_sub = true;
      std::vector<SubDesc> desc(H.nrows.size());
      for (size_t b = 0; b < desc.size(); ++b) {
        SubDesc &d = desc[b];
        d.row_begin = H.row_begin[b];
        d.nrows = H.nrows[b];
        d.f_ent_begin = H.f_ent_begin[b];
        d.f_nent = H.f_nent[b];
        d.b_ent_begin = H.b_ent_begin[b];
        d.b_nent = H.b_nent[b];
        d.f_lev_begin = H.f_lev_begin[b];
        d.f_nlev = (H.f_lev_begin[b + 1] - H.f_lev_begin[b]) / 4 - 1;  // barrier levels: one header per wavefront (4) each, + the closing one
        d.b_lev_begin = H.b_lev_begin[b];
        d.b_nlev = (H.b_lev_begin[b + 1] - H.b_lev_begin[b]) / 4 - 1;
        d.tgt_begin = H.tgt_begin[b];
        d.ntgt = H.tgt_begin[b + 1] - H.tgt_begin[b];
      }
      SubOpDev &Q = D.sub;
      Q.nblocks = static_cast<int>(desc.size());
      Q.ntop = static_cast<int>(f.plan.top_rows.size());
      Q.max_rows = H.max_rows;
      Q.max_ent = H.max_ent;
      Q.max_lev = H.max_lev;
      Q.max_level_lanes = H.max_level_lanes;
      Q.max_npl = H.max_npl;
      Q.aux_base = f.plan.aux_base;
      HIP_TRY(c, up(&Q.fwd.rows, H.rows));
      H.f_hdr.resize(H.f_hdr.size() + 8, 0);  // the kernel reads one header ahead
      H.f_idx.resize(H.f_idx.size() + 8, 0);  // ... and an entry past a block without entries
      H.f_val.resize(H.f_val.size() + 8, 0.0);
      H.b_idx.resize(H.b_idx.size() + 8, 0);
      H.b_val.resize(H.b_val.size() + 8, 0.0);
      H.b_hdr.resize(H.b_hdr.size() + 8, 0);
      HIP_TRY(c, up(&Q.fwd.hdr, H.f_hdr));
      HIP_TRY(c, up(&Q.fwd.idx, H.f_idx));
      // LAB BUILDS ONLY (-DCORA_SUB_F32=1 on capi.hip AND the kernels_tri units: the sweeps then read fp32 coefficients).  A
      // compile-time constant since round 6: as an environment switch of the product library it uploaded floats into a buffer
      // the default kernels read as doubles.  Measured at 10^5 poses, p = 5 (profiles/r06_kernel_evolution.md): forward sweep
      // 35.8 -> 33.3 us, backward 35.2 -> 34.6, iteration 110.5 -> 107.2 us (3 %): the sweeps are not bound by the factor's bytes.
#if defined(CORA_SUB_F32) && CORA_SUB_F32
      constexpr bool f32 = true;
#else
      constexpr bool f32 = false;
#endif
      if (f32) {
        std::vector<float> ff(H.f_val.begin(), H.f_val.end()), fb(H.b_val.begin(), H.b_val.end());
        HIP_TRY(c, up(reinterpret_cast<float **>(const_cast<double **>(&Q.fwd.val)), ff));
        HIP_TRY(c, up(reinterpret_cast<float **>(const_cast<double **>(&Q.bwd.val)), fb));
      } else
      HIP_TRY(c, up(&Q.fwd.val, H.f_val));
      HIP_TRY(c, up(&Q.bwd.rows, H.b_rows));
      HIP_TRY(c, up(&Q.bwd.hdr, H.b_hdr));
      HIP_TRY(c, up(&Q.bwd.idx, H.b_idx));
      if (!f32) HIP_TRY(c, up(&Q.bwd.val, H.b_val));
      HIP_TRY(c, up(&Q.tgt_row, H.tgt_row));
      HIP_TRY(c, up(&Q.tgt_slot, H.tgt_slot));
      HIP_TRY(c, up(&Q.c_ptr, H.c_ptr));
      HIP_TRY(c, up(&Q.c_idx, H.c_idx));
      HIP_TRY(c, up(&Q.c_val, H.c_val));
      HIP_TRY(c, up(&Q.top_rows, f.plan.top_rows));
      f.aux_rows = H.n_aux;
      tick("  sub: desc + arrays");
      {
        // memory-order I/O lists of both sweeps: {internal row, tile position} of every block row, sorted by row
        auto io_of = [&](const std::vector<int32_t> &rows) {
          std::vector<int2> io(rows.size() + 1);  // (+ 1: a block without rows still forms an address)
          const size_t nblk = desc.size();
          const unsigned nth = nblk < 64 ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
          auto part = [&](unsigned t) {  // blocks are independent: a range of them per thread
            std::vector<int32_t> ord;
            for (size_t b = nblk * t / nth; b < nblk * (t + 1) / nth; ++b) {
              const int32_t r0 = desc[b].row_begin, nb = desc[b].nrows;
              ord.resize(static_cast<size_t>(nb));
              for (int k = 0; k < nb; ++k) ord[k] = k;
              std::sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return rows[r0 + x] < rows[r0 + y]; });
              for (int k = 0; k < nb; ++k) io[static_cast<size_t>(r0) + k] = make_int2(rows[r0 + ord[k]], ord[k]);
            }
          };
          cora::parallel_parts(nth, part);
          io.back() = make_int2(0, 0);
          if (std::getenv("CORA_IO_STATS")) {  // lab: runs of consecutive rows per block
            size_t tot = 0, mx = 0;
            std::vector<size_t> hist(12, 0);
            for (size_t b = 0; b < nblk; ++b) {
              const int32_t r0 = desc[b].row_begin, nb = desc[b].nrows;
              size_t runs = nb > 0;
              for (int k = 1; k < nb; ++k) runs += io[static_cast<size_t>(r0) + k].x != io[static_cast<size_t>(r0) + k - 1].x + 1;
              tot += runs; mx = std::max(mx, runs); hist[std::min<size_t>(runs, 11)]++;
            }
            std::fprintf(stderr, "io runs per block: mean %.2f max %zu, histogram", double(tot) / std::max<size_t>(nblk, 1), mx);
            for (size_t h : hist) std::fprintf(stderr, " %zu", h);
            std::fprintf(stderr, "\n");
          }
          return io;
        };
        // (the rows of a block in memory order are the same set for both sweeps: the runs are found once, from the
        // forward list; only the tile positions differ)
        {
          const std::vector<int2> iof = io_of(H.rows), iob = io_of(H.b_rows);
          bool runs_ok = std::getenv("CORA_SUB_IO_LISTS") == nullptr;  // (lab switch: the 8-byte index lists)
          std::vector<uint16_t> tpf(iof.size() + 8, 0), tpb(iob.size() + 8, 0);
          for (size_t b = 0; b < desc.size(); ++b) {
            SubDesc &d = desc[b];
            const int32_t r0 = d.row_begin, nb = d.nrows;
            int nr = 0;
            for (int q = 0; q < kSubMaxRuns; ++q) { d.run_off[q] = 0; d.run_end[q] = INT32_MAX; }
            for (int k = 0; k < nb; ++k) {
              const int2 a = iof[static_cast<size_t>(r0) + k], bb = iob[static_cast<size_t>(r0) + k];
              if (a.x != bb.x || a.y > 0xffff || bb.y > 0xffff) runs_ok = false;
              tpf[static_cast<size_t>(r0) + k] = static_cast<uint16_t>(a.y);
              tpb[static_cast<size_t>(r0) + k] = static_cast<uint16_t>(bb.y);
              if (k == 0 || a.x != iof[static_cast<size_t>(r0) + k - 1].x + 1) {  // a new run starts at k
                if (nr > 0 && nr <= kSubMaxRuns) d.run_end[nr - 1] = k;
                if (nr < kSubMaxRuns) d.run_off[nr] = a.x - k;
                ++nr;
              }
            }
            if (nr > kSubMaxRuns) runs_ok = false;
          }
          Q.io_runs = runs_ok ? 1 : 0;
          HIP_TRY(c, up(&Q.fwd.io, iof));
          HIP_TRY(c, up(&Q.bwd.io, iob));
          HIP_TRY(c, up(&Q.fwd.tpos, tpf));
          HIP_TRY(c, up(&Q.bwd.tpos, tpb));
        }
        tick("  sub: io lists");
        // fused projection in the backward sweep: the first rotation row of a pose finds the others right behind it
        // in the tile, and a pose of the last stage has all its rows there.  Row units of a block: {tile position, row}
        const Layout &L = c->F.L;
        const int64_t rot0 = L.rot_base, rot1 = L.rot_base + static_cast<int64_t>(L.d) * L.nl_poses;
        bool ok = group != nullptr;  // (a shard's rows are rotations | ranges | translations in this order too: the same tests on the row index)
        std::vector<int2> units;
        for (size_t b = 0; b < desc.size(); ++b) {
          const int32_t *rows = H.b_rows.data() + desc[b].row_begin;
          const int nb = desc[b].nrows;
          int64_t leaders = 0, rot_rows = 0;
          desc[b].unit_begin = static_cast<int32_t>(units.size());
          for (int k = 0; k < nb; ++k) {
            if (rows[k] < rot0 || rows[k] >= rot1) {
              units.push_back(make_int2(k, rows[k]));
              continue;
            }
            ++rot_rows;
            if ((rows[k] - rot0) % L.d != 0) continue;
            ++leaders;
            units.push_back(make_int2(k, rows[k]));
            for (int a = 1; a < L.d && ok; ++a) ok = k + a < nb && rows[k + a] == rows[k] + a;
          }
          desc[b].nunits = static_cast<int32_t>(units.size()) - desc[b].unit_begin;
          ok = ok && rot_rows == leaders * L.d;
        }
        units.push_back(make_int2(0, 0));
        HIP_TRY(c, up(&Q.b_unit, units));
        if (ok) {
          std::vector<char> in_top(static_cast<size_t>(L.rows), 0);
          for (int32_t r : f.plan.top_rows) in_top[r] = 1;
          for (int32_t r : f.plan.top_rows)
            if (r >= rot0 && r < rot1) {
              const int64_t lead = r - (r - rot0) % L.d;
              for (int a = 0; a < L.d && ok; ++a) ok = in_top[lead + a] != 0;
            }
        }
        f.fuse_ok = ok;
        if (std::getenv("CORA_TRI_TIMING")) std::fprintf(stderr, "  [tri plan] sweep fusion possible: %d\n", int(ok));
      }
      HIP_TRY(c, up(&Q.desc, desc));
      tick("  sub: units");
      // the host copy is not needed any more: 130 MB of vectors, 16 ms to hand back at 10^5 poses and 0.1 s at 10^6 -- on
      // a thread of its own (joined before the next factor is installed and when the handle goes), started AFTER the
      // last stage's uploads: a thread that unmaps 130 MB holds the address space's lock, and the next copy from pageable
      // memory waited 12 ms for it (measured: a 2.3 MB copy, 0.0122 s)
      dead_sub = std::make_shared<SubBlockOpHost>(std::move(H));
      H = SubBlockOpHost();
      continue;
    }
    if (k == 1 && f.stages[0].is_sub) {  // the last stage of a two-stage plan: only its two explicit-inverse products
      D.aux_sum = !S.fwd_a.empty();      // (+ the sum of the aux rows as a product of its own on large plans)
      if (D.aux_sum) HIP_TRY(c, up_op(D.fwd_a, S.fwd_a));
      HIP_TRY(c, up_op(D.fwd_b, S.fwd_b));
      tick("  top: forward product");
      HIP_TRY(c, up_op(D.bwd_b, S.bwd_b));
      tick("  top: backward product");
      continue;
    }
    if (D.has_fwd_a) HIP_TRY(c, up_op(D.fwd_a, S.fwd_a));
    if (S.dense) {
      BlockOpHost &H = S.blocks_op;
      D.blocks.nblocks = static_cast<int>(H.nrows.size());
      {
        // per-row records, padded per block to a multiple of eight (empty masks) + one spare round at the end; the
        // value arrays get 64 * 65 zero entries: the kernel's prefetch of the next round reads past a block's end
        std::vector<BlockDesc> desc(H.nrows.size());
        std::vector<BlockLane> bc, br;
        for (size_t b = 0; b < desc.size(); ++b) {
          desc[b] = BlockDesc{H.row_begin[b], H.nrows[b], static_cast<int32_t>(bc.size()), 0, H.w_off[b], 0};
          for (int l = 0; l < H.nrows[b]; ++l) {
            const size_t i = static_cast<size_t>(H.row_begin[b]) + l;
            bc.push_back(BlockLane{H.mask_col[i], H.off_col[i], H.rows[i]});
            br.push_back(BlockLane{H.mask_row[i], H.off_row[i], H.rows[i]});
          }
          while (bc.size() % 8) {
            bc.push_back(BlockLane{0, 0, 0});
            br.push_back(BlockLane{0, 0, 0});
          }
        }
        bc.resize(bc.size() + 24, BlockLane{0, 0, 0});
        br.resize(br.size() + 24, BlockLane{0, 0, 0});
        H.w_by_col.resize(H.w_by_col.size() + 64 * 65, 0.0);
        H.w_by_row.resize(H.w_by_row.size() + 64 * 65, 0.0);
        HIP_TRY(c, up(&D.blocks.desc, desc));
        HIP_TRY(c, up(&D.blocks.by_col, bc));
        HIP_TRY(c, up(&D.blocks.by_row, br));
      }
      HIP_TRY(c, up(&D.blocks.w_by_col, H.w_by_col));
      HIP_TRY(c, up(&D.blocks.w_by_row, H.w_by_row));
      HIP_TRY(c, up(&D.blocks.ext_ptr, H.ext_ptr));
      HIP_TRY(c, up(&D.blocks.ext_col, H.ext_col));
      HIP_TRY(c, up(&D.blocks.ext_val, H.ext_val));
      H = BlockOpHost();
      continue;
    }
    HIP_TRY(c, up_op(D.fwd_b, S.fwd_b));
    if (D.has_bwd_a) HIP_TRY(c, up_op(D.bwd_a, S.bwd_a));
    HIP_TRY(c, up_op(D.bwd_b, S.bwd_b));
  }
  if (dead_sub) {
    if (c->deferred_free.valid()) c->deferred_free.get();
    c->deferred_free = std::async(std::launch::async, [dead = std::move(dead_sub)]() mutable { dead.reset(); });
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  tick("upload");
  f.ready = true;
  ++f.generation;
  return CORA_OK;
}

// out[rows of the factor] = (P^T L L^T P)^-1 rhs[rows of the factor]; rows outside the factor are not
// written.  rhs and out must be different resident vectors.  A fixed sequence of 4K-2 sparse products
// (trisolve.h): forward stages ascending, backward stages descending.
// project != nullptr (two-stage plans that allow it): the solution leaves the backward sweep projected to the tangent
// space of the current point (SubFuse with p == nullptr) and *project is set; otherwise the caller projects.
static int factor_solve(cora_ctx *c, cora_ctx::DevFactor &f, int ld, const double *rhs, double *out, bool *project = nullptr) {
  if (project) *project = false;
  if (rhs == out) return fail(c, CORA_ERR_ARG, "factor_solve: output aliases the right-hand side");
  const int K = static_cast<int>(f.stages.size());
  if (K == 0) return CORA_OK;
  double *t, *t2;
  int rc;
  if ((rc = get_scratch(c, 6, ld, &t, f.aux_rows))) return rc;
  if ((rc = get_scratch(c, 7, ld, &t2))) return rc;
  if (f.stages[0].is_sub) {  // two-stage plan: substitution blocks around one explicit inverse (trisolve.h)
    const cora_ctx::DevStage &S0 = f.stages[0], &S1 = f.stages[1];
    HIP_TRY(c, launch_subblock(S0.sub, ld, false, rhs, t, out, c->stream));  // y_0 -> out, couplings + rhs_1 -> t
    if (S1.aux_sum) HIP_TRY(c, launch_rowop(S1.fwd_a, ld, t, t, t, c->stream));  // t_1 += its aux rows (in place: a row is read and written by its own lanes only)
    HIP_TRY(c, launch_rowop(S1.fwd_b, ld, nullptr, t, t2, c->stream));       // y_1 = W_1 t_1 (with the aux sums folded in otherwise)
    HIP_TRY(c, launch_rowop(S1.bwd_b, ld, nullptr, t2, t, c->stream));       // x_1 = W_1^T y_1 -> t
    if (project && f.fuse_ok && ld * c->F.L.d <= 24 && ld <= 12 && c->have_point) {
      const Layout &L = c->F.L;
      SubFuse FB;
      FB.dot.st = c->d_stpcg;  // (coefficients unused in this mode)
      FB.Y = c->d_Y;
      FB.d = L.d;
      FB.rot_base = L.rot_base;
      FB.rng_base = L.rng_base;
      FB.trn_base = L.trn_base;
      HIP_TRY(c, launch_subblock_fused(S0.sub, ld, true, FB, t, out, c->stream));  // Proj_Y(x) -> out
      *project = true;
      return CORA_OK;
    }
    HIP_TRY(c, launch_subblock(S0.sub, ld, true, out, t, out, c->stream));   // x_0, and x_1 -> out
    return CORA_OK;
  }
  for (int k = 0; k < K; ++k) {  // L y = rhs
    const cora_ctx::DevStage &S = f.stages[k];
    if (S.dense) {  // stage 0, never the last one
      HIP_TRY(c, launch_blockop(S.blocks, ld, false, rhs, out, c->stream));
      continue;
    }
    const double *tk = rhs;
    if (S.has_fwd_a) {
      HIP_TRY(c, launch_rowop(S.fwd_a, ld, rhs, out, t, c->stream));   // t_k = rhs_k - L[k,<k] y_<k
      tk = t;
    }
    HIP_TRY(c, launch_rowop(S.fwd_b, ld, nullptr, tk, k == K - 1 ? t2 : out, c->stream));  // y_k = W_k t_k
  }
  for (int k = K - 1; k >= 0; --k) {  // L^T x = y
    const cora_ctx::DevStage &S = f.stages[k];
    if (S.dense) {
      HIP_TRY(c, launch_blockop(S.blocks, ld, true, out, out, c->stream));
      continue;
    }
    const double *tk = t2;  // last stage: y_K-1 was left in t2
    if (S.has_bwd_a) {
      HIP_TRY(c, launch_rowop(S.bwd_a, ld, out, out, t, c->stream));   // t_k = y_k - L[>k,k]^T x_>k
      tk = t;
    }
    HIP_TRY(c, launch_rowop(S.bwd_b, ld, nullptr, tk, out, c->stream));  // x_k = W_k^T t_k
  }
  return CORA_OK;
}

int cora_precond_set_cholesky(cora_ctx *c, int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                              const int32_t *perm) {
  NEED_DEVICE(c);
  const int64_t N = c->F.L.N;
  const Layout &Lo = c->F.L;
  // Partitioned handle: the factor is the one of THIS RANK'S rows -- the diagonal block of (Q + lambda I) on the rows
  // of its shard, all of them or all but one (the pinned variable, if it lives here): block Jacobi over the ranks.
  const bool sharded = Lo.world != 1;
  const int64_t owned = sharded ? Lo.local_rows : N;
  if (!Lp || !Li || !Lx || !perm || (m != owned && m != owned - 1))
    return fail(c, CORA_ERR_ARG, sharded ? "factor must cover the rank's own rows (all, or all but the pinned one)"
                                         : "factor must have N or N-1 rows");
  std::vector<int32_t> row_of(static_cast<size_t>(m));
  std::vector<char> seen(static_cast<size_t>(N), 0);
  for (int i = 0; i < m; ++i) {
    if (perm[i] < 0 || perm[i] >= N || seen[perm[i]]) return fail(c, CORA_ERR_ARG, "perm is not a permutation");
    seen[perm[i]] = 1;
    row_of[i] = c->F.api2int[perm[i]];
    if (sharded && (row_of[i] < Lo.base || row_of[i] >= Lo.base + Lo.shard_rows))
      return fail(c, CORA_ERR_ARG, "the factor of a partitioned handle may only hold rows of its own shard");
  }
  int32_t zero_row = -1;  // blockCholeskySolve: last row zeroed, src/CORA_preconditioners.cpp:78-79
  if (m == owned - 1)
    for (int64_t i = 0; i < N; ++i) {
      const int32_t ir = c->F.api2int[i];
      if (!seen[i] && (!sharded || (ir >= Lo.base && ir < Lo.base + Lo.shard_rows))) zero_row = ir;
    }
  // the d rotation rows of a pose stay in one block of the solve plan (row-unit work can then be fused into it)
  std::vector<int32_t> group(static_cast<size_t>(m), -1);
  const int64_t dn = static_cast<int64_t>(c->F.L.d) * c->F.L.n;
  for (int i = 0; i < m; ++i)
    if (perm[i] < dn) group[i] = perm[i] / c->F.L.d;
  return install_factor(c, c->precond_f, m, Lp, Li, Lx, row_of, zero_row, &group);
}

// dOut = [ (Q + lambda I)[0:m]^-1 V[0:m] ; 0 ]
static int chol_solve(cora_ctx *c, int ld, const double *dV, double *dOut) {
  return factor_solve(c, c->precond_f, ld, dV, dOut);  // the pinned row is zeroed by the plan itself
}

// ---- translation-implicit formulation (src/CORA_problem.cpp:714-753) ------------------------
// Q_impl Y = Qmain Y - B M^-1 B^T Y with M = Q33[0:nt-1] = top rows of Q [Y; t; 0] for
// t = -M^-1 (B^T Y), and B^T Y = translation rows of Q [Y; 0].  Two explicit products and
// one triangular solve on the translation block; vectors keep N rows, translation rows of the
// input are ignored and translation rows of the output are zero.
// Partitioned handle: the two products are the partitioned products (one exchange of the operand each); the translation
// solve in between is a recurrence over the whole chain, so it is REPLICATED -- the right-hand side's rows are gathered
// (whole shards in place: one collective) and every rank runs the same solve plan on the same numbers, then keeps its
// own translations (and already holds the remote ones the second product reads).  Exact: the operator is the single
// handle's, whatever the partition.
static int64_t pinned_translation_row(const cora_ctx *c) { return c->F.api2int[static_cast<size_t>(c->F.L.N) - 1]; }

static int implicit_lift(cora_ctx *c, const double *dX, int ld, double *w0, double *w1) {
  const Layout &L = c->F.L;
  const bool sharded = L.world != 1;
  const size_t toff = static_cast<size_t>(L.trn_base) * ld;
  const size_t tbytes = static_cast<size_t>(L.nl_trans) * ld * sizeof(double);
  HIP_TRY(c, hipMemcpyAsync(w0, dX, vec_bytes(c, ld), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(w0 + toff, 0, tbytes, c->stream));
  SpmmArgs A = spmm_args(c, w0, w1);
  int rc;
  if (sharded) {
    if ((rc = exchange_and_product(c, A, ld, EPI_NONE))) return rc;               // (the exported translations are zero)
  } else {
    HIP_TRY(c, launch_spmm(A, ld, L.d, EPI_NONE, c->stream));                      // w1[trans] = B^T X
  }
  const int64_t last = pinned_translation_row(c);                                  // pinned translation (the API's last row)
  const bool mine = last >= L.base && last < L.base + L.shard_rows;
  if (mine) HIP_TRY(c, launch_zero_row(w1, static_cast<size_t>(last), ld, c->stream));
  if (sharded) {  // every rank: all translation rows of B^T X (the solve is replicated)
    if (c->native_comm && c->comm_user == c->native_comm && !std::getenv("CORA_IMPLICIT_WHOLE_GATHER")) {
      // the library's own communication gathers the translation rows alone, packed: 2 / 9 of a shard's rows at d = 3
      if (native_allgather_rows(c->native_comm, w1, ld, L.trn_base - L.base, L.nl_trans))
        return fail(c, CORA_ERR_HIP, "all-gather step failed: " + native_error(c->native_comm));
    } else if ((rc = comm_allgather(c, w1, ld))) {
      return rc;
    }
  }
  double *w2;
  if ((rc = get_scratch(c, 8, ld, &w2))) return rc;
  if ((rc = factor_solve(c, c->implicit_f, ld, w1, w2))) return rc;                // w2[trans] = M^-1 B^T X
  HIP_TRY(c, launch_axpby(static_cast<int64_t>(L.nl_trans) * ld, -1.0, w2 + toff, 0.0, w0 + toff, c->stream));
  if (mine) HIP_TRY(c, launch_zero_row(w0, static_cast<size_t>(last), ld, c->stream));  // w0 = [X; t; 0]
  return CORA_OK;
}

static int implicit_product(cora_ctx *c, const double *dX, int ld, int epi, double *dOut) {
  if (!c->implicit_f.ready)
    return fail(c, CORA_ERR_NOT_READY, "implicit formulation needs cora_implicit_set_cholesky");
  double *w0, *w1;
  int rc;
  if ((rc = get_scratch(c, 3, ld, &w0))) return rc;
  if ((rc = get_scratch(c, 4, ld, &w1))) return rc;
  if ((rc = implicit_lift(c, dX, ld, w0, w1))) return rc;
  SpmmArgs A = spmm_args(c, w0, dOut);
  if (c->F.L.world != 1) {
    if ((rc = exchange_and_product(c, A, ld, epi))) return rc;
  } else {
    HIP_TRY(c, launch_spmm(A, ld, c->F.L.d, epi, c->stream));
  }
  const Layout &L = c->F.L;
  HIP_TRY(c, hipMemsetAsync(dOut + static_cast<size_t>(L.trn_base) * ld, 0,
                            static_cast<size_t>(L.nl_trans) * ld * sizeof(double), c->stream));
  return CORA_OK;
}

// one product in the active formulation
static int apply_product(cora_ctx *c, const double *dX, int ld, int epi, double *dOut) {
  if (c->implicit) return implicit_product(c, dX, ld, epi, dOut);
  return exchange_and_product(c, spmm_args(c, dX, dOut), ld, epi);
}

// true when products of this handle run as two launches around the exchange (exchange_and_product)
static bool product_overlaps_exchange(const cora_ctx *c) {
  static const bool off = std::getenv("CORA_NO_EXCHANGE_OVERLAP") != nullptr;
  // The split costs a second launch and two cross-stream dependencies (a few microseconds); it pays when the interior
  // slices run longer than that.  Measured with the in-process transport, 8 partitions of the 10^5-pose graph on one
  // GPU (500 slices per rank, a 3 us product): 290-350 us per step serial, 460 us split -- so the default takes the
  // split from kOverlapMinSlices interior slices on (about 10 us of product; 10^6 poses on 8 GPUs: 5 000 per rank).
  static const int min_slices = [] { const char *e = std::getenv("CORA_EXCHANGE_OVERLAP_MIN_SLICES"); return e ? std::atoi(e) : 2048; }();
  const bool wanted = c->overlap_exchange == 2 || (c->overlap_exchange == 1 && c->n_slices_int >= min_slices);
  return c->F.L.world > 1 && wanted && !off && c->native_comm && c->comm_user == c->native_comm &&
         c->comm_exchange != nullptr && c->n_slices_int > 0 && c->n_slices_bnd > 0 &&
         (c->F.chunks.empty() || !c->F.long_rows.empty());  // (whole long rows read columns of every shard)
}

// true when a product of this handle is ONE collective: the library's own communication and distributed long rows -- the
// chunks of the long rows run first (they read columns of this shard only), their partial sums travel with the exported
// rows of the operand, and every owner adds them up in rank order.  (Injected callbacks: exchange, product, all-reduce
// of the slots, as before.)
static bool product_one_collective(const cora_ctx *c) {
  return c->F.L.world > 1 && !c->local_products && c->native_comm && c->comm_user == c->native_comm && c->comm_exchange != nullptr &&
         !c->F.long_rows.empty();
}

// number of kappa slots an EPI_HVP_K product of this handle writes (launch_product / exchange_and_product)
static int product_kappa_slots(const cora_ctx *c, const SpmmArgs &A) {
  if (product_one_collective(c)) {  // chunk launch | one slot per long row | slice launch (or interior | boundary)
    SpmmArgs Ac = A, A1 = A, A2 = A;
    Ac.n_slices = 0;
    A1.n_chunks = A2.n_chunks = 0;
    if (!product_overlaps_exchange(c)) return launch_spmm_blocks(Ac) + A.n_long_rows + launch_spmm_blocks(A1);
    A1.n_slices = c->n_slices_int;
    A2.n_slices = c->n_slices_bnd;
    return launch_spmm_blocks(Ac) + A.n_long_rows + launch_spmm_blocks(A1) + launch_spmm_blocks(A2);
  }
  if (!product_overlaps_exchange(c)) return launch_spmm_kappa_slots(A);
  SpmmArgs A1 = A, A2 = A;
  A1.n_slices = c->n_slices_int;
  A2.n_slices = c->n_slices_bnd;
  A2.n_chunks = 0;
  return launch_spmm_blocks(A1) + A.n_long_rows + launch_spmm_blocks(A2);
}

// Exchange of the operand's remote rows + the product.
//   * the library's own communication: pack (+ zeroed slots) -> the long rows' chunks -> ONE all-gather of rows and slots
//     -> unpack (rows into X, slots summed into the owners' rows) -> the slices.  With the overlap on, gather and unpack
//     run on comm_stream while the interior slices -- which read rows of this rank's shard only -- run on the handle's
//     stream, and the boundary slices follow when the exchange has landed.
//   * injected callbacks (cora_set_comm): exchange, product, all-reduce of the long rows' slots, in that order.
static int exchange_and_product(cora_ctx *c, SpmmArgs A, int ld, int epi) {
  if (c->local_products) return launch_product(c, A, ld, epi, /*finish=*/false);  // timing hook: no collective step at all
  if (!product_one_collective(c)) {
    if (!product_overlaps_exchange(c)) {
      const int rc = comm_exchange(c, A.X, ld);
      if (rc) return rc;
      return launch_product(c, A, ld, epi);
    }
    // (overlap without distributed long rows: whole long rows are excluded by product_overlaps_exchange, so this is a
    // handle with no long rows at all)
    HIP_TRY(c, hipEventRecord(c->ev_operand, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->ev_operand, 0));
    SpmmArgs A1 = A;
    A1.slices = c->d_slices_int;
    A1.slices_pose_first = nullptr;
    A1.n_slices = c->n_slices_int;
    int rc = launch_product(c, A1, ld, epi, /*finish=*/false);
    if (rc) return rc;
    if (native_exchange_on(c->native_comm, const_cast<double *>(A.X), ld, c->comm_stream))
      return fail(c, CORA_ERR_HIP, "exchange step failed: " + native_error(c->native_comm));
    HIP_TRY(c, hipEventRecord(c->ev_exchanged, c->comm_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_exchanged, 0));
    SpmmArgs A2 = A;
    A2.slices = c->d_slices_bnd;
    A2.slices_pose_first = nullptr;
    A2.n_slices = c->n_slices_bnd;
    A2.n_chunks = 0;
    A2.n_long_rows = 0;
    if (A.kappa_partial) A2.kappa_partial = A.kappa_partial + launch_spmm_blocks(A1) + A.n_long_rows;
    HIP_TRY(c, launch_spmm(A2, ld, c->F.L.d, epi, c->stream));
    return CORA_OK;
  }
  cora_native_comm *nc = c->native_comm;
  const bool kap = epi == EPI_HVP_K && A.kappa_partial;
  double *slots = nullptr;
  auto mark = [&](int i) {  // (measurement hook: five phases, six events)
    if (c->phase_timing && static_cast<size_t>(i) < c->phase_events.size()) (void)hipEventRecord(c->phase_events[i], c->stream);
  };
  mark(0);
  if (native_product_pack(nc, A.X, ld, c->stream, &slots)) return fail(c, CORA_ERR_HIP, "exchange step failed: " + native_error(nc));
  mark(1);
  SpmmArgs Ac = A;  // the long rows' chunks: partial sums over this rank's columns -> the slots that travel
  Ac.n_slices = 0;
  Ac.slices_pose_first = nullptr;
  Ac.long_out = slots;
  HIP_TRY(c, launch_spmm(Ac, ld, c->F.L.d, epi, c->stream));
  mark(2);
  const int kc = launch_spmm_blocks(Ac);
  double *klong = kap ? A.kappa_partial + kc : nullptr;
  SpmmArgs A1 = A;
  A1.n_chunks = 0;
  A1.n_long_rows = 0;
  if (kap) A1.kappa_partial = A.kappa_partial + kc + A.n_long_rows;
  if (!product_overlaps_exchange(c)) {
    if (native_product_gather(nc, const_cast<double *>(A.X), ld, c->stream, A.out, klong, c->phase_timing ? c->phase_events[3] : nullptr))
      return fail(c, CORA_ERR_HIP, "exchange step failed: " + native_error(nc));
    mark(4);
    HIP_TRY(c, launch_spmm(A1, ld, c->F.L.d, epi, c->stream));
    mark(5);
    return CORA_OK;
  }
  HIP_TRY(c, hipEventRecord(c->ev_operand, c->stream));
  HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->ev_operand, 0));
  if (native_product_gather(nc, const_cast<double *>(A.X), ld, c->comm_stream, A.out, klong))
    return fail(c, CORA_ERR_HIP, "exchange step failed: " + native_error(nc));
  HIP_TRY(c, hipEventRecord(c->ev_exchanged, c->comm_stream));
  SpmmArgs A2 = A1;
  A1.slices = c->d_slices_int;
  A1.slices_pose_first = nullptr;
  A1.n_slices = c->n_slices_int;
  HIP_TRY(c, launch_spmm(A1, ld, c->F.L.d, epi, c->stream));   // interior slices: no row of another rank is read
  HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_exchanged, 0));
  A2.slices = c->d_slices_bnd;
  A2.slices_pose_first = nullptr;
  A2.n_slices = c->n_slices_bnd;
  if (kap) A2.kappa_partial = A1.kappa_partial + launch_spmm_blocks(A1);
  HIP_TRY(c, launch_spmm(A2, ld, c->F.L.d, epi, c->stream));
  return CORA_OK;
}

// A product of a partitioned handle WITHOUT the library's own communication ends with its DISTRIBUTED long rows
// (format_build.cpp): the slots of partial sums are added over the ranks through the injected all-reduce and the owner
// copies its rows to the result; the rows' shares of kappa follow (EPI_HVP_K).
static int launch_product(cora_ctx *c, SpmmArgs A, int ld, int epi, bool finish) {
  const int nl = static_cast<int>(c->F.long_rows.size());
  const bool dist = c->F.L.world > 1 && nl > 0;
  if (dist) {
    A.long_out = c->d_long_out;
    HIP_TRY(c, hipMemsetAsync(c->d_long_out, 0, static_cast<size_t>(nl) * ld * sizeof(double), c->stream));
  }
  HIP_TRY(c, launch_spmm(A, ld, c->F.L.d, epi, c->stream));
  return finish ? finish_long_rows(c, A, ld, epi) : CORA_OK;
}

// (A: the arguments of the launch that ran the long-row chunks -- its block count places the rows' kappa slots)
static int finish_long_rows(cora_ctx *c, const SpmmArgs &A, int ld, int epi) {
  const int nl = static_cast<int>(c->F.long_rows.size());
  const bool dist = c->F.L.world > 1 && nl > 0;
  if (!dist) return CORA_OK;
  const int n = nl * ld;
  if (c->native_comm && c->comm_user == c->native_comm) {
    if (native_allreduce_dev(c->native_comm, c->d_long_out, n))
      return fail(c, CORA_ERR_HIP, "all-reduce step failed: " + native_error(c->native_comm));
  } else if (c->comm_allreduce) {
    std::vector<double> h(static_cast<size_t>(n));
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_long_out, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int at = 0; at < n; at += 512) {  // "a few doubles" per call: the callbacks' contract
      const int rc = comm_allreduce(c, h.data() + at, std::min(512, n - at));
      if (rc) return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_long_out, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    // nobody can add the slots up: the landmark rows of the result would silently be PARTIAL sums (round-3 advice).
    // A handle that is to run without communication keeps whole long rows: cora_ctx_create_part_opts(...,
    // CORA_PART_WHOLE_LONG_ROWS).
    return fail(c, CORA_ERR_NOT_READY,
                "partitioned handle with distributed long rows and no communication: install cora_set_comm / cora_comm_create_*, "
                "or create the handle with CORA_PART_WHOLE_LONG_ROWS");
  }
  const int kbase = launch_spmm_blocks(A);
  HIP_TRY(c, launch_long_finish(nl, ld, c->F.L.rank, c->d_long_rows, c->d_long_owner, c->d_long_out, A.X, A.out,
                                (epi == EPI_HVP_K && A.kappa_partial) ? A.kappa_partial + kbase : nullptr, c->stream));
  return CORA_OK;
}

int cora_implicit_set_cholesky(cora_ctx *c, int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                               const int32_t *perm) {
  NEED_DEVICE(c);
  const Layout &L = c->F.L;
  if (!Lp || !Li || !Lx || !perm || m != L.nt - 1) return fail(c, CORA_ERR_ARG, "factor must have n + l - 1 rows");
  const int64_t tb = static_cast<int64_t>(L.d) * L.n + L.r;
  std::vector<int32_t> row_of(static_cast<size_t>(m));
  std::vector<char> seen(static_cast<size_t>(L.nt), 0);
  for (int i = 0; i < m; ++i) {
    if (perm[i] < 0 || perm[i] >= L.nt - 1 || seen[perm[i]]) return fail(c, CORA_ERR_ARG, "perm is not a permutation");
    seen[perm[i]] = 1;
    row_of[i] = c->F.api2int[tb + perm[i]];
  }
  return install_factor(c, c->implicit_f, m, Lp, Li, Lx, row_of, -1);
}

int cora_aux_set_cholesky(cora_ctx *c, int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                          const int32_t *perm) {
  NEED_DEVICE(c);
  const Layout &Lo = c->F.L;
  const int64_t N = Lo.N;
  // Partitioned handle: the factor of THIS RANK'S rows (block Jacobi over the ranks, like cora_precond_set_cholesky);
  // the solve then touches the rank's own rows of a vector and nothing else.
  const bool sharded = Lo.world != 1;
  const int64_t owned = sharded ? Lo.local_rows : N;
  if (!Lp || !Li || !Lx || !perm || m != owned)
    return fail(c, CORA_ERR_ARG, sharded ? "factor must cover the rank's own rows" : "factor must have N rows");
  std::vector<int32_t> row_of(static_cast<size_t>(m));
  std::vector<char> seen(static_cast<size_t>(N), 0);
  for (int i = 0; i < m; ++i) {
    if (perm[i] < 0 || perm[i] >= N || seen[perm[i]]) return fail(c, CORA_ERR_ARG, "perm is not a permutation");
    seen[perm[i]] = 1;
    row_of[i] = c->F.api2int[perm[i]];
    if (sharded && (row_of[i] < Lo.base || row_of[i] >= Lo.base + Lo.shard_rows))
      return fail(c, CORA_ERR_ARG, "the factor of a partitioned handle may only hold rows of its own shard");
  }
  return install_factor(c, c->aux_f, m, Lp, Li, Lx, row_of, -1);
}

int cora_aux_solve_dev(cora_ctx *c, const double *dB, int k, double *dX) {
  NEED_DEVICE(c);
  wrote(c, dX);
  if (!c->aux_f.ready) return fail(c, CORA_ERR_NOT_READY, "no factor installed (cora_aux_set_cholesky)");
  if (!dB || !dX || k <= 0 || k > kMaxLD || dB == dX) return fail(c, CORA_ERR_ARG, "bad arguments");
  return factor_solve(c, c->aux_f, ld_for(k), dB, dX);
}

int cora_set_formulation(cora_ctx *c, int implicit) {
  if (!c) return CORA_ERR_ARG;
  if (implicit && !c->implicit_f.ready)
    return fail(c, CORA_ERR_NOT_READY, "implicit formulation needs cora_implicit_set_cholesky");
  if (c->implicit != (implicit != 0)) {
    c->implicit = implicit != 0;
    c->have_point = false;  // cached QY / Lambda belong to the other operator
    c->trial_x = nullptr;
  }
  return CORA_OK;
}

int cora_translation_explicit_dev(cora_ctx *c, const double *dY, int k, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  if (!dY || !dOut || k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_ARG, "bad arguments");
  if (!c->implicit_f.ready)
    return fail(c, CORA_ERR_NOT_READY, "implicit formulation needs cora_implicit_set_cholesky");
  double *w1;
  int rc;
  if ((rc = get_scratch(c, 4, ld_for(k), &w1))) return rc;
  if (dOut == dY) return fail(c, CORA_ERR_ARG, "output aliases the input");
  return implicit_lift(c, dY, ld_for(k), dOut, w1);  // dOut = [Y; -M^-1 B^T Y; 0]
}

int cora_precondition_projected_dev(cora_ctx *c, const double *dV, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  NEED_RANK(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  const double *scale = nullptr;
  if (c->precond == CORA_PRECOND_JACOBI) scale = c->d_diag_inv;
  else if (c->precond == CORA_PRECOND_BLOCK_CHOLESKY || c->precond == CORA_PRECOND_REGULARIZED_CHOLESKY) {
    const double *rhs = dV;
    if (c->implicit || dOut == dV) {  // V_lift = [V; 0] (src/CORA_problem.cpp:878-884), or an in-place call
      double *tmp;
      int rc = get_scratch(c, 8, c->ld, &tmp);
      if (rc) return rc;
      HIP_TRY(c, hipMemcpyAsync(tmp, dV, vec_bytes(c, c->ld), hipMemcpyDeviceToDevice, c->stream));
      if (c->implicit)
        HIP_TRY(c, hipMemsetAsync(tmp + static_cast<size_t>(c->F.L.trn_base) * c->ld, 0,
                                  static_cast<size_t>(c->F.L.nl_trans) * c->ld * sizeof(double), c->stream));
      rhs = tmp;
    }
    bool projected = false;
    int rc = factor_solve(c, c->precond_f, c->ld, rhs, dOut, c->implicit ? nullptr : &projected);
    if (rc) return rc;
    if (c->implicit)
      HIP_TRY(c, hipMemsetAsync(dOut + static_cast<size_t>(c->F.L.trn_base) * c->ld, 0,
                                static_cast<size_t>(c->F.L.nl_trans) * c->ld * sizeof(double), c->stream));
    if (!projected) HIP_TRY(c, launch_tangent_project(row_args(c), c->ld, c->d_Y, dOut, nullptr, dOut, c->stream));
    return CORA_OK;
  } else if (c->precond != CORA_PRECOND_NONE) return fail(c, CORA_ERR_NOT_READY, "preconditioner not set up");
  HIP_TRY(c, launch_tangent_project(row_args(c), c->ld, c->d_Y, dV, scale, dOut, c->stream));
  return CORA_OK;
}

int cora_retract_dev(cora_ctx *c, const double *dV, double alpha, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  NEED_RANK(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  HIP_TRY(c, launch_project_manifold(row_args(c), c->ld, c->d_Y, dV, alpha, dOut, c->stream));
  return CORA_OK;
}

int cora_project_to_manifold_dev(cora_ctx *c, const double *dA, double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  NEED_RANK(c);
  HIP_TRY(c, launch_project_manifold(row_args(c), c->ld, dA, nullptr, 0.0, dOut, c->stream));
  return CORA_OK;
}

int cora_axpby_dev(cora_ctx *c, double a, const double *dX, double b, double *dY) {
  NEED_DEVICE(c);
  wrote(c, dY);
  NEED_RANK(c);
  const size_t off = static_cast<size_t>(c->F.L.base) * c->ld;
  HIP_TRY(c, launch_axpby(c->F.L.local_rows * c->ld, a, dX + off, b, dY + off, c->stream));
  return CORA_OK;
}

int cora_axpy2_dev(cora_ctx *c, double a1, const double *dX1, double *dY1, double a2, const double *dX2,
                   double *dY2) {
  NEED_DEVICE(c);
  wrote(c, dY1);
  wrote(c, dY2);
  NEED_RANK(c);
  if (!dX1 || !dY1 || !dX2 || !dY2 || dY1 == dY2) return fail(c, CORA_ERR_ARG, "bad arguments");
  const size_t off = static_cast<size_t>(c->F.L.base) * c->ld;
  HIP_TRY(c, launch_axpy2(c->F.L.local_rows * c->ld, a1, dX1 + off, dY1 + off, a2, dX2 + off, dY2 + off, c->stream));
  return CORA_OK;
}

int cora_fill_random_dev(cora_ctx *c, int k, unsigned long long seed, double *dX) {
  NEED_DEVICE(c);
  wrote(c, dX);
  if (k <= 0 || k > kMaxLD || !dX) return fail(c, CORA_ERR_ARG, "bad arguments");
  HIP_TRY(c, launch_fill_random(c->F.L.N, k, seed, c->d_api2int, dX, c->stream));
  return CORA_OK;
}

int cora_axpby_cols_dev(cora_ctx *c, int k, double a, const double *dX, double b, double *dY) {
  NEED_DEVICE(c);
  wrote(c, dY);
  if (k <= 0 || k > kMaxLD || !dX || !dY) return fail(c, CORA_ERR_ARG, "bad arguments");
  const int ld = ld_for(k);
  const size_t off = static_cast<size_t>(c->F.L.base) * ld;
  HIP_TRY(c, launch_axpby(c->F.L.local_rows * ld, a, dX + off, b, dY + off, c->stream));
  return CORA_OK;
}

int cora_copy_dev(cora_ctx *c, const double *dX, int k, double *dY) {
  NEED_DEVICE(c);
  wrote(c, dY);
  HIP_TRY(c, hipMemcpyAsync(dY, dX, vec_bytes(c, ld_for(k)), hipMemcpyDeviceToDevice, c->stream));
  return CORA_OK;
}

int cora_dots_dev(cora_ctx *c, int count, const double *const *dA, const double *const *dB, double *out) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (count < 1 || count > 4 || !dA || !dB || !out) return fail(c, CORA_ERR_ARG, "bad arguments");
  DotArgs D;
  const size_t off = static_cast<size_t>(c->F.L.base) * c->ld;
  for (int j = 0; j < 4; ++j) {
    D.a[j] = (j < count) ? dA[j] + off : nullptr;
    D.b[j] = (j < count) ? dB[j] + off : nullptr;
  }
  D.count = count;
  D.n2 = c->F.L.local_rows * c->ld;
  int rc = ensure_red(c, 4 * 512);
  if (rc) return rc;
  D.partial = c->d_red;
  D.ticket = c->d_ticket;
  D.out = c->h_scalars;  // pinned: the last block writes the results where the host reads them
  D.seq_out = reinterpret_cast<unsigned long long *>(c->h_scalars + 7);
  D.seq = ++c->dot_seq;
  D.mode = DOTS_PLAIN;
  D.st = D.st_host = nullptr;
  int nblocks = 0;
  HIP_TRY(c, launch_dots(D, &nblocks, c->stream));
  if ((rc = wait_dots(c, D.seq))) return rc;
  for (int j = 0; j < count; ++j) out[j] = c->h_scalars[j];
  return comm_allreduce(c, out, count);
}

// can cora_stpcg_dev run on this handle?  One GPU: always.  Partitioned: with the library's own communication (the
// reductions stay on the device), the explicit formulation, a row-local preconditioner and 16-byte aligned shards.
// The answer is the same on every rank (the two loops make different collective calls): nothing in it depends on the
// rank's own rows.  The flat vector passes of a partitioned handle run over the PADDED shard (shard_rows is a multiple
// of 8, so the range is even and 64-byte aligned whatever the row stride); padding rows are zero in every resident
// vector -- allocations are zeroed and no kernel writes them -- and add nothing to an update or an inner product.
static bool stpcg_device_ok(const cora_ctx *c) {
  if (c->F.L.world == 1) return true;
  // the library's own communication must be the ACTIVE transport (after cora_comm_native_enable(0) or a later
  // cora_set_comm the reductions would go one way and the operand's exchange another -- round-3 advice).  The answer must
  // not depend on anything rank-local: the Cholesky case is decided by the preconditioner's KIND (a missing factor fails
  // loudly in the solve, on every rank alike).
  return c->native_comm && c->comm_user == c->native_comm && c->comm_exchange != nullptr && !c->implicit && c->ld <= 12;
}

// Steihaug-Toint truncated PCG for  min <g,s> + 1/2 <s,Hs>,  ||s||_M <= Delta, entirely on the device
// (the inner solver of Optimization::Riemannian::TNT, called from src/CORA.cpp:139-140).  The scalar
// recurrences live in a StpcgState that the inner-product kernels update themselves, so the host only
// enqueues iterations -- a few at a time -- and looks at the state's pinned mirror between batches.
// dPg != nullptr: the caller already holds P g and the inner products <g, g>, <g, P g> (TNT computes them for its
// stopping tests): the solve starts without a preconditioner apply and without a reduction of its own.
static int stpcg_run(cora_ctx *c, const double *dGrad, const double *dPg, double gg, double gPg, double Delta,
                     double kappa_fgr, double theta, int max_iters, double *dS, double *dR, double *dV, double *dP,
                     double *dHp, int *iters, double *step_M_norm) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  if (!dGrad || !dS || !dR || !dV || !dP || !dHp || !iters || !step_M_norm || max_iters < 0)
    return fail(c, CORA_ERR_ARG, "bad arguments");
  for (const double *w : {dS, dR, dV, dP, dHp}) wrote(c, w);
  // Partitioned handle: the same fused iteration, every rank on its own rows, with the library's own communication
  // (cora_comm_create_*): the operand's remote rows are exchanged before the product, and the three inner products are
  // summed over the ranks ON THE DEVICE -- kappa after the product, <r, r> and <r, v> together after the projection --
  // by an all-reduce on the handle's stream, followed by a one-thread launch for the scalar step.  The host enqueues
  // and looks at the pinned mirror between batches, exactly as on one GPU.
  const bool sharded = c->F.L.world != 1;
  if (sharded && !stpcg_device_ok(c))
    return fail(c, CORA_ERR_ARG, "the device-resident STPCG on a partitioned handle needs cora_comm_create_* and the explicit formulation");
  int rc;
  double rr_rv[2];
  if (dPg) {  // s = 0, r = g, p = -P g in one pass
    const size_t off0 = static_cast<size_t>(c->F.L.base) * c->ld;
    HIP_TRY(c, launch_stpcg_init(c->F.L.local_rows * c->ld, dGrad + off0, dPg + off0, dS + off0, dR + off0, dP + off0,
                                 c->stream));
    rr_rv[0] = gg;
    rr_rv[1] = gPg;
  } else {  // s = 0, r = g, v = P r, p = -v
    if ((rc = cora_axpby_dev(c, 0.0, dGrad, 0.0, dS))) return rc;
    if ((rc = cora_axpby_dev(c, 1.0, dGrad, 0.0, dR))) return rc;
    if ((rc = cora_precondition_projected_dev(c, dR, dV))) return rc;
    if ((rc = cora_axpby_dev(c, -1.0, dV, 0.0, dP))) return rc;
    const double *A[2] = {dR, dR};
    const double *B[2] = {dR, dV};
    if ((rc = cora_dots_dev(c, 2, A, B, rr_rv))) return rc;
  }
  const double r0 = std::sqrt(rr_rv[0]);
  if (c->stpcg_pending_seq) {  // the last solve's neutral iteration writes the mirror too: it must be behind us before
    if ((rc = wait_dots(c, c->stpcg_pending_seq))) return rc;  // the mirror is reset (long finished by now: no wait)
    c->stpcg_pending_seq = 0;
  }
  StpcgState &H = c->h_stpcg[1];  // staging copy for the upload; h_stpcg[0] is the mirror the kernels write
  H = StpcgState();
  H.r_v = rr_rv[1];
  H.p_M2 = rr_rv[1];
  H.Delta2 = Delta * Delta;
  H.target = r0 * std::min(kappa_fgr, std::pow(r0, theta));
  H.coef_beta = 1.0;
  H.max_iters = max_iters;
  c->h_stpcg[0] = H;
  HIP_TRY(c, hipMemcpyAsync(c->d_stpcg, &H, sizeof(StpcgState), hipMemcpyHostToDevice, c->stream));
  const int64_t n = (sharded ? c->F.L.shard_rows : c->F.L.local_rows) * c->ld;  // sharded: the padded shard (stpcg_device_ok)
  DotArgs D;
  for (int j = 0; j < 4; ++j) D.a[j] = D.b[j] = nullptr;
  D.n2 = n;
  D.count = 1;
  D.mode = DOTS_PLAIN;
  D.seq_out = nullptr;
  D.seq = 0;
  if ((rc = ensure_red(c, 4 * 512))) return rc;
  D.partial = c->d_red;
  D.ticket = c->d_ticket;
  D.out = c->d_scalars;
  D.st = c->d_stpcg;
  D.st_host = &c->h_stpcg[0];
  // How far the host runs ahead of the state it has seen.  One GPU: ONE iteration on small problems (`depth`, below: the
  // host waits for iteration k - 1 before it enqueues k + 1, so the GPU never waits for the host and exactly one
  // iteration is enqueued past the stopping point, neutralised by the state), none on large ones (the reductions' block
  // runs in the launch BEFORE the backward sweep, which covers the host's reaction).  Partitioned handles and graph replay:
  // `batch` iterations between two looks, everything waited for (every rank must enqueue the same collective calls, so
  // the decision may only depend on a state that no iteration in flight can have advanced).
  // (the three switches of the host loop are read per solve: tests/test_gpu_solver.py runs one problem under each form)
  const int batch_env = [] { const char *e = std::getenv("CORA_STPCG_BATCH"); return e ? std::atoi(e) : 0; }();
  const int batch = batch_env > 0 ? batch_env : (n > 1000000 ? 1 : 4);
  int enqueued = 0;
  // Fused iteration (explicit formulation, one shard, row strides up to 12): six passes instead of nine --
  //   Hp = H p | kappa = <p, Hp> | r += alpha Hp with <r, r> | Cholesky solve | v = Proj_Y(x) with <r, v> |
  //   s += alpha p, p = -v + beta p
  // The scalar steps run in the last block of the pass that finishes the inner product they need.
  const bool chol = c->precond == CORA_PRECOND_BLOCK_CHOLESKY || c->precond == CORA_PRECOND_REGULARIZED_CHOLESKY;
  const size_t off = static_cast<size_t>(c->F.L.base) * c->ld;
  const bool fused = sharded || (!c->implicit && c->ld <= 12 && n % 2 == 0 && (off * sizeof(double)) % 16 == 0 &&
                                 !std::getenv("CORA_NO_FUSE"));
  // Sweep-fused iteration (the above, with a two-stage Cholesky solve plan): five passes and a scalar step --
  //   Hp = H p with the partials of kappa | kappa | forward sweep on r += alpha Hp with <r, r> | last stage (2 products) |
  //   backward sweep with v = Proj_Y(x) and <r, v> | s += alpha p, p = -v + beta p
  bool sweep_fused = false, inverse_fused = false;
  SubFuse FF, FB;
  double *kappa_partial = nullptr;
  int kappa_blocks = 0;
  RvTail tail{}, sq{};
  if (fused) {
    const RowArgs R = row_args(c);
    const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
    size_t need = std::max<size_t>(4 * 512, static_cast<size_t>((units + 255) / 256) + 8);
    const cora_ctx::DevFactor &f = c->precond_f;
    sweep_fused = !c->implicit && chol && f.ready && f.fuse_ok && !f.stages.empty() && f.stages[0].is_sub && c->ld * c->F.L.d <= 24 && c->ld <= 11 &&
                  !std::getenv("CORA_NO_SWEEP_FUSE");  // (row stride x d > 24: the fused backward sweep spills)
    // slots of the sweep-fused reductions: <r, r> per block of the forward sweep's launch, |y|^2 per solve block,
    // |row|^2 per row of the last stage's forward product
    size_t rr_slots = 0, yy_slots = 0, sq_slots = 0;
    if (sweep_fused) {
      rr_slots = static_cast<size_t>(launch_subblock_blocks(f.stages[0].sub)) + 8;
      yy_slots = static_cast<size_t>(f.stages[0].sub.nblocks) + 8;
      const RowOpDev &fb = f.stages[1].fwd_b;
      sq_slots = static_cast<size_t>(rowop_rowsq_slots(fb)) + 8;
    }
    // one explicit inverse W = L^-1 and nothing else (two products per solve), no pinned-row stage in between
    inverse_fused = !sharded && !sweep_fused && chol && f.ready && f.stages.size() == 1 && !f.stages[0].dense && !f.stages[0].is_sub &&
                    !f.stages[0].has_fwd_a && !f.stages[0].has_bwd_a && !std::getenv("CORA_NO_INVERSE_FUSE");
    static const bool residual_slots = !std::getenv("CORA_NO_RESIDUAL_SLOTS");
    if (inverse_fused) {
      const RowOpDev &fb = f.stages[0].fwd_b;
      sq_slots = static_cast<size_t>(rowop_rowsq_slots(fb)) + 8;
      if (residual_slots) rr_slots = static_cast<size_t>(kappa_residual_slots_blocks(n)) + 8;
    }
    kappa_blocks = product_kappa_slots(c, spmm_args(c, dP, dHp));
    if ((rc = ensure_red(c, need + static_cast<size_t>(kappa_blocks) + rr_slots + yy_slots + sq_slots))) return rc;
    D.partial = c->d_red;
    kappa_partial = c->d_red + need;
    if (inverse_fused) {
      double *rr_partial = kappa_partial + kappa_blocks, *rowsq = rr_partial + rr_slots;
      sq.rowsq_out = rowsq;
      // kappa and <r, r> are finished by the tail block too (k_kappa_residual_slots): the residual pass has no ticket and
      // no last block, and the iteration's three scalar steps run in one place.  (CORA_NO_RESIDUAL_SLOTS: the residual
      // pass finishes both itself, n_rr = n_kappa = 0 -- the form measured against in profiles/r05_kernel_evolution.md)
      tail.rr_partial = rr_partial;
      tail.n_rr = residual_slots ? kappa_residual_slots_blocks(n) : 0;
      if (residual_slots) {
        tail.kappa_partial = kappa_partial;
        tail.n_kappa = kappa_blocks;
      }
      tail.yy_partial = rowsq;
      tail.n_yy = 0;
      tail.rowsq = rowsq;
      tail.n_rowsq = static_cast<int>(sq_slots) - 8;
      tail.st = c->d_stpcg;
      tail.st_host = &c->h_stpcg[0];
      tail.seq_out = reinterpret_cast<unsigned long long *>(c->h_scalars + 7);
    }
    if (sweep_fused) {
      const Layout &L = c->F.L;
      double *rr_partial = kappa_partial + kappa_blocks, *yy_partial = rr_partial + rr_slots, *rowsq = yy_partial + yy_slots;
      FF.dot = D;
      FF.Hp = dHp;
      FF.r = dR;
      FF.d = L.d;
      FF.rot_base = L.rot_base;
      FF.rng_base = L.rng_base;
      FF.trn_base = L.trn_base;
      FF.rr_partial = rr_partial;
      FF.yy_partial = yy_partial;
      FB = FF;
      FB.Y = c->d_Y;
      FB.p = dP;
      FB.s = dS;
      // the two reductions of the iteration are finished by an extra block of the last stage's SECOND product:
      // <r, r> from the forward sweep's slots, <r, v> = |L^-1 r|^2 from its |y|^2 slots + the squared norms of the rows
      // of t_1, which the last stage's FIRST product leaves (sq)
      sq.rowsq_out = rowsq;
      tail.rr_partial = rr_partial;
      tail.n_rr = launch_subblock_blocks(f.stages[0].sub);
      tail.yy_partial = yy_partial;
      tail.n_yy = f.stages[0].sub.nblocks;
      tail.rowsq = rowsq;
      tail.n_rowsq = static_cast<int>(sq_slots) - 8;
      tail.st = c->d_stpcg;
      tail.st_host = &c->h_stpcg[0];
      tail.seq_out = reinterpret_cast<unsigned long long *>(c->h_scalars + 7);
      // kappa without a launch of its own (one GPU, mid-size problems): every block of the forward sweep adds the
      // product's partials and runs the scalar step privately, the tail block of the last stage advances the state.
      // Measured: with a few hundred partials (the reference's data sets: 288) an iteration loses the 4.5 us launch and
      // the sweep does not notice (plaza1 69.0 -> 66.6 us per product end to end, tiers 88.5 -> 84.8, mrclam6 109 -> 104.5);
      // with the 2 470 partials of 10^5 poses every block's sum costs the sweep the 5.2 us the launch took (1 060 blocks
      // reading the same 20 KB through eight L2s): there, and above, the launch stays.
      // (round 5: a solve block adds the partials BEHIND the loads of its right-hand sides -- kernels.hip, late_kappa --, which
      // moved the break-even up: at 10^5 poses the iteration goes from 111.8 to 110.6 us without the launch; 10^6 poses, 24 k
      // partials, keep it)
      static const int fold_max = [] { const char *e = std::getenv("CORA_KAPPA_FOLD_MAX"); return e ? std::atoi(e) : 4096; }();
      if (!sharded && kappa_blocks <= fold_max && !std::getenv("CORA_NO_KAPPA_FOLD")) {
        FF.kappa_partial = tail.kappa_partial = kappa_partial;
        FF.n_kappa = tail.n_kappa = kappa_blocks;
      }
    }
  }
  c->stpcg_path = sweep_fused ? 2 : inverse_fused ? 3 : fused ? 1 : 0;
  c->prof_kappa_folded = sweep_fused && FF.n_kappa > 0;
  // hipGraph replay of whole batches (one GPU, the fused forms): OPT-IN, CORA_STPCG_GRAPH=1.  Measured on this part, replaying the batches does not bring the launches of an iteration
  // closer together -- a dependent kernel of 5 us and more already has its successor's packet waiting, what is left
  // between them is the dependency itself -- and a six-launch graph per iteration costs the host more than six launches:
  // iteration at 10^5 poses 116 -> 122 us, the reference's data sets unchanged.  tools/launch_lab.hip shows the gain only
  // for kernels shorter than the launch rate, 3.5 -> 2.1 us each.  Kept: same bits, tested, one switch.
  const bool graphs_on = [] { const char *e = std::getenv("CORA_STPCG_GRAPH"); return e && e[0] == '1'; }();  // (read per solve: tests flip it)
  const bool use_graph = graphs_on && fused && !sharded && !c->prof_stpcg && max_iters >= batch;
  std::vector<uintptr_t> key;
  if (use_graph) {
    double *t = nullptr, *t2 = nullptr;
    if (sweep_fused && (rc = get_scratch(c, 6, c->ld, &t, c->precond_f.aux_rows))) return rc;  // (allocations happen here,
    if ((sweep_fused || inverse_fused) && (rc = get_scratch(c, 7, c->ld, &t2))) return rc;      // not under capture)
    if (chol && !sweep_fused && !inverse_fused) {  // the general solve allocates its own scratch on first use: one warm call
      if ((rc = chol_solve(c, c->ld, dR, dV))) return rc;
    }
    auto U = [](const void *q) { return reinterpret_cast<uintptr_t>(q); };
    key = {static_cast<uintptr_t>(c->stpcg_path), static_cast<uintptr_t>(c->ld), static_cast<uintptr_t>(batch), static_cast<uintptr_t>(n),
           U(dS), U(dR), U(dV), U(dP), U(dHp), U(c->d_red), static_cast<uintptr_t>(kappa_blocks), U(c->d_Y), U(c->d_lam_st), U(t), U(t2),
           static_cast<uintptr_t>(c->precond), static_cast<uintptr_t>(c->precond_f.generation), U(c->stream),
           static_cast<uintptr_t>(c->F.slices.size())};
    for (int i = 0; i < kScratchSlots; ++i) key.push_back(U(c->scratch[i]));  // (whatever a solve in the batch borrows)
    // the device's sequence counter = the host's count (launches of this solve take their numbers from it)
    unsigned long long *stage = reinterpret_cast<unsigned long long *>(c->h_scalars + 6);
    *stage = c->dot_seq;
    HIP_TRY(c, hipMemcpyAsync(c->d_seq_counter, stage, sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    D.seq_counter = c->d_seq_counter;
    tail.seq_counter = c->d_seq_counter;
    FF.dot.seq_counter = FB.dot.seq_counter = c->d_seq_counter;
  }
  const int depth_env = [] { const char *e = std::getenv("CORA_STPCG_DEPTH"); return e ? std::atoi(e) : -1; }();
  const bool pipelined = !sharded && !use_graph && batch_env <= 0;
  const int depth = depth_env >= 0 ? depth_env : (batch == 1 ? 0 : 1);
  // Small problems, the default: the host runs ahead by the next iteration's PRODUCT only.  The reductions' block of
  // iteration k runs in its second-to-last launch (or finishes it: the unfused forms); behind it the GPU still has the
  // last launch of k and the product of k + 1, which is what the host needs to see the state and enqueue the rest of
  // k + 1 -- the GPU does not wait for the host, and past the stopping point there is one product (its results are never
  // read), not one whole neutral iteration.  CORA_STPCG_DEPTH=1 is the whole-iteration form.
  const int ahead_env = [] { const char *e = std::getenv("CORA_STPCG_AHEAD"); return e ? std::atoi(e) : -1; }();  // (lab)
  const bool product_ahead = pipelined && !c->prof_stpcg && (ahead_env >= 0 ? ahead_env != 0 : (depth == 1 && depth_env < 0));
  bool have_product = false;
  std::deque<unsigned long long> in_flight;
  while (c->h_stpcg[0].status == 0 && enqueued < max_iters) {
    unsigned long long seq = 0;
    const bool whole = use_graph && max_iters - enqueued >= batch;
    if (whole && c->stpcg_graph && key == c->stpcg_graph_key) {  // replay: `batch` iterations in one call
      HIP_TRY(c, hipGraphLaunch(c->stpcg_graph, c->stream));
      enqueued += batch;
      seq = (c->dot_seq += static_cast<unsigned long long>(batch));
      ++c->stpcg_graph_replays;
      if ((rc = wait_dots(c, seq))) return rc;
      continue;
    }
    const bool capture = whole;
    if (capture) {
      if (c->stpcg_graph) (void)hipGraphExecDestroy(c->stpcg_graph);
      c->stpcg_graph = nullptr;
      c->stpcg_graph_key.clear();
      HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    }
    auto end_capture = [&](bool ok) -> int {  // closes the capture (always) and, when the batch was recorded whole, runs it
      if (!capture) return CORA_OK;
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (e != hipSuccess || !g) return fail(c, CORA_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
      if (!ok) { (void)hipGraphDestroy(g); return CORA_OK; }
      const hipError_t ei = hipGraphInstantiate(&c->stpcg_graph, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ei != hipSuccess) { c->stpcg_graph = nullptr; return fail(c, CORA_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
      c->stpcg_graph_key = key;
      ++c->stpcg_graph_captures;
      HIP_TRY(c, hipGraphLaunch(c->stpcg_graph, c->stream));
      return CORA_OK;
    };
    // (the launches of one iteration; under capture an error must still close the capture: the caller does)
    // part: 0 the whole iteration | 1 its product alone (Hp = H p, with the partials of kappa) | 2 everything after the product
    auto one_iteration = [&](int part = 0) -> int {
      // measurement hook (cora_debug_profile_stpcg): mark 0 before the product, 1 after it; mode 2 also after every
      // other launch of the sweep-fused form -- 2 kappa | 3 forward sweep | 4, 5 the last stage's two products | 6
      // backward sweep (a mark is an event on the handle's stream: it does not reorder anything)
      const bool prof = c->prof_stpcg && kProfMarks * (static_cast<size_t>(enqueued) + 1) <= c->prof_events.size();
      auto mark = [&](int i) {
        // (mode 1: the product's two marks and the two marks in a row that measure what a mark costs)
        if (prof && (i <= 1 || i >= 6 || c->prof_stpcg >= 2)) (void)hipEventRecord(c->prof_events[kProfMarks * enqueued + i], c->stream);
      };
      mark(0);
      if (fused) {
        // Hp = H p with the partials of kappa | kappa, alpha, r += alpha Hp with <r, r> | preconditioner | ...
        if (part != 2) {
          SpmmArgs A = spmm_args(c, dP, dHp);
          A.kappa_partial = kappa_partial;
          if ((rc = exchange_and_product(c, A, c->ld, EPI_HVP_K))) return rc;  // (one rank: the product alone)
        }
        if (part == 1) return CORA_OK;
        mark(1);
        if (sharded) {
          // kappa: local partials (fixed order) -> sum over the ranks -> scalar step;  then r += alpha Hp with <r, r>
          // and v = Proj_Y(D^-1 r) with <r, v>, both left on the device, one all-reduce for the two, scalar step
          double *ds = native_scalars(c->native_comm);
          HIP_TRY(c, launch_reduce_partials(kappa_partial, kappa_blocks, 1, ds, c->stream));
          if (native_allreduce_dev(c->native_comm, ds, 1)) return fail(c, CORA_ERR_HIP, "all-reduce step failed: " + native_error(c->native_comm));
          HIP_TRY(c, launch_stpcg_scalar_step(0, ds, c->d_stpcg, nullptr, nullptr, 0, c->stream));
          if (sweep_fused) {
            // the sweep-fused form on this rank's block-Jacobi factor: forward sweep (r += alpha Hp, slots of <r, r> and
            // |y|^2) | last stage, whose tail block leaves this rank's <r, r> and <r, v> = |L_k^-1 r_k|^2 | ONE all-reduce
            // for the two | scalar step | backward sweep (v = Proj_Y(x), s += alpha p, p = -v + beta p).  Three vector
            // passes fewer than the form below, the same two all-reduces per iteration.
            cora_ctx::DevFactor &f = c->precond_f;
            double *t, *t2;
            if ((rc = get_scratch(c, 6, c->ld, &t, f.aux_rows))) return rc;
            if ((rc = get_scratch(c, 7, c->ld, &t2))) return rc;
            const cora_ctx::DevStage &S0 = f.stages[0], &S1 = f.stages[1];
            HIP_TRY(c, launch_subblock_fused(S0.sub, c->ld, false, FF, t, dV, c->stream));
            if (S1.aux_sum) HIP_TRY(c, launch_rowop(S1.fwd_a, c->ld, t, t, t, c->stream));
            HIP_TRY(c, launch_rowop(S1.fwd_b, c->ld, nullptr, t, t2, c->stream, &sq));
            tail.sums_out = ds + 2;
            HIP_TRY(c, launch_rowop(S1.bwd_b, c->ld, nullptr, t2, t, c->stream, &tail));
            if (native_allreduce_dev(c->native_comm, ds + 2, 2)) return fail(c, CORA_ERR_HIP, "all-reduce step failed: " + native_error(c->native_comm));
            seq = ++c->dot_seq;
            HIP_TRY(c, launch_stpcg_scalar_step(1, ds + 2, c->d_stpcg, &c->h_stpcg[0],
                                                reinterpret_cast<unsigned long long *>(c->h_scalars + 7), seq, c->stream));
            HIP_TRY(c, launch_subblock_fused(S0.sub, c->ld, true, FB, t, dV, c->stream));
            return CORA_OK;
          }
          DotArgs Ds = D;
          Ds.mode = DOTS_PLAIN;
          Ds.count = 1;
          Ds.seq_out = nullptr;
          Ds.seq = 0;
          Ds.out = ds + 2;
          HIP_TRY(c, launch_stpcg_residual(Ds, n, dHp + off, dR + off, c->stream));
          Ds.out = ds + 3;
          const double *xs = dR;  // what is projected: r (none), D^-1 r (Jacobi, scaled in the pass), or the rank's own
          if (chol) {             // Cholesky solve of its diagonal block (block Jacobi over the ranks)
            if ((rc = chol_solve(c, c->ld, dR, dV))) return rc;
            xs = dV;
          }
          HIP_TRY(c, launch_tangent_project_dot(row_args(c), Ds, c->ld, c->d_Y, xs, c->precond == CORA_PRECOND_JACOBI ? c->d_diag_inv : nullptr,
                                                dR, dV, c->stream));
          if (native_allreduce_dev(c->native_comm, ds + 2, 2)) return fail(c, CORA_ERR_HIP, "all-reduce step failed: " + native_error(c->native_comm));
          seq = ++c->dot_seq;
          HIP_TRY(c, launch_stpcg_scalar_step(1, ds + 2, c->d_stpcg, &c->h_stpcg[0],
                                              reinterpret_cast<unsigned long long *>(c->h_scalars + 7), seq, c->stream));
          HIP_TRY(c, launch_stpcg_step_direction(n, c->d_stpcg, dV + off, dP + off, dS + off, c->stream));
          return CORA_OK;
        }
        if (sweep_fused) {
          if (FF.n_kappa == 0) HIP_TRY(c, launch_kappa_finish(kappa_partial, kappa_blocks, c->d_stpcg, c->stream));
          mark(2);
          // Hp = H p | kappa | forward sweep: r += alpha Hp, <r, r>, |y|^2 | last stage, <r, v> in its second product |
          // backward sweep: v = Proj_Y(x), s += alpha p, p = -v + beta p   -- six launches
          cora_ctx::DevFactor &f = c->precond_f;
          double *t, *t2;
          if ((rc = get_scratch(c, 6, c->ld, &t, f.aux_rows))) return rc;
          if ((rc = get_scratch(c, 7, c->ld, &t2))) return rc;
          const cora_ctx::DevStage &S0 = f.stages[0], &S1 = f.stages[1];
          HIP_TRY(c, launch_subblock_fused(S0.sub, c->ld, false, FF, t, dV, c->stream));
          mark(3);
          if (S1.aux_sum) HIP_TRY(c, launch_rowop(S1.fwd_a, c->ld, t, t, t, c->stream));
          HIP_TRY(c, launch_rowop(S1.fwd_b, c->ld, nullptr, t, t2, c->stream, &sq));
          mark(4);
          tail.seq = seq = ++c->dot_seq;
          HIP_TRY(c, launch_rowop(S1.bwd_b, c->ld, nullptr, t2, t, c->stream, &tail));
          mark(5);
          HIP_TRY(c, launch_subblock_fused(S0.sub, c->ld, true, FB, t, dV, c->stream));
          mark(6);
          mark(7);  // (two marks with nothing between them: what a mark itself costs the stream)
          return CORA_OK;
        }
        // kappa, the scalar step and r += alpha Hp with <r, r> in ONE launch (every block adds the partials: the plans
        // that come here are small, and a launch is what costs them)
        if (inverse_fused && tail.n_rr > 0)
          HIP_TRY(c, launch_kappa_residual_slots(c->d_stpcg, kappa_partial, kappa_blocks, n, dHp + off, dR + off,
                                                 const_cast<double *>(tail.rr_partial), c->stream));
        else
          HIP_TRY(c, launch_kappa_residual(D, kappa_partial, kappa_blocks, n, dHp + off, dR + off, c->stream));
        if (inverse_fused) {
          // one explicit inverse (every data set of the reference): v = Proj_Y(W^T W r) and <r, v> = |W r|^2 -- the first
          // product leaves the squared norms of its rows, an extra block of the second adds them and runs the scalar step,
          // and the projection consumes v at once (s += alpha p, p = -v + beta p): five launches per iteration
          cora_ctx::DevFactor &f = c->precond_f;
          double *t2;
          if ((rc = get_scratch(c, 7, c->ld, &t2))) return rc;
          const cora_ctx::DevStage &S = f.stages[0];
          HIP_TRY(c, launch_rowop(S.fwd_b, c->ld, nullptr, dR, t2, c->stream, &sq));
          tail.seq = seq = ++c->dot_seq;
          HIP_TRY(c, launch_rowop(S.bwd_b, c->ld, nullptr, t2, dV, c->stream, &tail));
          HIP_TRY(c, launch_tangent_project_update(row_args(c), c->d_stpcg, c->ld, c->d_Y, dV, dP, dS, c->stream));
          return CORA_OK;
        }
        const double *x = dR, *scale = nullptr;
        if (chol) {
          if ((rc = chol_solve(c, c->ld, dR, dV))) return rc;
          x = dV;
        } else if (c->precond == CORA_PRECOND_JACOBI) {
          scale = c->d_diag_inv;
        }
        D.mode = DOTS_STPCG_RV;
        D.count = 1;
        D.seq_out = reinterpret_cast<unsigned long long *>(c->h_scalars + 7);
        D.seq = seq = ++c->dot_seq;
        HIP_TRY(c, launch_tangent_project_dot(row_args(c), D, c->ld, c->d_Y, x, scale, dR, dV, c->stream));
        HIP_TRY(c, launch_stpcg_step_direction(n, c->d_stpcg, dV + off, dP + off, dS + off, c->stream));
        D.seq_out = nullptr;
        D.seq = 0;
        return CORA_OK;
      }
      if (part != 2 && (rc = apply_product(c, dP, c->ld, EPI_HVP, dHp))) return rc;
      if (part == 1) return CORA_OK;
      mark(1);
      int nblocks = 0;
      D.count = 1;
      D.a[0] = dP;
      D.b[0] = dHp;
      D.mode = DOTS_STPCG_KAPPA;
      D.seq_out = nullptr;
      D.seq = 0;
      HIP_TRY(c, launch_dots(D, &nblocks, c->stream));
      HIP_TRY(c, launch_stpcg_update(n, c->d_stpcg, dP, dHp, dS, dR, c->stream));
      if ((rc = cora_precondition_projected_dev(c, dR, dV))) return rc;
      D.count = 2;
      D.a[0] = dR;
      D.b[0] = dR;
      D.a[1] = dR;
      D.b[1] = dV;
      D.mode = DOTS_STPCG_BETA;
      D.seq_out = reinterpret_cast<unsigned long long *>(c->h_scalars + 7);
      D.seq = seq = ++c->dot_seq;
      HIP_TRY(c, launch_dots(D, &nblocks, c->stream));
      HIP_TRY(c, launch_stpcg_direction(n, c->d_stpcg, dV, dP, c->stream));
      return CORA_OK;
    };
    if (product_ahead) {
      if (!have_product && (rc = one_iteration(1))) return rc;
      if ((rc = one_iteration(2))) return rc;
      ++enqueued;
      have_product = enqueued < max_iters;
      if (have_product && (rc = one_iteration(1))) return rc;
      if ((rc = wait_dots(c, seq))) return rc;
      std::atomic_thread_fence(std::memory_order_acquire);
      continue;
    }
    if (pipelined) {
      if ((rc = one_iteration())) return rc;
      ++enqueued;
      in_flight.push_back(seq);
      if (static_cast<int>(in_flight.size()) > depth) {
        if ((rc = wait_dots(c, in_flight.front()))) return rc;
        in_flight.pop_front();
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      continue;
    }
    for (int b = 0; b < batch && enqueued < max_iters; ++b, ++enqueued) {
      if ((rc = one_iteration())) {
        (void)end_capture(false);
        return rc;
      }
    }
    if ((rc = end_capture(true))) return rc;
    if ((rc = wait_dots(c, seq))) return rc;
  }
  if (c->h_stpcg[0].status == 0 || c->prof_stpcg) {  // the iteration limit ended the loop: what is in flight decides
    while (!in_flight.empty()) {
      if ((rc = wait_dots(c, in_flight.front()))) return rc;
      in_flight.pop_front();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (c->h_stpcg[0].status != 0 && !c->prof_stpcg) {
    // the mirror was written by the iteration that set the status (and only rewritten with the same iteration count,
    // step norm and status by a neutral one after it): final, nothing to wait for.  What is still in flight is
    // neutral and ordered before anything enqueued after this call; the next solve makes sure of it before it resets the mirror.
    H = c->h_stpcg[0];
    c->stpcg_pending_seq = in_flight.empty() ? 0 : in_flight.back();
  } else {
    // an iteration that starts at the limit only records the status: flush it so that the mirror is final
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(&H, c->d_stpcg, sizeof(StpcgState), hipMemcpyDeviceToHost));
  }
  if (c->prof_stpcg) {  // iterations that really ran (enqueued-ahead ones after the stop are neutral but timed)
    const bool phases = c->prof_stpcg >= 2 && c->stpcg_path == 2 && !sharded;
    for (int k = 0; k < 7; ++k) c->prof_phase_us[k] = -1.0;
    const bool tail_marks = c->stpcg_path == 2 && !sharded;  // (marks 6 and 7 exist on the sweep-fused form)
    for (int k = 0; k < 7; ++k) {
      if (!(k == 0 || phases || (k == 6 && tail_marks))) continue;
      double tot = 0.0;
      int cnt = 0;
      for (int i = 0; i < H.iters && kProfMarks * (static_cast<size_t>(i) + 1) <= c->prof_events.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->prof_events[kProfMarks * i + k], c->prof_events[kProfMarks * i + k + 1]) == hipSuccess) {
          tot += ms * 1e3;
          ++cnt;
        }
      }
      c->prof_phase_us[k] = cnt ? tot / cnt : -1.0;
      if (k == 0) {
        c->prof_hvp_us = cnt ? tot / cnt : 0.0;
        c->prof_hvp_count = cnt;
      }
    }
  }
  *iters = H.iters;
  *step_M_norm = H.step_M_norm;
  return CORA_OK;
}

int cora_stpcg_dev(cora_ctx *c, const double *dGrad, double Delta, double kappa_fgr, double theta, int max_iters,
                   double *dS, double *dR, double *dV, double *dP, double *dHp, int *iters, double *step_M_norm) {
  return stpcg_run(c, dGrad, nullptr, 0.0, 0.0, Delta, kappa_fgr, theta, max_iters, dS, dR, dV, dP, dHp, iters, step_M_norm);
}

int cora_stpcg_warm_dev(cora_ctx *c, const double *dGrad, const double *dPg, double g_g, double g_Pg, double Delta,
                        double kappa_fgr, double theta, int max_iters, double *dS, double *dR, double *dV, double *dP,
                        double *dHp, int *iters, double *step_M_norm) {
  if (!c || !dPg || dPg == dS || dPg == dR || dPg == dV || dPg == dP || dPg == dHp) return fail(c, CORA_ERR_ARG, "bad arguments");
  return stpcg_run(c, dGrad, dPg, g_g, g_Pg, Delta, kappa_fgr, theta, max_iters, dS, dR, dV, dP, dHp, iters, step_M_norm);
}

int cora_set_comm(cora_ctx *c, cora_exchange_fn exchange, cora_allreduce_fn allreduce, cora_allgather_fn allgather,
                  void *user) {
  if (!c) return CORA_ERR_ARG;
  c->comm_exchange = exchange;
  c->comm_allreduce = allreduce;
  c->comm_allgather = allgather;
  c->comm_user = user;
  return CORA_OK;
}

int cora_stpcg_device_ok(const cora_ctx *c) { return (c && c->has_device && c->p > 0 && stpcg_device_ok(c)) ? 1 : 0; }

int cora_require_comm(cora_ctx *c, int on) {
  if (!c) return CORA_ERR_ARG;
  c->comm_required = on != 0;
  return CORA_OK;
}

int cora_rank(const cora_ctx *c) { return c ? c->F.L.rank : 0; }
int cora_world(const cora_ctx *c) { return c ? c->F.L.world : 0; }

int cora_pack_rows_dev(cora_ctx *c, const double *dX, int ld, const int32_t *d_rows, int64_t n, double *dPacked) {
  NEED_DEVICE(c);
  wrote(c, dPacked);
  if (!dX || !d_rows || !dPacked || ld <= 0 || n < 0) return fail(c, CORA_ERR_ARG, "bad arguments");
  HIP_TRY(c, launch_move_rows(0, n, ld, d_rows, dX, dPacked, c->stream));
  return CORA_OK;
}

int cora_scatter_rows_dev(cora_ctx *c, const double *dPacked, int ld, const int32_t *d_rows, int64_t n, double *dX) {
  NEED_DEVICE(c);
  wrote(c, dX);
  if (!dX || !d_rows || !dPacked || ld <= 0 || n < 0) return fail(c, CORA_ERR_ARG, "bad arguments");
  HIP_TRY(c, launch_move_rows(1, n, ld, d_rows, dPacked, dX, c->stream));
  return CORA_OK;
}

int cora_copy_rows_dev(cora_ctx *c, const double *dSrc, int ld, const int32_t *d_rows, int64_t n, double *dDst) {
  NEED_DEVICE(c);
  wrote(c, dDst);
  if (!dSrc || !d_rows || !dDst || ld <= 0 || n < 0) return fail(c, CORA_ERR_ARG, "bad arguments");
  HIP_TRY(c, launch_move_rows(2, n, ld, d_rows, dSrc, dDst, c->stream));
  return CORA_OK;
}

int cora_copy_shard_dev(cora_ctx *c, const double *dSrc, int ld, int shard, double *dDst) {
  NEED_DEVICE(c);
  wrote(c, dDst);
  if (!dSrc || !dDst || ld <= 0 || shard < 0 || shard >= c->F.L.world) return fail(c, CORA_ERR_ARG, "bad arguments");
  const size_t off = static_cast<size_t>(shard) * c->F.L.shard_rows * ld;
  HIP_TRY(c, hipMemcpyAsync(dDst + off, dSrc + off, static_cast<size_t>(c->F.L.shard_rows) * ld * sizeof(double),
                            hipMemcpyDeviceToDevice, c->stream));
  return CORA_OK;
}

int cora_debug_profile_stpcg(cora_ctx *c, int on) {
  NEED_DEVICE(c);
  c->prof_stpcg = on < 0 ? 0 : (on > 2 ? 2 : on);
  if (on && c->prof_events.empty()) {
    c->prof_events.resize(kProfMarks * 256, nullptr);
    for (hipEvent_t &e : c->prof_events) HIP_TRY(c, hipEventCreate(&e));
  }
  return CORA_OK;
}

int cora_debug_stpcg_phase_us(cora_ctx *c, double us[8]) {
  if (!c || !us) return CORA_ERR_ARG;
  for (int k = 0; k < 7; ++k) us[k] = c->prof_phase_us[k];
  us[7] = c->prof_kappa_folded ? 1.0 : 0.0;
  return CORA_OK;
}

int cora_debug_stpcg_path(const cora_ctx *c) { return c ? c->stpcg_path : -1; }

int cora_debug_stpcg_graph(const cora_ctx *c, long out[2]) {
  if (!c || !out) return CORA_ERR_ARG;
  out[0] = c->stpcg_graph_captures;
  out[1] = c->stpcg_graph_replays;
  return CORA_OK;
}

int cora_debug_stpcg_hvp_us(cora_ctx *c, double *mean_us, int *count) {
  if (!c || !mean_us || !count) return CORA_ERR_ARG;
  *mean_us = c->prof_hvp_us;
  *count = c->prof_hvp_count;
  return CORA_OK;
}

int cora_dot_dev(cora_ctx *c, const double *dA, const double *dB, int k, double *out) {
  NEED_DEVICE(c);
  if (k <= 0 || k > kMaxLD || !dA || !dB || !out) return fail(c, CORA_ERR_ARG, "bad arguments");
  const int ld = ld_for(k);
  DotArgs D;
  const size_t off = static_cast<size_t>(c->F.L.base) * ld;
  for (int j = 0; j < 4; ++j) { D.a[j] = nullptr; D.b[j] = nullptr; }
  D.a[0] = dA + off;
  D.b[0] = dB + off;
  D.count = 1;
  D.n2 = c->F.L.local_rows * ld;
  int rc = ensure_red(c, 4 * 512);
  if (rc) return rc;
  D.partial = c->d_red;
  D.ticket = c->d_ticket;
  D.out = c->h_scalars;
  D.seq_out = reinterpret_cast<unsigned long long *>(c->h_scalars + 7);
  D.seq = ++c->dot_seq;
  D.mode = DOTS_PLAIN;
  D.st = D.st_host = nullptr;
  int nblocks = 0;
  HIP_TRY(c, launch_dots(D, &nblocks, c->stream));
  if ((rc = wait_dots(c, D.seq))) return rc;
  *out = c->h_scalars[0];
  return comm_allreduce(c, out, 1);
}

int cora_gram_dev(cora_ctx *c, const double *dA, int ka, const double *dB, int kb, double *G) {
  NEED_DEVICE(c);
  if (!dA || !dB || !G || ka <= 0 || kb <= 0 || ka > kMaxLD || kb > kMaxLD) return fail(c, CORA_ERR_ARG, "bad arguments");
  const int nblocks = 256, nel = ka * kb;
  int rc = ensure_red(c, static_cast<size_t>(nel) * nblocks + nel);
  if (rc) return rc;
  if (!c->h_gram) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_gram), 16 * kMaxLD * kMaxLD * sizeof(double)));
  // (the reduction writes the results to pinned host memory itself: no copy, one wait)
  HIP_TRY(c, launch_gram(c->F.L.base, c->F.L.local_rows, dA, ka, dB, kb, c->d_red, nblocks, c->h_gram, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const double *tmp = c->h_gram;
  for (int a = 0; a < ka; ++a)  // device result is row-major ka x kb
    for (int b = 0; b < kb; ++b) G[static_cast<size_t>(b) * ka + a] = tmp[static_cast<size_t>(a) * kb + b];
  return comm_allreduce(c, G, nel);
}

int cora_gram_batch_dev(cora_ctx *c, int n, const double *const *dA, const int *ka, const double *const *dB,
                        const int *kb, double *const *G) {
  NEED_DEVICE(c);
  if (n < 1 || n > 16 || !dA || !ka || !dB || !kb || !G) return fail(c, CORA_ERR_ARG, "bad arguments");
  const int nblocks = 256;
  size_t need = 0, nel_all = 0;
  for (int e = 0; e < n; ++e) {
    if (!dA[e] || !dB[e] || !G[e] || ka[e] <= 0 || kb[e] <= 0 || ka[e] > kMaxLD || kb[e] > kMaxLD)
      return fail(c, CORA_ERR_ARG, "bad block");
    const size_t nel = static_cast<size_t>(ka[e]) * kb[e];
    need += nel * nblocks;
    nel_all += nel;
  }
  // every product is the block of cora_gram_dev's kernel on its own piece of the reduction buffer: the numbers are those
  // of n separate calls -- in TWO launches (all products | all reductions, which write the results to pinned host memory)
  // and one wait, where a Rayleigh-Ritz step's twelve products were 24 launches, a copy and a wait
  int rc = ensure_red(c, need);
  if (rc) return rc;
  if (!c->h_gram) HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_gram), 16 * kMaxLD * kMaxLD * sizeof(double)));
  HIP_TRY(c, launch_gram_batch(c->F.L.base, c->F.L.local_rows, n, dA, ka, dB, kb, c->d_red, nblocks, c->h_gram, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const double *tmp = c->h_gram;
  size_t off = 0;
  for (int e = 0; e < n; ++e) {
    for (int a = 0; a < ka[e]; ++a)  // device result is row-major ka x kb
      for (int b = 0; b < kb[e]; ++b) G[e][static_cast<size_t>(b) * ka[e] + a] = tmp[off + static_cast<size_t>(a) * kb[e] + b];
    rc = comm_allreduce(c, G[e], ka[e] * kb[e]);
    if (rc) return rc;
    off += static_cast<size_t>(ka[e]) * kb[e];
  }
  return CORA_OK;
}

int cora_combine_dev(cora_ctx *c, int n, const double *const *dX, const int *k, const double *const *C, int kout,
                     double *dOut) {
  NEED_DEVICE(c);
  wrote(c, dOut);
  if (n < 1 || n > 4 || !dX || !k || !C || !dOut || kout <= 0 || kout > kMaxLD) return fail(c, CORA_ERR_ARG, "bad arguments");
  std::vector<double> coef;
  int coff[4] = {0, 0, 0, 0};
  for (int b = 0; b < n; ++b) {
    if (k[b] <= 0 || k[b] > kMaxLD || !dX[b] || !C[b]) return fail(c, CORA_ERR_ARG, "bad block");
    if (dX[b] == dOut) return fail(c, CORA_ERR_ARG, "output aliases an input block");
    coff[b] = static_cast<int>(coef.size());
    for (int i = 0; i < k[b]; ++i)
      for (int j = 0; j < kout; ++j) coef.push_back(C[b][static_cast<size_t>(j) * k[b] + i]);  // row-major on device
  }
  if (coef.size() <= static_cast<size_t>(kCombineKargMax)) {  // the coefficients ride in the kernel's arguments: nothing to wait for
    HIP_TRY(c, launch_combine(c->F.L.base, c->F.L.local_rows, n, dX, k, coff, nullptr, static_cast<int>(coef.size()), kout, dOut,
                              c->stream, coef.data()));
    return CORA_OK;
  }
  int rc = ensure_red(c, coef.size() + 8);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_red, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // coef is a stack-lifetime host buffer
  HIP_TRY(c, launch_combine(c->F.L.base, c->F.L.local_rows, n, dX, k, coff, c->d_red, static_cast<int>(coef.size()),
                            kout, dOut, c->stream));
  return CORA_OK;
}

int cora_timer_start(cora_ctx *c) {
  NEED_DEVICE(c);
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  return CORA_OK;
}

int cora_timer_stop_ms(cora_ctx *c, float *ms) {
  NEED_DEVICE(c);
  if (!ms) return fail(c, CORA_ERR_ARG, "null pointer");
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  HIP_TRY(c, hipEventSynchronize(c->ev1));
  HIP_TRY(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return CORA_OK;
}

int cora_sync(cora_ctx *c) {
  NEED_DEVICE(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CORA_OK;
}

// ------------------------------------------------- host-pointer operator API

#define CHECK_LD(c, ld)                                                                  \
  do {                                                                                   \
    if ((ld) < (c)->F.L.N) return fail((c), CORA_ERR_SHAPE, "leading dimension smaller than N"); \
  } while (0)

int cora_data_matrix_product(cora_ctx *c, const double *X, int ldx, int k, double *out, int ldo) {
  NEED_DEVICE(c);
  if (k <= 0 || k > kMaxLD) return fail(c, CORA_ERR_SHAPE, "column count must be in [1, 24]");
  double *dX, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, ld_for(k), &dX))) return rc;
  if ((rc = get_scratch(c, 1, ld_for(k), &dO))) return rc;
  if ((rc = upload_impl(c, X, ldx, k, dX))) return rc;
  HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, ld_for(k)), c->stream));
  if ((rc = cora_spmm_dev(c, dX, k, dO))) return rc;
  return download_impl(c, dO, k, out, ldo);
}

int cora_evaluate_objective(cora_ctx *c, const double *Y, int ldy, double *f) {
  if (!f) return fail(c, CORA_ERR_ARG, "null pointer");
  int rc = cora_set_point(c, Y, ldy);
  if (rc) return rc;
  *f = c->f;
  return CORA_OK;
}

int cora_euclidean_gradient(cora_ctx *c, const double *Y, int ldy, double *out, int ldo) {
  int rc = cora_set_point(c, Y, ldy);
  if (rc) return rc;
  return download_impl(c, c->d_G, c->p, out, ldo);
}

int cora_riemannian_gradient(cora_ctx *c, const double *Y, int ldy, double *out, int ldo) {
  int rc = cora_set_point(c, Y, ldy);
  if (rc) return rc;
  return download_impl(c, c->d_rgrad, c->p, out, ldo);
}

int cora_tangent_space_projection(cora_ctx *c, const double *Y, int ldy, const double *V, int ldv,
                                  double *out, int ldo) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  double *dV, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, c->ld, &dV))) return rc;
  if ((rc = get_scratch(c, 1, c->ld, &dO))) return rc;
  if ((rc = upload_impl(c, Y, ldy, c->p, c->d_Y))) return rc;
  c->have_point = false;  // Y replaced without refreshing the cached gradient
  c->trial_x = nullptr;
  if ((rc = upload_impl(c, V, ldv, c->p, dV))) return rc;
  HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, c->ld), c->stream));
  HIP_TRY(c, launch_tangent_project(row_args(c), c->ld, c->d_Y, dV, nullptr, dO, c->stream));
  return download_impl(c, dO, c->p, out, ldo);
}

int cora_riemannian_hessian_vector_product(cora_ctx *c, const double *Y, int ldy, const double *G, int ldg,
                                           const double *dotY, int ldd, double *out, int ldo) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  double *dX, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, c->ld, &dX))) return rc;
  if ((rc = get_scratch(c, 1, c->ld, &dO))) return rc;
  // honour the reference signature: Lambda is built from the nablaF_Y passed in
  if ((rc = upload_impl(c, Y, ldy, c->p, c->d_Y))) return rc;
  if ((rc = upload_impl(c, G, ldg, c->p, c->d_G))) return rc;
  if ((rc = point_finish(c))) return rc;
  if ((rc = upload_impl(c, dotY, ldd, c->p, dX))) return rc;
  HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, c->ld), c->stream));
  if ((rc = cora_hvp_dev(c, dX, dO))) return rc;
  return download_impl(c, dO, c->p, out, ldo);
}

int cora_project_to_manifold(cora_ctx *c, const double *A, int lda, double *out, int ldo) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  double *dA, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, c->ld, &dA))) return rc;
  if ((rc = get_scratch(c, 1, c->ld, &dO))) return rc;
  if ((rc = upload_impl(c, A, lda, c->p, dA))) return rc;
  HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, c->ld), c->stream));
  HIP_TRY(c, launch_project_manifold(row_args(c), c->ld, dA, nullptr, 0.0, dO, c->stream));
  return download_impl(c, dO, c->p, out, ldo);
}

int cora_retract(cora_ctx *c, const double *Y, int ldy, const double *V, int ldv, double *out, int ldo) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  double *dY, *dV, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, c->ld, &dY))) return rc;
  if ((rc = get_scratch(c, 1, c->ld, &dV))) return rc;
  if ((rc = get_scratch(c, 2, c->ld, &dO))) return rc;
  if ((rc = upload_impl(c, Y, ldy, c->p, dY))) return rc;
  if ((rc = upload_impl(c, V, ldv, c->p, dV))) return rc;
  HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, c->ld), c->stream));
  HIP_TRY(c, launch_project_manifold(row_args(c), c->ld, dY, dV, 1.0, dO, c->stream));
  return download_impl(c, dO, c->p, out, ldo);
}

int cora_precondition(cora_ctx *c, const double *V, int ldv, double *out, int ldo) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (c->precond == CORA_PRECOND_NONE)
    return fail(c, CORA_ERR_NOT_READY, "preconditioner not set up (cora_precond_setup)");
  double *dV, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, c->ld, &dV))) return rc;
  if ((rc = get_scratch(c, 1, c->ld, &dO))) return rc;
  const size_t off = static_cast<size_t>(c->F.L.base) * c->ld;
  if (c->precond == CORA_PRECOND_JACOBI) {
    if ((rc = upload_impl(c, V, ldv, c->p, dV))) return rc;
    HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, c->ld), c->stream));
    HIP_TRY(c, launch_scale_rows(c->F.L.local_rows, c->ld, c->d_diag_inv, dV + off, dO + off, c->stream));
  } else {
    if ((rc = upload_impl(c, V, ldv, c->p, dV))) return rc;
    if ((rc = chol_solve(c, c->ld, dV, dO))) return rc;
  }
  // NaN guard, src/CORA_problem.cpp:898-901
  HIP_TRY(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  HIP_TRY(c, launch_has_nan(c->F.L.local_rows * c->ld, dO + off, c->d_flag, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (*c->h_flag) return fail(c, CORA_ERR_NAN, "NaNs in preconditioned vector");
  return download_impl(c, dO, c->p, out, ldo);
}

int cora_compute_lambda_blocks(cora_ctx *c, const double *Y, int ldy, double *stiefel, double *oblique) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  int rc = cora_set_point(c, Y, ldy);
  if (rc) return rc;
  const Layout &L = c->F.L;
  if (L.world != 1) {
    // Partitioned handle: every rank holds the blocks of its own poses and range rows.  They travel as a resident
    // vector with d columns -- row (pose, a) carries row a of the pose's block, a range row its multiplier in column 0
    // -- through the collective download (one all-gather of the shards); every rank gets all of them.
    const int k = L.d, ld = ld_for(k);
    if (ld != L.d) return fail(c, CORA_ERR_ARG, "unexpected row stride");
    double *vec;
    if ((rc = get_scratch(c, 2, ld, &vec))) return rc;
    HIP_TRY(c, hipMemsetAsync(vec, 0, vec_bytes(c, ld), c->stream));
    if (L.nl_poses > 0)  // [pose][d * d] IS rows (pose, a) x d columns at row stride d
      HIP_TRY(c, hipMemcpyAsync(vec + static_cast<size_t>(L.rot_base) * ld, c->d_lam_st,
                                static_cast<size_t>(L.nl_poses) * L.d * L.d * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (L.nl_ranges > 0)
      HIP_TRY(c, hipMemcpy2DAsync(vec + static_cast<size_t>(L.rng_base) * ld, ld * sizeof(double), c->d_lam_ob, sizeof(double),
                                  sizeof(double), static_cast<size_t>(L.nl_ranges), hipMemcpyDeviceToDevice, c->stream));
    std::vector<double> M(static_cast<size_t>(L.N) * k);
    if ((rc = download_impl(c, vec, k, M.data(), static_cast<int>(L.N)))) return rc;
    if (stiefel)
      for (int64_t i = 0; i < L.n; ++i)
        for (int a = 0; a < L.d; ++a)
          for (int b = 0; b < L.d; ++b)
            stiefel[(i * L.d + b) * L.d + a] = M[static_cast<size_t>(i * L.d + a) + static_cast<size_t>(L.N) * b];
    if (oblique)
      for (int64_t j = 0; j < L.r; ++j) oblique[j] = M[static_cast<size_t>(L.d) * L.n + j];
    return CORA_OK;
  }
  // a symmetric d x d block is the same row- or column-major, so the device
  // array [pose][d*d] already is the d x (d n) column-major matrix
  if (L.n > 0 && stiefel)
    HIP_TRY(c, hipMemcpyAsync(stiefel, c->d_lam_st, static_cast<size_t>(L.n) * L.d * L.d * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
  std::vector<double> ob(static_cast<size_t>(std::max(L.r, 1)));
  if (L.r > 0 && oblique)
    HIP_TRY(c, hipMemcpyAsync(ob.data(), c->d_lam_ob, static_cast<size_t>(L.r) * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // the device keeps range rows in its internal (pose-sorted) order
  if (L.r > 0 && oblique)
    for (int k = 0; k < L.r; ++k)
      oblique[k] = ob[static_cast<size_t>(c->F.api2int[static_cast<size_t>(L.d) * L.n + k] - L.rng_base)];
  return CORA_OK;
}

int cora_certificate_product(cora_ctx *c, const double *X, int ldx, int k, double *out, int ldo) {
  NEED_DEVICE(c);
  if (!c->have_point) return fail(c, CORA_ERR_NOT_READY, "no current point (cora_set_point)");
  double *dX, *dO;
  int rc;
  if ((rc = get_scratch(c, 0, ld_for(k), &dX))) return rc;
  if ((rc = get_scratch(c, 1, ld_for(k), &dO))) return rc;
  if ((rc = upload_impl(c, X, ldx, k, dX))) return rc;
  HIP_TRY(c, hipMemsetAsync(dO, 0, vec_bytes(c, ld_for(k)), c->stream));
  if ((rc = cora_certificate_product_dev(c, dX, k, dO))) return rc;
  return download_impl(c, dO, k, out, ldo);
}

int cora_inner_product(cora_ctx *c, const double *A, int lda, const double *B, int ldb, int k, double *out) {
  NEED_DEVICE(c);
  double *dA, *dB;
  int rc;
  if ((rc = get_scratch(c, 0, ld_for(k), &dA))) return rc;
  if ((rc = get_scratch(c, 1, ld_for(k), &dB))) return rc;
  if ((rc = upload_impl(c, A, lda, k, dA))) return rc;
  if ((rc = upload_impl(c, B, ldb, k, dB))) return rc;
  return cora_dot_dev(c, dA, dB, k, out);
}

int cora_debug_format_spmm_host(const cora_ctx *c, const double *X, int ldx, int k, double *out, int ldo) {
  if (!c || !X || !out || k <= 0 || k > kMaxLD) return CORA_ERR_ARG;
  const HostFormat &F = c->F;
  const int ld = ld_for(k);
  const int64_t N = F.L.N;
  std::vector<double> xi(static_cast<size_t>(F.L.rows) * ld, 0.0), oi(static_cast<size_t>(F.L.rows) * ld, 0.0);
  for (int cc = 0; cc < k; ++cc)
    for (int64_t i = 0; i < N; ++i)
      xi[static_cast<size_t>(F.api2int[i]) * ld + cc] = X[static_cast<size_t>(cc) * ldx + i];
  format_spmm_host(F, xi.data(), ld, oi.data());
  for (int cc = 0; cc < k; ++cc)
    for (int64_t i = 0; i < N; ++i)
      out[static_cast<size_t>(cc) * ldo + i] = oi[static_cast<size_t>(F.api2int[i]) * ld + cc];
  return CORA_OK;
}

int cora_debug_factor_solve_host(int m, const int32_t *Lp, const int32_t *Li, const double *Lx, int k,
                                 const double *B, double *X, int64_t stats[4]) {
  if (m <= 0 || !Lp || !Li || !Lx || !B || !X || k <= 0) return CORA_ERR_ARG;
  try {
    std::vector<int32_t> row_of(static_cast<size_t>(m));
    for (int i = 0; i < m; ++i) row_of[i] = i;
    TriPlan P;
    build_tri_plan(m, Lp, Li, Lx, row_of, m, P, nullptr, m + 1);  // row m plays the pinned variable
    std::vector<double> rhs(static_cast<size_t>(m) + 1), out(static_cast<size_t>(m) + 1);
    for (int cc = 0; cc < k; ++cc) {
      std::copy(B + static_cast<size_t>(cc) * m, B + static_cast<size_t>(cc + 1) * m, rhs.begin());
      rhs[m] = 1.0;
      std::fill(out.begin(), out.end(), 7.0);
      tri_plan_solve_host(P, m + 1, rhs.data(), out.data());
      if (out[m] != 0.0) return fail(nullptr, CORA_ERR_ARG, "the pinned row was not zeroed");
      std::copy(out.begin(), out.begin() + m, X + static_cast<size_t>(cc) * m);
    }
    if (stats) {
      stats[0] = static_cast<int64_t>(P.stages.size());
      stats[1] = P.nnzW;
      stats[2] = P.nnzL;
      stats[3] = P.stages.empty() ? 0 : (P.stages[0].dense ? static_cast<int64_t>(P.stages[0].blocks_op.nrows.size())
                                                        : (P.stages[0].sub ? static_cast<int64_t>(P.stages[0].sub_op.nrows.size()) : 0));
    }
  } catch (const std::exception &e) {
    return fail(nullptr, CORA_ERR_ARG, e.what());
  }
  return CORA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Native communication of a partitioned handle (the collective steps of include/cora_hip.h,
// cora_set_comm, provided by the library itself instead of injected callbacks).  Two transports behind one plan:
//   RCCL  -- one process per GPU: ncclAllGather / ncclAllReduce on the handle's stream (librccl.so is opened at run
//            time, so single-GPU users carry no dependency); the id is created on rank 0 (cora_rccl_unique_id) and
//            handed to the other ranks by whatever launched them (torch.distributed in bench.py, MPI, a file);
//   local -- every rank a thread of ONE process with its own handle (and stream) on one or several visible devices:
//            device-to-device copies between the ranks' buffers behind a host barrier.  This is how the sharded
//            solver is tested on a one-GPU box, and it runs the same planning, pack and scatter code as RCCL.
// The exchange moves only the rows somebody reads: pack (k_move_rows) -> ONE all-gather of the packed rows ->
// scatter.  Every rank pads its export list to the longest one with its own first row.
// ---------------------------------------------------------------------------------------------------------

struct cora_local_group {
  int world = 0;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  uint64_t generation = 0;
  bool broken = false;
  std::vector<const void *> ptrs;          // what every rank published for the current step
  std::vector<std::vector<double>> vals;   // host all-reduce operands
  // returns false if the group was broken (a rank failed): nobody waits for ever
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t gen = generation;
    if (++waiting == world) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(600), [&] { return generation != gen || broken; })) broken = true;
    if (broken) cv.notify_all();
    return !broken;
  }
};

namespace {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // (optional: what the communicator itself reports)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};

const RcclApi *rccl_api(std::string *err) {
  static RcclApi api;
  static std::once_flag once;
  static std::string load_error;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      load_error = std::string("cannot open librccl.so: ") + dlerror();
      return;
    }
    auto sym = [&](const char *n) {
      void *q = dlsym(api.lib, n);
      if (!q && load_error.empty()) load_error = std::string("librccl.so lacks ") + n;
      return q;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.lib, "ncclCommUserRank"));
  });
  if (!load_error.empty()) {
    if (err) *err = load_error;
    return nullptr;
  }
  return &api;
}

}  // namespace

struct cora_native_comm {
  cora_ctx *c = nullptr;
  int rank = 0, world = 1;
  cora_local_group *g = nullptr;  // local transport
  const RcclApi *api = nullptr;   // RCCL transport
  ncclComm_t nccl = nullptr;
  cora::P2PState *p2p = nullptr;  // device-side transport over peer-mapped mailboxes (p2p.h): no RCCL, no host on the data path
  long n_p2p_gather = 0, n_p2p_reduce = 0;
  // exchange plan
  int e_max = 0;
  int64_t exchanged_rows = 0;       // rows received per exchange (world * e_max)
  int32_t *d_export = nullptr;      // [e_max] rows of this rank's shard that some other rank reads (padded)
  int32_t *d_recv_idx = nullptr;    // [world * e_max] where the gathered rows go, rank by rank
  struct Buf { double *send = nullptr, *recv = nullptr; };
  std::map<int, Buf> buf;           // per row stride
  double *d_scal = nullptr, *h_scal = nullptr;  // 1024 doubles each (device / pinned): all-reduce staging
  std::string err;
  long n_allgather = 0, n_allreduce = 0;  // collectives issued on the data path (cora_comm_counters)
  int n_long() const { return static_cast<int>(c->F.long_rows.size()); }

  int fail_(const std::string &m) {
    err = m;
    if (c) c->err = m;
    if (g) {  // release the other ranks
      std::lock_guard<std::mutex> lk(g->m);
      g->broken = true;
      g->cv.notify_all();
    }
    return 1;
  }
  int hip(hipError_t e, const char *what) { return e == hipSuccess ? 0 : fail_(std::string(what) + ": " + hipGetErrorString(e)); }
  int nc(ncclResult_t r, const char *what) {
    return r == ncclSuccess ? 0 : fail_(std::string(what) + ": " + (api && api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
  }

  // all-gather of `bytes` bytes per rank between DEVICE buffers, ordered on the handle's stream
  int allgather_dev(const void *send, void *recv, size_t bytes, hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    if (p2p) {  // one kernel: push into the peers' mailboxes, wait for theirs, copy out (counted apart: not a library collective)
      ++n_p2p_gather;
      return cora::p2p_allgather(p2p, send, recv, bytes, st, &err) ? fail_(err) : 0;
    }
    ++n_allgather;
    if (nccl) return nc(api->AllGather(send, recv, bytes, ncclChar, nccl, st), "ncclAllGather");
    if (hip(hipStreamSynchronize(st), "hipStreamSynchronize")) return 1;
    g->ptrs[rank] = send;
    if (!g->barrier()) return fail_("local group broken");
    for (int r = 0; r < world; ++r)
      if (hip(hipMemcpyAsync(static_cast<char *>(recv) + static_cast<size_t>(r) * bytes, g->ptrs[r], bytes,
                             hipMemcpyDeviceToDevice, st), "hipMemcpyAsync")) return 1;
    if (hip(hipStreamSynchronize(st), "hipStreamSynchronize")) return 1;
    if (!g->barrier()) return fail_("local group broken");  // nobody reuses its send buffer before everyone has copied
    return 0;
  }
  // sum of n device doubles over the ranks, in place, ordered on the handle's stream (the same bits on every rank)
  int allreduce_dev(double *d, int n) {
    if (p2p) {
      ++n_p2p_reduce;
      return cora::p2p_allreduce(p2p, d, n, c->stream, &err) ? fail_(err) : 0;
    }
    if (nccl) {
      ++n_allreduce;
      return nc(api->AllReduce(d, d, static_cast<size_t>(n), ncclDouble, ncclSum, nccl, c->stream), "ncclAllReduce");
    }
    std::vector<double> h(static_cast<size_t>(n));
    if (hip(hipMemcpyAsync(h.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync")) return 1;
    if (hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return 1;
    if (allreduce_host(h.data(), n)) return 1;
    if (hip(hipMemcpyAsync(d, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync")) return 1;
    return hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  }
  int allreduce_host(double *vals, int n) {
    if (!p2p) ++n_allreduce;
    if (nccl || p2p) {
      if (n > 1016) return fail_("all-reduce of more than 1016 doubles");
      std::memcpy(h_scal, vals, sizeof(double) * n);
      if (hip(hipMemcpyAsync(d_scal, h_scal, sizeof(double) * n, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync")) return 1;
      if (p2p) {
        if (allreduce_dev(d_scal, n)) return 1;
      } else if (nc(api->AllReduce(d_scal, d_scal, static_cast<size_t>(n), ncclDouble, ncclSum, nccl, c->stream), "ncclAllReduce")) return 1;
      if (hip(hipMemcpyAsync(h_scal, d_scal, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync")) return 1;
      if (hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return 1;
      std::memcpy(vals, h_scal, sizeof(double) * n);
      return 0;
    }
    g->vals[rank].assign(vals, vals + n);
    if (!g->barrier()) return fail_("local group broken");
    std::vector<double> tot(static_cast<size_t>(n), 0.0);
    for (int r = 0; r < world; ++r)  // rank order: the same bits on every rank
      for (int i = 0; i < n; ++i) tot[i] += g->vals[r][i];
    if (!g->barrier()) return fail_("local group broken");
    std::copy(tot.begin(), tot.end(), vals);
    return 0;
  }
  // all-gather of host data (planning): `bytes` per rank
  int allgather_host(const void *send, void *recv, size_t bytes) {
    void *ds = nullptr, *dr = nullptr;
    if (hip(hipMalloc(&ds, std::max<size_t>(bytes, 8)), "hipMalloc") || hip(hipMalloc(&dr, std::max<size_t>(bytes, 8) * world), "hipMalloc")) return 1;
    int rc = hip(hipMemcpy(ds, send, bytes, hipMemcpyHostToDevice), "hipMemcpy");
    if (!rc) rc = allgather_dev(ds, dr, bytes);
    if (!rc) rc = hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (!rc) rc = hip(hipMemcpy(recv, dr, bytes * world, hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(ds);
    (void)hipFree(dr);
    return rc;
  }

  int plan() {
    const Layout &L = c->F.L;
    int64_t n_need = 0;
    cora_remote_rows(c, nullptr, &n_need);
    std::vector<int32_t> need(static_cast<size_t>(std::max<int64_t>(n_need, 1)));
    cora_remote_rows(c, need.data(), &n_need);
    // every rank learns every need-list (padded to the longest with -1)
    std::vector<int64_t> counts(static_cast<size_t>(world));
    if (allgather_host(&n_need, counts.data(), sizeof(int64_t))) return 1;
    const int64_t n_max = std::max<int64_t>(1, *std::max_element(counts.begin(), counts.end()));
    std::vector<int32_t> padded(static_cast<size_t>(n_max), -1), lists(static_cast<size_t>(n_max) * world);
    std::copy(need.begin(), need.begin() + n_need, padded.begin());
    if (allgather_host(padded.data(), lists.data(), sizeof(int32_t) * n_max)) return 1;
    // rows of rank r's shard that any OTHER rank reads, ascending
    std::vector<std::vector<int32_t>> exports(static_cast<size_t>(world));
    {
      std::vector<char> wanted(static_cast<size_t>(L.rows), 0);
      for (int r = 0; r < world; ++r)
        for (int64_t k = 0; k < counts[r]; ++k) wanted[lists[static_cast<size_t>(r) * n_max + k]] = 1;  // a rank never lists its own rows
      for (int r = 0; r < world; ++r)
        for (int64_t row = L.shard_rows * r; row < L.shard_rows * (r + 1); ++row)
          if (wanted[row]) exports[r].push_back(static_cast<int32_t>(row));
    }
    size_t em = 1;
    for (const auto &e : exports) em = std::max(em, e.size());
    e_max = static_cast<int>(em);
    exchanged_rows = static_cast<int64_t>(world) * e_max;
    std::vector<int32_t> recv_idx;
    for (int r = 0; r < world; ++r) {
      std::vector<int32_t> e = exports[r];
      e.resize(em, static_cast<int32_t>(L.shard_rows * r));  // padding: the shard's first row, sent with its own value
      recv_idx.insert(recv_idx.end(), e.begin(), e.end());
      if (r == rank) {
        if (hip(hipMalloc(reinterpret_cast<void **>(&d_export), sizeof(int32_t) * em), "hipMalloc")) return 1;
        if (hip(hipMemcpy(d_export, e.data(), sizeof(int32_t) * em, hipMemcpyHostToDevice), "hipMemcpy")) return 1;
      }
    }
    if (hip(hipMalloc(reinterpret_cast<void **>(&d_recv_idx), sizeof(int32_t) * recv_idx.size()), "hipMalloc")) return 1;
    return hip(hipMemcpy(d_recv_idx, recv_idx.data(), sizeof(int32_t) * recv_idx.size(), hipMemcpyHostToDevice), "hipMemcpy");
  }

  int buffers(int ld, Buf **out) {
    Buf &b = buf[ld];
    if (!b.send) {  // per rank: e_max rows + one slot per distributed long row
      const size_t per = sizeof(double) * (static_cast<size_t>(e_max) + n_long()) * ld;
      if (hip(hipMalloc(reinterpret_cast<void **>(&b.send), per), "hipMalloc")) return 1;
      if (hip(hipMalloc(reinterpret_cast<void **>(&b.recv), per * world), "hipMalloc")) return 1;
    }
    *out = &b;
    return 0;
  }

  // pack -> all-gather -> scatter, ordered on `st` (default: the handle's stream)
  int exchange(double *dX, int ld, hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    if (hip(hipSetDevice(c->device), "hipSetDevice")) return 1;
    Buf *b;
    if (buffers(ld, &b)) return 1;
    if (hip(launch_move_rows(0, e_max, ld, d_export, dX, b->send, st), "pack")) return 1;
    if (allgather_dev(b->send, b->recv, sizeof(double) * e_max * ld, st)) return 1;
    return hip(launch_move_rows(1, static_cast<int64_t>(world) * e_max, ld, d_recv_idx, b->recv, dX, st), "scatter");
  }
  int product_pack(const double *dX, int ld, hipStream_t st, double **slots) {
    if (hip(hipSetDevice(c->device), "hipSetDevice")) return 1;
    Buf *b;
    if (buffers(ld, &b)) return 1;
    *slots = b->send + static_cast<size_t>(e_max) * ld;
    return hip(launch_exchange_pack(e_max, ld, d_export, static_cast<int64_t>(n_long()) * ld, dX, b->send, st), "pack");
  }
  int product_gather(double *dX, int ld, hipStream_t st, double *out, double *kappa, hipEvent_t after_collective = nullptr) {
    Buf *b;
    if (buffers(ld, &b)) return 1;
    if (p2p && cora::p2p_exchange_unpack_fits(p2p, e_max, n_long(), ld)) {  // hand-over, wait and unpack in ONE kernel, from the mailbox
      ++n_p2p_gather;
      if (after_collective) (void)hipEventRecord(after_collective, st);
      return cora::p2p_exchange_unpack(p2p, b->send, e_max, n_long(), ld, d_recv_idx, dX, c->d_long_rows, c->d_long_owner, out, kappa, st, &err)
                 ? fail_(err) : 0;
    }
    if (allgather_dev(b->send, b->recv, sizeof(double) * (static_cast<size_t>(e_max) + n_long()) * ld, st)) return 1;
    if (after_collective) (void)hipEventRecord(after_collective, st);
    return hip(launch_exchange_unpack(world, e_max, n_long(), ld, d_recv_idx, b->recv, dX, rank, c->d_long_rows, c->d_long_owner,
                                      out, kappa, st), "unpack");
  }
  int allgather(double *dX, int ld) {  // whole shards, in place
    ++n_allgather;
    rows_gathered += static_cast<long long>(c->F.L.shard_rows) * world;
    if (hip(hipSetDevice(c->device), "hipSetDevice")) return 1;
    const Layout &L = c->F.L;
    const size_t n = static_cast<size_t>(L.shard_rows) * ld;
    if (p2p) {
      --n_allgather;
      return allgather_dev(dX + n * rank, dX, n * sizeof(double));  // in place: a rank's own piece is delivered onto itself
    }
    if (nccl) return nc(api->AllGather(dX + n * rank, dX, n, ncclDouble, nccl, c->stream), "ncclAllGather");
    if (hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return 1;
    g->ptrs[rank] = dX;
    if (!g->barrier()) return fail_("local group broken");
    for (int r = 0; r < world; ++r)
      if (r != rank && hip(hipMemcpyAsync(dX + n * r, static_cast<const double *>(g->ptrs[r]) + n * r, n * sizeof(double),
                                          hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync")) return 1;
    if (hip(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return 1;
    return g->barrier() ? 0 : fail_("local group broken");
  }
  // All-gather of ONE contiguous piece of every shard -- rows [row0, row0 + nrows) of the caller's own shard, every rank
  // its own piece (different offsets and lengths) -- into X in place: what the replicated translation solve of the
  // implicit formulation needs (the translation rows: 2 / 9 of a shard's rows at d = 3) instead of whole shards.
  // The pieces travel packed: own piece -> send buffer, one all-gather of the longest piece's size, one scatter kernel.
  struct PackedRows {
    int64_t row0 = -1, nrows = -1, maxn = 0;
    std::vector<int64_t> meta;   // {row0, nrows} of every rank
    int64_t *d_meta = nullptr;
    std::map<int, Buf> buf;      // per row stride
  } packed;
  long long rows_gathered = 0;   // rows received by all-gathers of resident vectors so far (whole shards or packed pieces)
  int allgather_rows(double *dX, int ld, int64_t row0, int64_t nrows) {
    if (hip(hipSetDevice(c->device), "hipSetDevice")) return 1;
    const Layout &L = c->F.L;
    if (packed.row0 != row0 || packed.nrows != nrows) {  // first call (the piece of a handle does not change): learn every rank's
      const int64_t mine[2] = {row0, nrows};
      packed.meta.assign(static_cast<size_t>(2 * world), 0);
      if (allgather_host(mine, packed.meta.data(), sizeof(mine))) return 1;
      if (!p2p) --n_allgather;  // (planning, not the data path)
      packed.maxn = 0;
      for (int r = 0; r < world; ++r) packed.maxn = std::max(packed.maxn, packed.meta[static_cast<size_t>(2 * r + 1)]);
      if (!packed.d_meta && hip(hipMalloc(&packed.d_meta, sizeof(int64_t) * 2 * world), "hipMalloc")) return 1;
      if (hip(hipMemcpy(packed.d_meta, packed.meta.data(), sizeof(int64_t) * 2 * world, hipMemcpyHostToDevice), "hipMemcpy")) return 1;
      for (auto &kv : packed.buf) {
        if (kv.second.send) (void)hipFree(kv.second.send);
        if (kv.second.recv) (void)hipFree(kv.second.recv);
      }
      packed.buf.clear();
      packed.row0 = row0;
      packed.nrows = nrows;
    }
    const size_t per = static_cast<size_t>(packed.maxn) * ld;
    rows_gathered += packed.maxn * world;
    if (per == 0) return 0;
    // (one path for both transports: the in-process one runs the same staging, collective and scatter kernel as RCCL)
    Buf &B = packed.buf[ld];
    if (!B.send) {
      if (hip(hipMalloc(&B.send, per * sizeof(double)), "hipMalloc") || hip(hipMalloc(&B.recv, per * world * sizeof(double)), "hipMalloc")) return 1;
      if (hip(hipMemsetAsync(B.send, 0, per * sizeof(double), c->stream), "hipMemsetAsync")) return 1;
    }
    if (nrows > 0 && hip(hipMemcpyAsync(B.send, dX + static_cast<size_t>(rank * L.shard_rows + row0) * ld, static_cast<size_t>(nrows) * ld * sizeof(double),
                                        hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync")) return 1;
    if (allgather_dev(B.send, B.recv, per * sizeof(double))) return 1;
    return hip(launch_scatter_shard_rows(world, rank, packed.maxn, ld, L.shard_rows, packed.d_meta, B.recv, dX, c->stream), "scatter");
  }
  ~cora_native_comm() {
    if (c && c->has_device) (void)hipSetDevice(c->device);
    for (auto &kv : packed.buf) {
      if (kv.second.send) (void)hipFree(kv.second.send);
      if (kv.second.recv) (void)hipFree(kv.second.recv);
    }
    if (packed.d_meta) (void)hipFree(packed.d_meta);
    for (auto &kv : buf) {
      if (kv.second.send) (void)hipFree(kv.second.send);
      if (kv.second.recv) (void)hipFree(kv.second.recv);
    }
    if (d_export) (void)hipFree(d_export);
    if (d_recv_idx) (void)hipFree(d_recv_idx);
    if (d_scal) (void)hipFree(d_scal);
    if (h_scal) (void)hipHostFree(h_scal);
    if (nccl && api) (void)api->CommDestroy(nccl);
    if (p2p) cora::p2p_destroy(p2p);
  }
};

static void native_comm_destroy(cora_native_comm *nc) { delete nc; }
static double *native_scalars(cora_native_comm *nc) { return nc->d_scal + 1016; }  // behind the host all-reduce's staging
static int native_allreduce_dev(cora_native_comm *nc, double *d, int n) { return nc->allreduce_dev(d, n); }
static int native_exchange_on(cora_native_comm *nc, double *dX, int ld, hipStream_t st) { return nc->exchange(dX, ld, st); }
static int native_product_pack(cora_native_comm *nc, const double *dX, int ld, hipStream_t st, double **slots) { return nc->product_pack(dX, ld, st, slots); }
static int native_product_gather(cora_native_comm *nc, double *dX, int ld, hipStream_t st, double *out, double *kappa,
                                 hipEvent_t after_collective) {
  return nc->product_gather(dX, ld, st, out, kappa, after_collective);
}
static const std::string &native_error(const cora_native_comm *nc) { return nc->err; }
static int native_allgather_rows(cora_native_comm *nc, double *dX, int ld, int64_t row0, int64_t nrows) { return nc->allgather_rows(dX, ld, row0, nrows); }

namespace {
int native_exchange_cb(void *u, double *dX, int ld) { return static_cast<cora_native_comm *>(u)->exchange(dX, ld); }
int native_allreduce_cb(void *u, double *vals, int n) { return static_cast<cora_native_comm *>(u)->allreduce_host(vals, n); }
int native_allgather_cb(void *u, double *dX, int ld) { return static_cast<cora_native_comm *>(u)->allgather(dX, ld); }

int native_finish(cora_ctx *c, cora_native_comm *nc) {
  if (nc->hip(hipMalloc(reinterpret_cast<void **>(&nc->d_scal), 1024 * sizeof(double)), "hipMalloc") ||
      nc->hip(hipHostMalloc(reinterpret_cast<void **>(&nc->h_scal), 1024 * sizeof(double)), "hipHostMalloc") || nc->plan()) {
    const std::string m = nc->err;
    delete nc;
    return fail(c, CORA_ERR_HIP, "native communication: " + m);
  }
  delete c->native_comm;
  c->native_comm = nc;
  c->comm_exchange = native_exchange_cb;
  c->comm_allreduce = native_allreduce_cb;
  c->comm_allgather = native_allgather_cb;
  c->comm_user = nc;
  return CORA_OK;
}
}  // namespace

extern "C" {

int cora_rccl_unique_id(void *id128) {
  if (!id128) return CORA_ERR_ARG;
  std::string err;
  const RcclApi *api = rccl_api(&err);
  if (!api) return fail(nullptr, CORA_ERR_HIP, err);
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return fail(nullptr, CORA_ERR_HIP, "ncclGetUniqueId failed");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return CORA_OK;
}

int cora_comm_create_rccl(cora_ctx *c, const void *id128) {
  NEED_DEVICE(c);
  if (!id128) return fail(c, CORA_ERR_ARG, "bad arguments");
  std::string err;
  const RcclApi *api = rccl_api(&err);
  if (!api) return fail(c, CORA_ERR_HIP, err);
  auto *nc = new cora_native_comm;
  nc->c = c;
  nc->rank = c->F.L.rank;
  nc->world = c->F.L.world;
  nc->api = api;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = api->CommInitRank(&nc->nccl, nc->world, id, nc->rank);
  if (r != ncclSuccess) {
    const std::string m = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
    nc->nccl = nullptr;
    delete nc;
    return fail(c, CORA_ERR_HIP, m);
  }
  // the communicator must be what the partition assumes: world ranks, this rank's number (a launcher that hands the
  // id to the wrong set of processes would otherwise gather shards in the wrong order, silently)
  if (api->CommCount && api->CommUserRank) {
    int cnt = -1, ur = -1;
    if (api->CommCount(nc->nccl, &cnt) != ncclSuccess || api->CommUserRank(nc->nccl, &ur) != ncclSuccess || cnt != nc->world || ur != nc->rank) {
      const std::string m = "RCCL communicator reports " + std::to_string(cnt) + " ranks / rank " + std::to_string(ur) +
                            ", the handle was partitioned for " + std::to_string(nc->world) + " / " + std::to_string(nc->rank);
      delete nc;
      return fail(c, CORA_ERR_ARG, m);
    }
  }
  return native_finish(c, nc);
}

int cora_comm_rccl_ranks(const cora_ctx *c, int out[2]) {
  if (!c || !out) return CORA_ERR_ARG;
  out[0] = out[1] = -1;
  const cora_native_comm *nc = c->native_comm;
  if (!nc || !nc->nccl || !nc->api || !nc->api->CommCount || !nc->api->CommUserRank) return CORA_OK;
  if (nc->api->CommCount(nc->nccl, &out[0]) != ncclSuccess || nc->api->CommUserRank(nc->nccl, &out[1]) != ncclSuccess) out[0] = out[1] = -1;
  return CORA_OK;
}

cora_local_group *cora_local_group_create(int world) {
  if (world < 1) return nullptr;
  auto *g = new cora_local_group;
  g->world = world;
  g->ptrs.assign(static_cast<size_t>(world), nullptr);
  g->vals.resize(static_cast<size_t>(world));
  return g;
}
void cora_local_group_destroy(cora_local_group *g) { delete g; }
void cora_local_group_abort(cora_local_group *g) {  // a rank gave up: nobody keeps waiting for it
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->m);
  g->broken = true;
  g->cv.notify_all();
}

int cora_comm_create_local(cora_ctx *c, cora_local_group *g) {
  NEED_DEVICE(c);
  if (!g || g->world != c->F.L.world) return fail(c, CORA_ERR_ARG, "the group's size is not the handle's world size");
  auto *nc = new cora_native_comm;
  nc->c = c;
  nc->rank = c->F.L.rank;
  nc->world = c->F.L.world;
  nc->g = g;
  return native_finish(c, nc);
}

// Device-side transport (p2p.h).  Two steps, because the peers' mailboxes must exist before anybody can map them:
//   cora_comm_p2p_handle  -> this rank's mailbox and its 128-byte export blob;
//   (the launcher gathers the blobs of all ranks in rank order: torch.distributed, MPI, a file -- like the RCCL id)
//   cora_comm_create_p2p  -> maps the peers and plans the exchange (the planning all-gathers already run on the mailboxes).
int cora_comm_p2p_handle(cora_ctx *c, void *blob128) {
  NEED_DEVICE(c);
  if (!blob128) return fail(c, CORA_ERR_ARG, "bad arguments");
  if (c->p2p_pending) {
    cora::p2p_destroy(c->p2p_pending);
    c->p2p_pending = nullptr;
  }
  std::string err;
  if (cora::p2p_create(c->device, c->F.L.rank, c->F.L.world, &c->p2p_pending, blob128, &err)) return fail(c, CORA_ERR_HIP, err);
  return CORA_OK;
}

int cora_comm_create_p2p(cora_ctx *c, const void *blobs) {
  NEED_DEVICE(c);
  if (!blobs) return fail(c, CORA_ERR_ARG, "bad arguments");
  if (!c->p2p_pending) return fail(c, CORA_ERR_NOT_READY, "cora_comm_p2p_handle was not called on this handle");
  std::string err;
  if (cora::p2p_connect(c->p2p_pending, blobs, &err)) return fail(c, CORA_ERR_HIP, err);
  auto *nc = new cora_native_comm;
  nc->c = c;
  nc->rank = c->F.L.rank;
  nc->world = c->F.L.world;
  nc->p2p = c->p2p_pending;
  c->p2p_pending = nullptr;
  return native_finish(c, nc);
}

int cora_comm_p2p_status(const cora_ctx *c, long out[6]) {
  if (!c || !out) return CORA_ERR_ARG;
  for (int i = 0; i < 6; ++i) out[i] = 0;
  out[3] = -1;
  const cora_native_comm *nc = c->native_comm;
  if (!nc || !nc->p2p) return CORA_OK;
  cora::p2p_status(nc->p2p, out);
  out[4] = nc->n_p2p_gather;
  out[5] = nc->n_p2p_reduce;
  return CORA_OK;
}

int cora_comm_native_enable(cora_ctx *c, int on) {
  if (!c || !c->native_comm) return fail(c, CORA_ERR_NOT_READY, "no native communication on this handle");
  c->comm_exchange = on ? native_exchange_cb : nullptr;
  c->comm_allreduce = on ? native_allreduce_cb : nullptr;
  c->comm_allgather = on ? native_allgather_cb : nullptr;
  c->comm_user = on ? c->native_comm : nullptr;
  return CORA_OK;
}

int cora_debug_spmm_window_min_slices(int min_slices) {
  const int old = g_win_min_slices;
  if (min_slices >= 0) g_win_min_slices = min_slices;
  return old;
}

int cora_debug_local_products(cora_ctx *c, int on) {
  if (!c) return CORA_ERR_ARG;
  c->local_products = on != 0;
  return CORA_OK;
}

int cora_debug_product_phases(cora_ctx *c, const double *dX, double *dOut, int epi, int reps, double us[5]) {
  NEED_DEVICE(c);
  NEED_RANK(c);
  if (!dX || !dOut || !us || reps < 1 || epi < EPI_NONE || epi > EPI_HVP) return fail(c, CORA_ERR_ARG, "bad arguments");
  // The call is collective (reps + 3 all-gathers), so whether it runs must not depend on anything rank-local (round-4
  // advice): product_one_collective() is the same on every rank (world, transport, the common list of long rows), the
  // interior / boundary overlap is not (it counts THIS rank's slices) -- the serial order is forced for the duration of
  // the call instead of being required.
  if (!product_one_collective(c))
    return fail(c, CORA_ERR_NOT_READY, "phase timing needs the library's own communication and distributed long rows");
  while (c->phase_events.size() < 6) {
    hipEvent_t e;
    HIP_TRY(c, hipEventCreate(&e));
    c->phase_events.push_back(e);
  }
  for (int k = 0; k < 5; ++k) us[k] = 0.0;
  const int overlap_saved = c->overlap_exchange;
  c->overlap_exchange = 0;
  int rc = CORA_OK;
  for (int it = 0; it < reps + 3 && !rc; ++it) {   // (three warm-up rounds)
    c->phase_timing = true;
    rc = apply_product(c, dX, c->ld, epi, dOut);
    c->phase_timing = false;
    if (rc) break;
    if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(c, CORA_ERR_HIP, "hipStreamSynchronize"); break; }
    if (it < 3) continue;
    for (int k = 0; k < 5 && !rc; ++k) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->phase_events[k], c->phase_events[k + 1]) != hipSuccess) rc = fail(c, CORA_ERR_HIP, "hipEventElapsedTime");
      us[k] += ms * 1e3 / reps;
    }
  }
  c->overlap_exchange = overlap_saved;
  return rc;
}

long long cora_comm_gathered_rows(const cora_ctx *c) { return (c && c->native_comm) ? c->native_comm->rows_gathered : 0; }

int cora_comm_counters(const cora_ctx *c, long out[2]) {
  if (!c || !out) return CORA_ERR_ARG;
  out[0] = c->native_comm ? c->native_comm->n_allgather : 0;
  out[1] = c->native_comm ? c->native_comm->n_allreduce : 0;
  return CORA_OK;
}

int cora_comm_overlap_enable(cora_ctx *c, int on) {
  if (!c) return CORA_ERR_ARG;
  if (on < 0 || on > 2) return fail(c, CORA_ERR_ARG, "overlap mode must be 0, 1 or 2");
  c->overlap_exchange = on;
  return CORA_OK;
}

int cora_comm_overlap_active(const cora_ctx *c) { return c && product_overlaps_exchange(c) ? 1 : 0; }

int64_t cora_comm_exchanged_rows(const cora_ctx *c) { return (c && c->native_comm) ? c->native_comm->exchanged_rows : 0; }

}  // extern "C"
