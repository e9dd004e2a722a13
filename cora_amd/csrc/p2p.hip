// Device-side collectives over peer-mapped mailboxes (p2p.h): kernels and the host side of the transport.
#include "p2p.h"

#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace cora {

namespace {

// ---- mailbox layout (every rank's is the same; offsets in bytes) ----------------------------------------------------
constexpr size_t kAgFlagOff = 0;                                                             // uint64 [2][world max][blocks]
constexpr size_t kArFlagOff = kAgFlagOff + 8ull * 2 * kP2PMaxWorld * kP2PMaxBlocks;          // uint64 [2][world max]
constexpr size_t kArDataOff = kArFlagOff + 8ull * 2 * kP2PMaxWorld;                          // double [2][world max][reduce max]
constexpr size_t kErrorOff = kArDataOff + 8ull * 2 * kP2PMaxWorld * kP2PReduceMax;           // uint64: timeouts raised here
constexpr size_t kHostErrOff = kErrorOff + 8;                                                // pointer to a pinned host word (this process's)
constexpr size_t kAgDataOff = (kHostErrOff + 8 + 4095) & ~static_cast<size_t>(4095);         // bytes [2][world][slot]

struct Peers {
  char *mail[kP2PMaxWorld];
};

__device__ __forceinline__ unsigned long long *ag_flag(char *m, int parity, int src, int blk) {
  return reinterpret_cast<unsigned long long *>(m + kAgFlagOff) + (static_cast<size_t>(parity) * kP2PMaxWorld + src) * kP2PMaxBlocks + blk;
}
__device__ __forceinline__ unsigned long long *ar_flag(char *m, int parity, int src) {
  return reinterpret_cast<unsigned long long *>(m + kArFlagOff) + static_cast<size_t>(parity) * kP2PMaxWorld + src;
}
__device__ __forceinline__ double *ar_data(char *m, int parity, int src) {
  return reinterpret_cast<double *>(m + kArDataOff) + (static_cast<size_t>(parity) * kP2PMaxWorld + src) * kP2PReduceMax;
}
__device__ __forceinline__ char *ag_data(char *m, int parity, int src, int world) {
  return m + kAgDataOff + (static_cast<size_t>(parity) * world + src) * kP2PSlotBytes;
}

// Spins until *flag >= seq (acquire, system scope).  A peer that never arrives (a dead rank) must not hang the device:
// after `timeout` ticks of the 100 MHz wall clock the error word of this rank's mailbox is raised and the wait ends.
__device__ __forceinline__ void wait_flag(unsigned long long *flag, unsigned long long seq, unsigned long long timeout, char *mine) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > timeout) {
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(mine + kErrorOff), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // (and in pinned host memory, where the host sees it without a synchronisation: the next collective call fails loudly)
      unsigned long long *h = *reinterpret_cast<unsigned long long **>(mine + kHostErrOff);
      if (h) __hip_atomic_fetch_add(h, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

// One all-gather (or one piece of a long one).  Block b hands over ITS slice of the payload and waits for the same slice
// of every peer: no block waits for another block of its own grid.  W: uint64_t or uint32_t words.
template <typename W>
__global__ __launch_bounds__(256) void k_p2p_allgather(Peers P, int rank, int world, const W *__restrict__ send, W *recv, size_t words,
                                                       size_t recv_stride, unsigned long long seq, unsigned long long timeout) {
  const int b = static_cast<int>(blockIdx.x), tid = static_cast<int>(threadIdx.x);
  const int parity = static_cast<int>(seq & 1);
  const size_t per = (words + gridDim.x - 1) / gridDim.x;
  const size_t w0 = per * b < words ? per * b : words, w1 = w0 + per < words ? w0 + per : words;
  char *mine = P.mail[rank];
  // push: my slice into slot [rank] of every mailbox (my own included: one code path, and the self copy is local)
  for (int q = 0; q < world; ++q) {
    W *dst = reinterpret_cast<W *>(ag_data(P.mail[q], parity, rank, world));
    for (size_t i = w0 + tid; i < w1; i += 256) __hip_atomic_store(dst + i, send[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < world) {
    __hip_atomic_store(ag_flag(P.mail[tid], parity, rank, b), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    wait_flag(ag_flag(mine, parity, tid, b), seq, timeout, mine);
  }
  __syncthreads();
  // deliver
  for (int q = 0; q < world; ++q) {
    const W *src = reinterpret_cast<const W *>(ag_data(mine, parity, q, world));
    W *out = recv + static_cast<size_t>(q) * recv_stride;
    for (size_t i = w0 + tid; i < w1; i += 256)
      out[i] = __hip_atomic_load(const_cast<W *>(src) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Sum of n <= kP2PReduceMax doubles over the ranks, in place, added in rank order (the same bits everywhere).
__global__ __launch_bounds__(64) void k_p2p_allreduce(Peers P, int rank, int world, double *d, int n, unsigned long long seq,
                                                      unsigned long long timeout) {
  const int tid = static_cast<int>(threadIdx.x);
  const int parity = static_cast<int>(seq & 1);
  char *mine = P.mail[rank];
  double v = 0.0;
  if (tid < n) v = d[tid];
  for (int q = 0; q < world; ++q)
    if (tid < n) __hip_atomic_store(ar_data(P.mail[q], parity, rank) + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (tid < world) {
    __hip_atomic_store(ar_flag(P.mail[tid], parity, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    wait_flag(ar_flag(mine, parity, tid), seq, timeout, mine);
  }
  __syncthreads();
  if (tid < n) {
    double s = 0.0;
    for (int q = 0; q < world; ++q) s += __hip_atomic_load(ar_data(mine, parity, q) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    d[tid] = s;
  }
}

// The exchange of a product in ONE kernel and ONE block (the payload is the chain's halo rows + one slot per landmark row:
// about a hundred doubles per rank): push this rank's [ e_max rows | n_long slots ] into every mailbox, hand over, wait, then
// unpack STRAIGHT FROM THE MAILBOX -- rows to X[recv_idx], and the owner of long row j adds slot j of every rank in rank order
// into its row of the result (the same operations as k_exchange_unpack in kernels.hip: the same bits as the other transports).
// export_rows != nullptr: the rows are taken from X itself (rows export_rows[0 .. e_max) of this rank's shard) and only the slots from
// `send` (which then points at the slots): no pack launch in front of this kernel.
__global__ __launch_bounds__(256) void k_p2p_exchange_unpack(Peers P, int rank, int world, const double *__restrict__ send,
                                                             const int32_t *__restrict__ export_rows, int64_t e_max,
                                                             int n_long, int ld, const int32_t *__restrict__ recv_idx, double *X,
                                                             const int32_t *__restrict__ long_rows, const int32_t *__restrict__ long_owner,
                                                             double *out, double *kappa, unsigned long long seq, unsigned long long timeout) {
  const int tid = static_cast<int>(threadIdx.x);
  const int parity = static_cast<int>(seq & 1);
  const int64_t stride = (e_max + n_long) * ld;
  char *mine = P.mail[rank];
  const int64_t per = e_max * ld;
  for (int64_t i = tid; i < stride; i += 256) {  // (a payload of a hundred doubles: one trip for most threads)
    double v;
    if (export_rows) {
      if (i < per) {
        const int64_t k = i / ld;
        v = X[static_cast<int64_t>(export_rows[k]) * ld + (i - k * ld)];
      } else {
        v = send[i - per];
      }
    } else {
      v = send[i];
    }
    for (int q = 0; q < world; ++q)
      __hip_atomic_store(reinterpret_cast<double *>(ag_data(P.mail[q], parity, rank, world)) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < world) {
    __hip_atomic_store(ag_flag(P.mail[tid], parity, rank, 0), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    wait_flag(ag_flag(mine, parity, tid, 0), seq, timeout, mine);
  }
  __syncthreads();
  const int64_t tot = per * world;
  for (int64_t t = tid; t < tot; t += 256) {
    const int64_t r = t / per, w = t - r * per, k = t / ld, j = t - k * ld;
    const double *src = reinterpret_cast<const double *>(ag_data(mine, parity, static_cast<int>(r), world));
    X[static_cast<int64_t>(recv_idx[k]) * ld + j] = __hip_atomic_load(const_cast<double *>(src) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // long rows: one wavefront per row, lanes < ld (a landmark's own row of X is this rank's: the scatter above does not write it)
  const int wave = tid >> 6, lane = tid & 63;
  for (int j = wave; j < n_long; j += 4) {
    double v = 0.0, k = 0.0;
    if (lane < ld)
      for (int r = 0; r < world; ++r)
        v += __hip_atomic_load(reinterpret_cast<double *>(ag_data(mine, parity, r, world)) + (e_max + j) * ld + lane, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    if (long_owner[j] == rank && lane < ld) {
      const size_t at = static_cast<size_t>(long_rows[j]) * ld + lane;
      out[at] = v;
      if (kappa) k = v * X[at];
    }
    if (kappa) {
      for (int off = 32; off > 0; off >>= 1) k += __shfl_xor(k, off, 64);
      if (lane == 0) kappa[j] = k;
    }
  }
}

struct Blob {  // kP2PBlobBytes
  hipIpcMemHandle_t handle;  // 64 bytes
  int64_t pid;
  uint64_t ptr;
  int32_t device, world, rank;
  uint32_t magic;
  char pad[kP2PBlobBytes - 64 - 8 - 8 - 16];
};
static_assert(sizeof(Blob) == kP2PBlobBytes, "blob size");
constexpr uint32_t kMagic = 0x50325043u;  // "CP2P"

}  // namespace

struct P2PState {
  int device = 0, rank = 0, world = 1;
  char *mine = nullptr;
  size_t bytes = 0;
  int mem_kind = 2;
  Peers peers{};
  std::vector<void *> opened;          // what hipIpcOpenMemHandle returned (closed on destroy)
  unsigned long long ag_seq = 0, ar_seq = 0;
  unsigned long long timeout_ticks = 0;
  long collectives = 0, kernels = 0;
  bool connected = false;
  unsigned long long *h_err = nullptr;  // pinned: timeouts raised by this rank's waiting kernels (read without a synchronisation)
};

static int set_err(std::string *err, const std::string &m) {
  if (err) *err = m;
  return 1;
}
static int hip_err(std::string *err, hipError_t e, const char *what) {
  return e == hipSuccess ? 0 : set_err(err, std::string(what) + ": " + hipGetErrorString(e));
}

int p2p_create(int device, int rank, int world, P2PState **out, void *blob_out, std::string *err) {
  if (!out || !blob_out || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world)
    return set_err(err, "p2p: bad arguments (at most " + std::to_string(kP2PMaxWorld) + " ranks)");
  if (hip_err(err, hipSetDevice(device), "hipSetDevice")) return 1;
  auto *s = new P2PState;
  s->device = device;
  s->rank = rank;
  s->world = world;
  s->bytes = kAgDataOff + 2ull * world * kP2PSlotBytes;
  const char *te = std::getenv("CORA_P2P_TIMEOUT_S");
  const double secs = te ? std::max(0.001, std::atof(te)) : 60.0;
  s->timeout_ticks = static_cast<unsigned long long>(secs * 1e8);  // wall_clock64: 100 MHz
  // Memory the peers write and this rank polls: uncached first (what RCCL's own flags and buffers live in on this part),
  // then fine-grained, then ordinary device memory (coherent at system scope through the atomics' cache policy only).
  Blob b;
  std::memset(&b, 0, sizeof(b));
  void *p = nullptr;
  const unsigned kinds[2] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained};
  for (int k = 0; k < 3 && !p; ++k) {
    const hipError_t e = k < 2 ? hipExtMallocWithFlags(&p, s->bytes, kinds[k]) : hipMalloc(&p, s->bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      continue;
    }
    if (world > 1 && hipIpcGetMemHandle(&b.handle, p) != hipSuccess) {  // this kind cannot be exported here: next one
      (void)hipGetLastError();
      (void)hipFree(p);
      p = nullptr;
      continue;
    }
    s->mem_kind = k;
  }
  if (!p) {
    delete s;
    return set_err(err, "p2p: no exportable device memory for the mailbox (hipExtMallocWithFlags / hipMalloc + hipIpcGetMemHandle)");
  }
  s->mine = static_cast<char *>(p);
  if (hip_err(err, hipMemset(p, 0, s->bytes), "hipMemset") ||
      hip_err(err, hipHostMalloc(reinterpret_cast<void **>(&s->h_err), sizeof(unsigned long long), hipHostMallocMapped), "hipHostMalloc")) {
    (void)hipFree(p);
    delete s;
    return 1;
  }
  *s->h_err = 0;
  {
    unsigned long long *dev_view = nullptr;  // the device's address of the pinned word, kept in the mailbox for the kernels
    if (hip_err(err, hipHostGetDevicePointer(reinterpret_cast<void **>(&dev_view), s->h_err, 0), "hipHostGetDevicePointer") ||
        hip_err(err, hipMemcpy(s->mine + kHostErrOff, &dev_view, sizeof(dev_view), hipMemcpyHostToDevice), "hipMemcpy") ||
        hip_err(err, hipDeviceSynchronize(), "hipDeviceSynchronize")) {
      (void)hipHostFree(s->h_err);
      (void)hipFree(p);
      delete s;
      return 1;
    }
  }
  b.pid = static_cast<int64_t>(getpid());
  b.ptr = reinterpret_cast<uint64_t>(p);
  b.device = device;
  b.world = world;
  b.rank = rank;
  b.magic = kMagic;
  std::memcpy(blob_out, &b, sizeof(b));
  *out = s;
  return 0;
}

int p2p_connect(P2PState *s, const void *blobs, std::string *err) {
  if (!s || !blobs) return set_err(err, "p2p: bad arguments");
  if (hip_err(err, hipSetDevice(s->device), "hipSetDevice")) return 1;
  const Blob *B = static_cast<const Blob *>(blobs);
  for (int q = 0; q < s->world; ++q) {
    if (B[q].magic != kMagic || B[q].world != s->world || B[q].rank != q)
      return set_err(err, "p2p: blob " + std::to_string(q) + " is not rank " + std::to_string(q) + "'s handle of a " +
                              std::to_string(s->world) + "-rank group (handles must be gathered in rank order)");
    if (q == s->rank) {
      s->peers.mail[q] = s->mine;
      continue;
    }
    if (B[q].pid == static_cast<int64_t>(getpid())) {  // a thread of this process: the pointer itself
      if (B[q].device != s->device) {
        const hipError_t e = hipDeviceEnablePeerAccess(B[q].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return hip_err(err, e, "hipDeviceEnablePeerAccess");
        (void)hipGetLastError();
      }
      s->peers.mail[q] = reinterpret_cast<char *>(B[q].ptr);
      continue;
    }
    void *p = nullptr;
    if (hip_err(err, hipIpcOpenMemHandle(&p, B[q].handle, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return 1;
    s->opened.push_back(p);
    s->peers.mail[q] = static_cast<char *>(p);
  }
  s->connected = true;
  return 0;
}

void p2p_destroy(P2PState *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  for (void *p : s->opened) (void)hipIpcCloseMemHandle(p);
  if (s->mine) (void)hipFree(s->mine);
  if (s->h_err) (void)hipHostFree(s->h_err);
  delete s;
}

// A timeout raised by an earlier collective of this rank (a peer did not arrive): whatever that collective delivered is garbage, so
// every later call fails instead of computing on (the pinned word is read without a synchronisation).
static int timed_out(const P2PState *s, std::string *err) {
  if (!s->h_err || *reinterpret_cast<volatile unsigned long long *>(s->h_err) == 0) return 0;
  return set_err(err, "p2p: a peer did not arrive within CORA_P2P_TIMEOUT_S (" + std::to_string(*s->h_err) +
                          " waits gave up): the results of that collective are invalid");
}

int p2p_allgather(P2PState *s, const void *send, void *recv, size_t bytes, hipStream_t st, std::string *err) {
  if (!s || !s->connected) return set_err(err, "p2p: not connected");
  if (timed_out(s, err)) return 1;
  if (bytes % 4 != 0) return set_err(err, "p2p all-gather: the payload must be a multiple of 4 bytes");
  ++s->collectives;
  const bool wide = bytes % 8 == 0 && reinterpret_cast<uintptr_t>(send) % 8 == 0 && reinterpret_cast<uintptr_t>(recv) % 8 == 0;
  if (bytes == 0) return 0;
  for (size_t off = 0; off < bytes; off += kP2PSlotBytes) {
    const size_t piece = std::min(kP2PSlotBytes, bytes - off);
    const int blocks = static_cast<int>(std::max<size_t>(1, std::min<size_t>(kP2PMaxBlocks, (piece + 8191) / 8192)));
    const unsigned long long seq = ++s->ag_seq;
    ++s->kernels;
    if (wide)
      hipLaunchKernelGGL(k_p2p_allgather<uint64_t>, dim3(blocks), dim3(256), 0, st, s->peers, s->rank, s->world,
                         reinterpret_cast<const uint64_t *>(static_cast<const char *>(send) + off),
                         reinterpret_cast<uint64_t *>(static_cast<char *>(recv) + off), piece / 8, bytes / 8, seq, s->timeout_ticks);
    else
      hipLaunchKernelGGL(k_p2p_allgather<uint32_t>, dim3(blocks), dim3(256), 0, st, s->peers, s->rank, s->world,
                         reinterpret_cast<const uint32_t *>(static_cast<const char *>(send) + off),
                         reinterpret_cast<uint32_t *>(static_cast<char *>(recv) + off), piece / 4, bytes / 4, seq, s->timeout_ticks);
    if (hip_err(err, hipGetLastError(), "k_p2p_allgather")) return 1;
  }
  return 0;
}

bool p2p_exchange_unpack_fits(const P2PState *s, int64_t e_max, int n_long, int ld) {
  return s && static_cast<size_t>((e_max + n_long) * ld) * sizeof(double) <= std::min<size_t>(kP2PSlotBytes, 32 * 1024);
}

int p2p_exchange_unpack(P2PState *s, const double *send, const int32_t *export_rows, int64_t e_max, int n_long, int ld, const int32_t *recv_idx, double *X,
                        const int32_t *long_rows, const int32_t *long_owner, double *out, double *kappa, hipStream_t st, std::string *err) {
  if (!s || !s->connected) return set_err(err, "p2p: not connected");
  if (!p2p_exchange_unpack_fits(s, e_max, n_long, ld)) return set_err(err, "p2p: the exchange does not fit one block");
  if (timed_out(s, err)) return 1;
  ++s->collectives;
  ++s->kernels;
  const unsigned long long seq = ++s->ag_seq;
  hipLaunchKernelGGL(k_p2p_exchange_unpack, dim3(1), dim3(256), 0, st, s->peers, s->rank, s->world, send, export_rows, e_max, n_long, ld, recv_idx, X,
                     long_rows, long_owner, out, kappa, seq, s->timeout_ticks);
  return hip_err(err, hipGetLastError(), "k_p2p_exchange_unpack");
}

int p2p_allreduce(P2PState *s, double *d, int n, hipStream_t st, std::string *err) {
  if (!s || !s->connected) return set_err(err, "p2p: not connected");
  if (timed_out(s, err)) return 1;
  ++s->collectives;
  for (int at = 0; at < n; at += kP2PReduceMax) {
    const unsigned long long seq = ++s->ar_seq;
    ++s->kernels;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(64), 0, st, s->peers, s->rank, s->world, d + at, std::min(kP2PReduceMax, n - at), seq,
                       s->timeout_ticks);
    if (hip_err(err, hipGetLastError(), "k_p2p_allreduce")) return 1;
  }
  return 0;
}

void p2p_status(const P2PState *s, long out[4]) {
  out[0] = out[1] = out[2] = 0;
  out[3] = -1;
  if (!s) return;
  out[0] = s->collectives;
  out[1] = s->kernels;
  unsigned long long e = 0;
  (void)hipSetDevice(s->device);
  (void)hipMemcpy(&e, s->mine + kErrorOff, sizeof(e), hipMemcpyDeviceToHost);  // (synchronises: a status call, not the data path)
  out[2] = static_cast<long>(e);
  out[3] = s->mem_kind;
}

}  // namespace cora
