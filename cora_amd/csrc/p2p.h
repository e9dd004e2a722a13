// Device-side collectives of a partitioned handle WITHOUT RCCL (mitigation 3; round-5 review item 4).
//
// The exchange of a product is under 1 KB per rank (the chain's halo rows + one slot per landmark row), the reductions of
// an STPCG iteration are one to three doubles: an RCCL collective (12-25 us of protocol and launch) costs more than the
// 4-10 us kernel it feeds.  Here every rank owns a MAILBOX in its own device memory (fine-grained / uncached, exported with
// hipIpcGetMemHandle and mapped by every peer -- the same mechanism RCCL's own transport buffers use); a collective is
// ONE kernel on the handle's stream:
//     push    every rank writes its payload into slot [its rank] of EVERY peer's mailbox (stores over xGMI / the local
//             fabric), fences at system scope and then sets flag [its rank] there to the collective's sequence number
//             (release, system scope) -- a 1-to-1 hand-off per (receiver, sender) pair at distinct addresses, not one
//             hot word;
//     wait    the same kernel spins on the flags of its OWN mailbox (local memory) until every sender's sequence number
//             has arrived (acquire), with a wall-clock timeout that raises an error word instead of hanging the GPU (the
//             word is also counted in pinned host memory: the rank's next collective call fails loudly);
//     deliver it copies the gathered payloads out of the mailbox (all-gather), or adds them IN RANK ORDER (all-reduce:
//             the same bits on every rank, like the other transports).
// No host involvement, no stream synchronisation, no RCCL call on the data path.  Slots are double-buffered by the
// parity of the sequence number: a rank can be at most one collective ahead of a peer (it cannot finish collective k + 1
// before the peer has pushed k + 1, which the peer does only after it has delivered k), so parity k + 2 is free again.
//
// Ranks may be processes (one per GPU, or several sharing one GPU -- how the transport is tested on the 1-GPU box) or
// threads of one process (the export blob then carries the pointer itself).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace cora {

constexpr int kP2PMaxWorld = 16;
constexpr int kP2PMaxBlocks = 32;              // blocks of one all-gather kernel (each hands its own slice over)
constexpr size_t kP2PSlotBytes = 256 * 1024;   // payload per (sender, parity); longer all-gathers go in pieces
constexpr int kP2PReduceMax = 64;              // doubles per all-reduce call (longer ones go in pieces)
constexpr int kP2PBlobBytes = 128;             // what cora_comm_p2p_handle returns: IPC handle + pid + pointer + device

struct P2PState;

// (1) this rank's mailbox; blob (kP2PBlobBytes) is what the peers need to map it
int p2p_create(int device, int rank, int world, P2PState **out, void *blob, std::string *err);
// (2) all blobs in rank order -> peers mapped, ready for collectives
int p2p_connect(P2PState *s, const void *blobs, std::string *err);
void p2p_destroy(P2PState *s);

// all-gather of `bytes` (a multiple of 4) per rank: recv[q * bytes ...] = rank q's send.  recv may alias send's place in it.
int p2p_allgather(P2PState *s, const void *send, void *recv, size_t bytes, hipStream_t st, std::string *err);
// The exchange of a product, fused: send = this rank's [ e_max rows | n_long slots ] (ld doubles each); one kernel pushes it,
// waits for the peers' and unpacks from the mailbox -- rows to X[recv_idx[world * e_max]], every long row's slots added in rank
// order into the owner's row of `out` (kappa: the rows' shares of <X, out>, one per long row, or nullptr).
// export_rows != nullptr: the e_max rows are read from X itself (X[export_rows[k]]) and `send` holds the n_long slots only --
// no pack launch in front of the kernel: a product is chunks | exchange | slices.
bool p2p_exchange_unpack_fits(const P2PState *s, int64_t e_max, int n_long, int ld);
int p2p_exchange_unpack(P2PState *s, const double *send, const int32_t *export_rows, int64_t e_max, int n_long, int ld, const int32_t *recv_idx, double *X,
                        const int32_t *long_rows, const int32_t *long_owner, double *out, double *kappa, hipStream_t st, std::string *err);
// sum over the ranks of n device doubles, in place, added in rank order
int p2p_allreduce(P2PState *s, double *d, int n, hipStream_t st, std::string *err);

// [0] collectives issued, [1] kernels launched for them, [2] timeouts raised by a waiting kernel (0 unless a rank died),
// [3] kind of memory the mailbox lives in: 0 uncached, 1 fine-grained, 2 ordinary device memory
void p2p_status(const P2PState *s, long out[4]);

}  // namespace cora
