// Internal declarations shared by the format builder (host C++), the kernels
// (HIP) and the C-ABI layer.  Not installed; the public interface is
// include/cora_hip.h.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace cora {

constexpr int kWave = 64;         // CDNA4 wavefront
constexpr int kLongRow = 96;      // translation rows longer than this -> long path
constexpr int kLongChunk = 1024;  // default nnz per long-row chunk (one wavefront each)
extern int g_long_chunk;
extern int g_interleave;            // pose-slice slot order (see format_build.cpp)
constexpr int kSigma = 256;       // default sorting window (rows) for translation slices
extern int g_sigma;               // tunable copy of kSigma (format_build.cpp)
extern int g_chain_slices;        // pose slices in the chain layout (kSliceChainFlag, format_build.cpp); 0: plain layout
constexpr int kMaxLD = 24;

// Row stride (doubles) used for a k-column resident vector: the number of columns itself, whatever k (<= kMaxLD).  Odd
// strides cost 8-byte instead of 16-byte row accesses but save the padding traffic (Hvp at p = 5 measured 24.45 ->
// 23.28 us against a stride of 6); rounds 1-2 padded 13..24 columns to 16 / 20 / 24 -- the certificate block of rank
// p >= 11 has max(10, p + 2) >= 13 columns (src/CORA_problem.cpp:1062-1063) and paid up to 23 % of padding.
extern int g_pad_even;  // lab switch: 1 = round odd k up to even (format_build.cpp)
inline int ld_for(int k) {
  if (g_pad_even && k > 1) return (k + 1) & ~1;
  return k < 2 ? 2 : k;
}

enum SliceType : int32_t {
  kSliceStiefel = 0,   // lane = one pose (its d rotation rows); d x 1 column blocks
  kSliceOblique = 1,   // lane = one unit-sphere row
  kSliceEuclid = 2,    // lane = one translation row, identity row order
  kSliceEuclidPerm = 3 // lane = one translation row, rows given by perm[]
};
constexpr int32_t kSliceTypeMask = 0xff;
// Flag on a pose slice: the CHAIN layout.  The lane owns its pose's d rotation rows AND the pose's translation row, the
// columns every pose of a chain has are implied by the lane (no index stored), and everything Q's symmetry gives is
// stored once:
//   fixed slots (values only, [..][lane], in this order; P = the lane's local pose, t_P its translation row):
//     s0[a], a <= d : column t_P      -- Q(rot(P)_a, t_P) for a < d,  Q(t_P, t_P) for a = d
//     s1[a], a <= d : column t_{P+1}  -- Q(rot(P)_a, t_{P+1}),        Q(t_P, t_{P+1})       (zeros without a next local pose)
//     nxt[c][a]     : column rot(P+1)_c, rows rot(P)_a               (zeros without a next local pose)
//     own[c][a]     : column rot(P)_c,   rows rot(P)_a
//   taken from elsewhere (checked bit for bit when the format is built; a slice that fails keeps the plain layout and
//   its translation rows go to row slices):
//     Q(t_P, rot(P)_c)      = s0[c] of the same lane
//     Q(rot(P)_a, rot(P-1)_c) = nxt[a][c], Q(t_P, rot(P-1)_c) = s1[c], Q(t_P, t_{P-1}) = s1[d] of the lane BEFORE
//                             (lane 0: HostFormat::head_val, kChainHead(d) doubles per pose slice)
//   general slots (index + d values, SliceDesc::width of them): every other column of the rotation rows
//   tail (PAIRS of {index, value}, compact, sorted by lane; lane's range of pairs in tinfo = start | count << 16):
//     every other column of the translation row -- its range measurements (two entries each: the range row and the
//     landmark's translation), loop closures, a predecessor on another shard; an odd count is padded with a zero
// Value stream from SliceDesc::off: fixed (kChainFixed(d) x 64) | general (width x d x 64) | tail (T x 2);  index stream
// from SliceDesc::coff: tinfo (64) | general (width x 64) | tail (T x 2);  T = SliceDesc::type >> kSliceTailShift pairs.
constexpr int32_t kSliceChainFlag = 0x100;
// tinfo = start | count << 16 | nlocal << 24 (pairs; count, nlocal <= 127): the first nlocal pairs of a lane's tail have
// columns of the handle's own shard, the rest are rows another rank owns (partitioned handles sort them so).
// SliceDesc::nrows carries two launch-time flags above the lane count (capi.hip builds the lists of a partitioned
// handle's overlapped product with them): kSliceSkipRemoteTail -- the slice runs before the exchange has landed and
// leaves out the remote pairs of its tails; kSliceRemoteTailOnly -- the rest: out[t] += the remote pairs, nothing else.
constexpr int32_t kSliceRowsMask = 0x7f;
constexpr int32_t kSliceSkipRemoteTail = 0x100;
constexpr int32_t kSliceRemoteTailOnly = 0x200;
constexpr int kSliceTailShift = 16;
constexpr int kSliceTailMaxShift = 9;   // SliceDesc::type bits 9..15: the longest tail (in pairs) of a lane of the slice
constexpr int32_t kSliceTailMaxMask = 0x7f;
constexpr int kChainFixed(int d) { return 2 * (d + 1) + 2 * d * d; }   // doubles per lane in the fixed slots
constexpr int kChainHead(int d) { return d * d + d + 1; }              // doubles per slice in head_val

// One wavefront's work, stored slot-major ([k][lane]) so that every load is a
// fully coalesced 512 B (values) / 256 B (columns).
//  - row slices (types 1-3): lane = row, slot k = one nonzero:
//        col  = scol[coff + k*64 + lane],  val = sval[off + k*64 + lane]
//  - pose slices (type 0): lane = pose, slot k = one column of the union pattern
//    of the pose's d rows, carrying d values (a d x 1 block; zero where a row
//    does not have the column):
//        col  = scol[coff + k*64 + lane],  val_a = sval[off + (k*d + a)*64 + lane]
struct SliceDesc {
  int32_t row0;    // first internal row (or offset into perm[] for kSliceEuclidPerm)
  int32_t nrows;   // active lanes
  int32_t width;   // slots per lane
  int32_t type;    // SliceType
  int64_t off;     // element offset into sval
  int32_t coff;    // element offset into scol
  int32_t aux0;    // Stiefel: first LOCAL pose index; Oblique: first LOCAL range index
};
static_assert(sizeof(SliceDesc) == 32, "SliceDesc must be 32 bytes");

// A chunk of one long row, processed by one 256-thread workgroup.
struct LongChunk {
  int32_t row;       // internal row
  int32_t k0, k1;    // nonzero range in lval / lcol
  int32_t nchunks;   // chunks of this row
  int32_t first;     // index of this row's first chunk (partials slot base)
  int32_t slot;      // long-row ordinal (ticket counter index)
  int32_t pad0, pad1;
};
static_assert(sizeof(LongChunk) == 32, "LongChunk must be 32 bytes");

// Region of the internal row order owned by this handle.
struct Layout {
  int d = 0, n = 0, r = 0, nt = 0;  // global problem dims (nt = n + l)
  int64_t N = 0;                    // n*d + r + nt
  int rank = 0, world = 1;
  int64_t shard_rows = 0;           // padded rows per rank
  int64_t rows = 0;                 // world * shard_rows
  // local (owned) counts and internal bases
  int64_t base = 0;                 // rank * shard_rows
  int nl_poses = 0, nl_ranges = 0, nl_trans = 0;
  int64_t rot_base = 0, rng_base = 0, trn_base = 0;  // internal row of first local rot/range/trans row
  int64_t local_rows = 0;
};

struct HostFormat {
  Layout L;
  std::vector<int32_t> api2int;   // N: internal row of API row
  std::vector<int32_t> int2api;   // rows: API row of internal row (-1 = padding)
  std::vector<SliceDesc> slices;
  std::vector<SliceDesc> slices_pose_first;  // same slices, pose slices first inside each eighth (small row strides)
  // [pose slice][kChainHead(d)]: what lane 0 of a chain slice takes from the pose before it --
  // [a * d + c] = Q(rot(P)_a, rot(P-1)_c), [d * d + c] = Q(t_P, rot(P-1)_c), [d * d + d] = Q(t_P, t_{P-1})
  std::vector<double> head_val;
  std::vector<double> sval;
  std::vector<int32_t> scol;
  std::vector<int32_t> perm;      // internal rows for kSliceEuclidPerm slices
  std::vector<LongChunk> chunks;
  std::vector<int32_t> chunk_order;  // launch order of the chunks: by first column, so that chunks of
                                     // different long rows that read the same region of X run together
  std::vector<double> lval;
  std::vector<int32_t> lcol;
  int n_long_rows = 0;
  // partitioned handles: the long rows are DISTRIBUTED (format_build.cpp) -- the same list on every rank: internal row
  // and owner rank of long row j (LongChunk::slot = j); empty on one GPU
  std::vector<int32_t> long_rows, long_owner;
  std::vector<double> diag;       // diag(Q) for LOCAL rows, indexed by (internal row - base)
  int64_t nnz_global = 0, nnz_local = 0, padded_nnz = 0, long_nnz = 0;
  int max_width = 0;
};

// Builds the partition + sliced format.  Throws std::runtime_error on invalid input.
// distribute_long_rows (world > 1 only): see HostFormat::long_rows.
void build_format(int d, int n, int r, int nt, const int32_t *rowptr,
                  const int32_t *col, const double *val, int rank, int world,
                  HostFormat &out, bool distribute_long_rows = true);

// Every column index a slice stores (general slots and tail of a chain slice; all slots of the others); the implied
// columns of a chain slice are rows of the local shard.
void slice_columns(const HostFormat &F, const SliceDesc &sd, std::vector<int32_t> &out);

// Host execution of the FORMAT (test hook, see cora_debug_format_spmm_host).
void format_spmm_host(const HostFormat &F, const double *X_int, int ld,
                      double *out_int);

}  // namespace cora
