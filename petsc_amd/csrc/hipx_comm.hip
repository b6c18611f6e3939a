// hipx_comm.hip -- multi-GPU leg of the hot path: MPIAIJ ghost exchange and scalar all-reduces on RCCL.
//
// One process per GPU.  The reference does the exchange with persistent MPI send/recv inside PetscSF
// (sfbasic.c:25-49,90-91; VecScatterBegin/End vscat.c:1294,1353) and the reductions with MPI_Allreduce
// (pvecimpl.h:97-175).  Here:
//   * pack kernel (gather x[send_idx] -> contiguous send buffer) on the compute stream,
//   * ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the comm stream, receiving straight into
//     lvec (lvec[k] <-> garray[k], mmaij.c:108-117) -- xGMI is point-to-point, a 1-D row-slab partition
//     talks to <= 2 neighbours (2 of the 7 links),
//   * the diagonal-block SpMV runs on the compute stream meanwhile; an event makes the off-diagonal
//     MatMultAdd wait for the receive (the mpiaij.c:1056-1059 shape with stream-level overlap).
#include "hipx_internal.h"
#include "hipx_reduce.h"
#include "hipx_ipc.h"
#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace hipx;

namespace {

struct Comm {
  bool       active = false;
  ncclComm_t comm   = nullptr;  // ghost exchange, comm stream
  ncclComm_t rcomm  = nullptr;  // scalar all-reduces, compute stream
  int        rank = 0, nranks = 1;
  double    *d_red = nullptr;  // all-reduce staging
  double    *d_gather = nullptr;  // compensated mode over RCCL: the ranks' (hi, lo) pairs, all-gathered, folded in rank order
  double    *h_red = nullptr;  // pinned
  // IPC transport of the scalar all-reduces (hipxCommIpcExport / Attach): every rank stores its partial sums into every peer's
  // arena and publishes a sequence number; each rank then adds the nranks contributions in rank order (bitwise the same sum on
  // every rank).  Double-buffered by sequence parity: a rank can be at most one reduction ahead of a peer that still reads.
  bool                ipc = false;
  char               *arena = nullptr;         // fine-grained: [flags[nranks] | in[nranks][2][64]]
  size_t              hdr = 0;
  std::vector<char *> peer;                    // arenas of all ranks as mapped here
  char              **d_peer = nullptr;
  std::vector<void *> opened;
  unsigned long long  seq = 0;
  unsigned int       *d_err = nullptr;  // device alias of h_err
  unsigned int       *h_err = nullptr;  // pinned, host-mapped: a wait that gave up (read by the host after every reduction it waits for)
  // split-phase all-reduce (round 6): one may be in flight between its Begin and its End
  double             *d_red2 = nullptr;     // its staging line (d_red stays free for the blocking chains)
  int                 sp_n = 0;             // sums of the reduction in flight (0: none)
  bool                sp_pairs = false;
  hipStream_t         sp_stream = nullptr;  // RCCL: the collective runs here, between two events, while the compute stream goes on
  hipEvent_t          sp_ev_begin = nullptr, sp_ev_done = nullptr;
};
Comm &cm()
{
  static Comm c;
  return c;
}

#define HIPX_NCCL(call) \
  do { \
    ncclResult_t r_ = (call); \
    if (r_ != ncclSuccess) return hipx::fail(HIPX_ERR_GPU, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// What travels to the neighbours.  Plain: x[idx].  CG (round 6, hipxMatMultMPICGDirectionDotBegin): the NEW direction p = (z * dconst) + b x[idx] formed on the
// way out (cg.c:248-249) -- the operations, operands and order of cg_aypx_axpy_kernel / spmv_march2_kernel's prologue, hence the bits the owner's product kernel
// forms for the same elements a moment later; b = *dev_beta_new / *dev_beta_old when the sums are device-resident (launch-ahead loop), else the argument.
struct PackCG {
  const double *z = nullptr;
  double        dconst = 1.0, b = 0.0;
  const double *dev_beta_new = nullptr, *dev_beta_old = nullptr;
};
template <bool CG>
__device__ __forceinline__ double halo_value(const double *__restrict__ x, hipx_int i, const PackCG &cg, double b)
{
  if (!CG) return x[i];
  const double zv = cg.z[i] * cg.dconst;
  return zv + b * x[i];
}
template <bool CG>
__global__ void pack_kernel(const double *__restrict__ x, const hipx_int *__restrict__ idx, double *__restrict__ buf, hipx_int n, const PackCG cg)
{
  const double b = CG ? (cg.dev_beta_new ? (*cg.dev_beta_new / *cg.dev_beta_old) : cg.b) : 0.0;
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (hipx_int)gridDim.x * blockDim.x) buf[i] = halo_value<CG>(x, idx[i], cg, b);
}

// ---- IPC transport kernels.  Flags live in fine-grained memory and are accessed with system-scope atomics; payload writes
// are released with __threadfence_system() before the sequence number is stored (the R1 shape of cdna_hip_programming.md
// Guideline 16, at system scope because the reader is another process / another GPU).
// How long a rank waits for a peer before it gives up (ticks of the 100 MHz wall clock).  MPI semantics allow arbitrary skew
// between collective calls (first-call format builds, rank-0 I/O, a debugger), so the default is generous: 300 s;
// HIPX_IPC_WAIT_SECONDS changes it (0 = wait for ever).  The error word lives in pinned host-mapped memory: the host reads
// it after every reduction / exchange it waits for and turns a timeout into HIPX_ERR_GPU instead of returning stale sums.
static long long ipc_wait_ticks()
{
  static const long long t = [] {
    const char *e = getenv("HIPX_IPC_WAIT_SECONDS");
    const double s = e ? atof(e) : 300.0;
    return s <= 0.0 ? (long long)0x7fffffffffffffffLL : (long long)(s * 1e8);
  }();
  return t;
}

// Round 6: ONE launch sends to every neighbour (blockIdx.y = the neighbour's segment), on the COMPUTE stream in front of the product: a workgroup waits until
// the receiver has consumed the buffer of exchange seq - 2, gathers x[idx[k]] (CG: forms the new direction on the way, PackCG) and stores PAIRS of doubles
// straight into the neighbour's ghost buffer as 16-byte write-through stores; the wave drains them, a ticket collects the segment's workgroups, the last one
// raises the neighbour's sequence flag (hipx_ipc.h: no fences).  The stores are posted: over xGMI the wire time passes while the product kernel runs behind
// this kernel on the same stream -- the overlap of mpiaij.c:1056-1058 without a second kernel that fights the product for registers (round 5's put kernel
// on the comm stream could not become resident beside a 512-workgroup product that owns every SIMD's register file: it ran when the product had finished).
struct PutSeg {
  const hipx_int           *idx;
  hipx_int                  n;
  double                   *dst;
  const unsigned long long *ack;
  unsigned long long        need_ack;
  unsigned long long       *flag;
  unsigned int             *ticket;
};
struct PutArgs {
  PutSeg seg[4];
};
template <bool CG>
__global__ __launch_bounds__(256) void ipc_put_kernel(const double *__restrict__ x, const PutArgs args, unsigned long long seq, unsigned int *err, long long limit, const PackCG cg)
{
  __shared__ int ok;
  const PutSeg  &sg = args.seg[blockIdx.y];
  const hipx_int n2 = sg.n >> 1;
  if ((hipx_int)blockIdx.x * 256 >= (n2 ? n2 : 1)) return;  // (segments shorter than the longest one: no workgroup, no ticket -- the host counts the same way)
  if (threadIdx.x == 0) ok = ipc_wait_ge(sg.ack, sg.need_ack, err, limit) ? 1 : 0;  // the buffer of exchange seq - 2 has been consumed
  __syncthreads();
  const double b = CG ? (cg.dev_beta_new ? (*cg.dev_beta_new / *cg.dev_beta_old) : cg.b) : 0.0;
  if (ok) {
    const bool     al = (reinterpret_cast<uintptr_t>(sg.dst) & 15) == 0;
    const hipx_int T  = (hipx_int)gridDim.x * 256;
    constexpr int  U  = 4;  // pairs in flight per thread (few workgroups -- few arrivals on the ticket word -- with deep loads instead of many shallow ones)
    for (hipx_int q0 = (hipx_int)blockIdx.x * 256 + threadIdx.x; q0 < n2; q0 += U * T) {
      hipx_int i0[U], i1[U];
      double   v0[U], v1[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int q = q0 + u * T;
        i0[u] = q < n2 ? sg.idx[2 * q] : 0;
        i1[u] = q < n2 ? sg.idx[2 * q + 1] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        v0[u] = halo_value<CG>(x, i0[u], cg, b);
        v1[u] = halo_value<CG>(x, i1[u], cg, b);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int q = q0 + u * T;
        if (q < n2) {
          if (al) ipc_store16(sg.dst + 2 * q, v0[u], v1[u]);
          else {
            ipc_store8(sg.dst + 2 * q, v0[u]);
            ipc_store8(sg.dst + 2 * q + 1, v1[u]);
          }
        }
      }
    }
    if ((sg.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) ipc_store8(sg.dst + sg.n - 1, halo_value<CG>(x, sg.idx[sg.n - 1], cg, b));
  }
  ipc_drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nwg = (unsigned)(((n2 ? n2 : 1) + 255) / 256) < gridDim.x ? (unsigned)(((n2 ? n2 : 1) + 255) / 256) : gridDim.x;  // workgroups of this segment that got past the early return
    const unsigned t   = __hip_atomic_fetch_add(sg.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nwg - 1) {
      __hip_atomic_store(sg.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sg.flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void ipc_wait_kernel(const unsigned long long *data_seq, const int *recv_ranks, int nrecv, unsigned long long seq, unsigned int *err, long long limit)
{
  if ((int)threadIdx.x < nrecv) ipc_wait_ge(data_seq + recv_ranks[threadIdx.x], seq, err, limit);
}

__global__ void ipc_ack_kernel(unsigned long long *const *ack_ptrs, int nrecv, unsigned long long seq)
{
  if ((int)threadIdx.x < nrecv) __hip_atomic_store(ack_ptrs[threadIdx.x], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the kernels that read the ghost values have finished: stream order)
}


// pairs = 1 (compensated mode): vals holds n unrounded (hi, lo) pairs -- 2n words travel -- and every rank folds the nranks pairs of
// each sum in rank order with TwoSum, rounding hi + lo once: the same bits on every rank and for every way of cutting the rows.
// Round 6: the kernel also carries what used to be two more launches around it (5 us each on the critical path of every reduction) -- in front, the
// acknowledgement of the ghost buffers the kernels before it have consumed (ipc_ack_kernel); behind, the publication of the reduced values to the host slot
// and to device memory for the kernels queued ahead (red_signal_kernel).
struct PostAck {
  unsigned long long *const *ptrs = nullptr;  // per receive neighbour: &peer.ack_seq[me]
  int                        n    = 0;
  unsigned long long         seq  = 0;
};
struct PostSignal {
  unsigned long long *flag = nullptr;  // nullptr: no publication
  unsigned long long  seq  = 0;
  double             *results = nullptr, *dres = nullptr;
};
// phase (round 6, the split-phase form: hipxPipeCGUpdateBeginAllreduce ... hipxAllreduceEnd = PetscCommSplitReductionBegin ... PetscSplitReductionEnd, comb.c:168-290):
// 1 = POST only (acknowledgement, this rank's words into every peer's arena, the sequence flags), 2 = FINISH only (wait for the peers' flags, fold, publish),
// 3 = both in one launch.  Between the two launches of one reduction the stream may run anything that is not another reduction.
__global__ __launch_bounds__(64) void ipc_allreduce_kernel(double *vals, int nsums, int pairs, int me, int nranks, char *const *peer, size_t hdr, unsigned long long seq, unsigned int *err,
                                                           long long limit, const PostAck ack, const PostSignal sig, const int phase)
{
  const int t = threadIdx.x, q = (int)(seq & 1);
  const int n = pairs ? 2 * nsums : nsums;
  if (phase & 1) {
    if (t < ack.n) __hip_atomic_store(ack.ptrs[t], ack.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t < n) {
      const double v = vals[t];
      for (int p = 0; p < nranks; p++) {
        double *in = reinterpret_cast<double *>(peer[p] + hdr) + ((size_t)me * 2 + q) * 64;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(in + t), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __threadfence_system();
    __syncthreads();
    if (t < nranks) __hip_atomic_store(reinterpret_cast<unsigned long long *>(peer[t]) + me, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!(phase & 2)) return;
  }
  if (t < nranks) ipc_wait_ge(reinterpret_cast<const unsigned long long *>(peer[me]) + t, seq, err, limit);
  __threadfence_system();
  __syncthreads();
  if (pairs) {
    double res = 0.0;
    if (t < nsums) {
      Acc<true> acc;
      for (int p = 0; p < nranks; p++) {
        const double *in = reinterpret_cast<const double *>(peer[me] + hdr) + ((size_t)p * 2 + q) * 64;
        const double  hi = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(in + 2 * t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        const double  lo = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(in + 2 * t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        acc.merge(hi, lo);
      }
      res = acc.s + acc.c;
    }
    __syncthreads();  // every pair of vals[] has been read (by the stores above) before the sums overwrite the front of it
    if (t < nsums) vals[t] = res;
  } else if (t < n) {
    double sum = 0.0;
    for (int p = 0; p < nranks; p++) {
      const double *in = reinterpret_cast<const double *>(peer[me] + hdr) + ((size_t)p * 2 + q) * 64;
      sum += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(in + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
    vals[t] = sum;
  }
  if (sig.flag) {
    __syncthreads();
    if (t < nsums) {
      const double v = vals[t];
      sig.results[t] = v;  // pinned, host-mapped
      if (sig.dres) sig.dres[t] = v;
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(sig.flag, sig.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

// IPC transport: what a rank publishes about its receive side (one blob per rank, all-gathered by the host: MPI_Allgather in
// the PETSc plugin, torch.distributed in bench.py)
constexpr int IPC_MAXN = 60;
struct IpcBlob {
  hipIpcMemHandle_t handle;  // the receive arena
  int               rank, nrecv;
  long long         nghost;
  int               recv_ranks[IPC_MAXN];
  hipx_int          recv_off[IPC_MAXN + 1];
};
static_assert(sizeof(IpcBlob) <= HIPX_HALO_IPC_BLOB_BYTES, "IPC blob size");

struct hipxHalo_s {
  int                   nsend = 0, nrecv = 0;
  std::vector<int>      send_ranks, recv_ranks;
  std::vector<hipx_int> send_off, recv_off;
  hipx_int             *d_send_idx = nullptr;
  double               *d_sendbuf  = nullptr;
  hipEvent_t            ev_packed = nullptr, ev_done = nullptr;
  // ---- IPC transport (peer stores; see hipxHaloIpcExport)
  bool                  ipc = false;
  int                   me = -1, nranks = 0;
  unsigned long long    seq = 0;            // exchanges started
  char                 *arena = nullptr;    // fine-grained device memory: [data_seq[nranks] | ack_seq[nranks] | ghost buffer 0 | ghost buffer 1]
  size_t                hdr_bytes = 0, nghost = 0;
  std::vector<char *>   send_base, recv_base;  // neighbours' arenas as mapped here (own arena for a self-exchange)
  std::vector<hipx_int> send_peer_off;         // where my values start in that neighbour's ghost buffer
  std::vector<long long> send_peer_nghost;     // length of that neighbour's ghost buffer (stride between its two buffers)
  std::vector<void *>   opened;                // hipIpcOpenMemHandle results (closed at destroy)
  unsigned int         *d_ticket = nullptr;    // one per send neighbour
  int                  *d_recv_ranks = nullptr;
  unsigned long long  **d_ack_ptrs = nullptr;  // per recv neighbour: &peer.ack_seq[me]
  unsigned int         *d_err = nullptr;       // device alias of h_err
  unsigned int         *h_err = nullptr;       // pinned, host-mapped; checked at the next call (no sync on the data path)
  bool                  wait_pending = false;  // hipxHaloEnd has not launched the wait kernel: the consumer kernel waits itself (halo_wait_args)
  double               *ghost_cur = nullptr;   // ghost values of the exchange in progress / last completed
};


// the ranks' (hi, lo) pairs of nsums sums (gathered rank by rank) -> the nsums correctly rounded totals, in rank order
__global__ __launch_bounds__(64) void dd_fold_ranks_kernel(const double *gathered, int nsums, int nranks, double *vals)
{
  const int t = threadIdx.x;
  if (t >= nsums) return;
  Acc<true> acc;
  for (int p = 0; p < nranks; p++) acc.merge(gathered[(size_t)p * 2 * nsums + 2 * t], gathered[(size_t)p * 2 * nsums + 2 * t + 1]);
  vals[t] = acc.s + acc.c;
}

// in-place sum of n doubles in device memory over all ranks, enqueued on the compute stream.  Plain mode: n <= 64 sums.  Compensated
// mode (pairs): d_vals holds n <= 32 unrounded (hi, lo) pairs on entry (what the local kernels leave when RedOut::pairs is set) and the
// n rounded totals on exit; the pairs of all ranks are folded in rank order (IPC: inside the kernel; RCCL: all-gather + a fold kernel).
static int allreduce_dev(double *d_vals, int n, bool pairs = false, const PostAck *ack = nullptr, const PostSignal *sig = nullptr)
{
  Comm &c = cm();
  int   ierr;
  if ((ierr = prof_section(HIPX_PROF_ALLREDUCE, true, rt().compute))) return ierr;
  if (c.ipc) {
    ipc_allreduce_kernel<<<1, 64, 0, rt().compute>>>(d_vals, n, pairs ? 1 : 0, c.rank, c.nranks, c.d_peer, c.hdr, ++c.seq, c.d_err, ipc_wait_ticks(), ack ? *ack : PostAck{}, sig ? *sig : PostSignal{}, 3);
    HIPX_LAUNCH_CHECK();
  } else if (pairs) {
    HIPX_NCCL(ncclAllGather(d_vals, c.d_gather, (size_t)(2 * n), ncclDouble, c.rcomm, rt().compute));
    dd_fold_ranks_kernel<<<1, 64, 0, rt().compute>>>(c.d_gather, n, c.nranks, d_vals);
    HIPX_LAUNCH_CHECK();
  } else HIPX_NCCL(ncclAllReduce(d_vals, d_vals, (size_t)n, ncclDouble, ncclSum, c.rcomm, rt().compute));
  return prof_section(HIPX_PROF_ALLREDUCE, false, rt().compute);
}
// all-reduce of c.d_red[0 .. n) on the stream, then the values to the slot's host line + sequence flag and (dres) to device memory.  IPC transport: ONE kernel,
// which (h != NULL) first acknowledges the ghost buffers of h's exchange to their senders; RCCL: ncclAllReduce + red_signal_kernel (the release is the caller's)
static int halo_release(hipxHalo h);
static int allreduce_signal(double *d_vals, int n, bool pairs, int slot, double *dres, hipxHalo h);
static int matmult_mpi(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y, bool release);
// what the local kernels of the all-reduce chains leave in c.d_red: sums, or (compensated mode) unrounded pairs
static inline bool red_pairs() { return rt().red_exact != 0; }

// the IPC all-reduce kernel gave up on a peer (wait limit): the sums it produced are not sums -- fail loudly
static int comm_err_check()
{
  Comm &c = cm();
  if (c.ipc && c.h_err && *reinterpret_cast<volatile unsigned int *>(c.h_err))
    return fail(HIPX_ERR_GPU, "all-reduce (IPC): a peer rank did not contribute within the wait limit (HIPX_IPC_WAIT_SECONDS); the result is invalid", __FILE__, __LINE__);
  return HIPX_SUCCESS;
}

extern "C" {

int hipxCommIpcExport(int rank, int nranks, void *handle64)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(!c.active && nranks >= 1 && nranks <= 64 && rank >= 0 && rank < nranks && handle64, "bad arguments (at most 64 ranks)");
  c.rank   = rank;
  c.nranks = nranks;
  c.hdr    = ((size_t)8 * nranks + 255) & ~(size_t)255;
  const size_t bytes = c.hdr + sizeof(double) * 64 * 2 * (size_t)nranks;
  HIPX_HIP(hipExtMallocWithFlags((void **)&c.arena, bytes, hipDeviceMallocFinegrained));
  HIPX_HIP(hipMemset(c.arena, 0, bytes));
  hipIpcMemHandle_t h;
  HIPX_HIP(hipIpcGetMemHandle(&h, c.arena));
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle size");
  memset(handle64, 0, 64);
  memcpy(handle64, &h, sizeof(h));
  return HIPX_SUCCESS;
}

int hipxCommIpcAttach(const void *all_handles)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.arena && all_handles && !c.active, "hipxCommIpcExport() first");
  c.peer.assign((size_t)c.nranks, nullptr);
  for (int r = 0; r < c.nranks; r++) {
    if (r == c.rank) c.peer[(size_t)r] = c.arena;
    else {
      hipIpcMemHandle_t h;
      memcpy(&h, (const char *)all_handles + 64 * (size_t)r, sizeof(h));
      void *q = nullptr;
      HIPX_HIP(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
      c.opened.push_back(q);
      c.peer[(size_t)r] = (char *)q;
    }
  }
  HIPX_HIP(hipMalloc((void **)&c.d_peer, sizeof(char *) * (size_t)c.nranks));
  HIPX_HIP(hipMemcpy(c.d_peer, c.peer.data(), sizeof(char *) * (size_t)c.nranks, hipMemcpyHostToDevice));
  HIPX_HIP(hipHostMalloc((void **)&c.h_err, sizeof(unsigned int), hipHostMallocMapped));
  *c.h_err = 0;
  HIPX_HIP(hipHostGetDevicePointer((void **)&c.d_err, c.h_err, 0));
  HIPX_HIP(hipMalloc((void **)&c.d_red, sizeof(double) * 64));
  HIPX_HIP(hipHostMalloc((void **)&c.h_red, sizeof(double) * 64, hipHostMallocDefault));
  c.ipc    = true;
  c.active = true;
  return HIPX_SUCCESS;
}

}  // extern "C"

extern "C" {

int hipxCommGetUniqueId(void *id256)
{
  HIPX_CHECK_INIT();
  static_assert(2 * NCCL_UNIQUE_ID_BYTES == HIPX_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id[2];
  HIPX_NCCL(ncclGetUniqueId(&id[0]));
  HIPX_NCCL(ncclGetUniqueId(&id[1]));
  memcpy(id256, id, HIPX_COMM_ID_BYTES);
  return HIPX_SUCCESS;
}

int hipxCommInit(const void *id256, int rank, int nranks)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  if (c.active) return HIPX_SUCCESS;
  HIPX_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
  ncclUniqueId id[2];
  memcpy(id, id256, HIPX_COMM_ID_BYTES);
  HIPX_NCCL(ncclCommInitRank(&c.comm, nranks, id[0], rank));
  HIPX_NCCL(ncclCommInitRank(&c.rcomm, nranks, id[1], rank));
  c.rank   = rank;
  c.nranks = nranks;
  HIPX_HIP(hipMalloc((void **)&c.d_red, sizeof(double) * 64));
  HIPX_HIP(hipMalloc((void **)&c.d_gather, sizeof(double) * 64 * (size_t)nranks));
  HIPX_HIP(hipHostMalloc((void **)&c.h_red, sizeof(double) * 64, hipHostMallocDefault));
  c.active = true;
  return HIPX_SUCCESS;
}

int hipxCommFinalize(void)
{
  Comm &c = cm();
  if (!c.active) return HIPX_SUCCESS;
  HIPX_HIP(hipDeviceSynchronize());
  if (c.ipc) {
    for (void *q : c.opened) (void)hipIpcCloseMemHandle(q);
    (void)hipFree(c.arena);
    (void)hipFree(c.d_peer);
    if (c.h_err) (void)hipHostFree(c.h_err);
  } else {
    HIPX_NCCL(ncclCommDestroy(c.comm));
    HIPX_NCCL(ncclCommDestroy(c.rcomm));
  }
  (void)hipFree(c.d_red);
  (void)hipFree(c.d_gather);
  (void)hipHostFree(c.h_red);
  if (c.d_red2) (void)hipFree(c.d_red2);
  if (c.sp_stream) {
    (void)hipStreamDestroy(c.sp_stream);
    (void)hipEventDestroy(c.sp_ev_begin);
    (void)hipEventDestroy(c.sp_ev_done);
  }
  c = Comm();
  return HIPX_SUCCESS;
}

int hipxCommRank(int *rank, int *nranks)
{
  Comm &c = cm();
  if (rank) *rank = c.active ? c.rank : 0;
  if (nranks) *nranks = c.active ? c.nranks : 1;
  return HIPX_SUCCESS;
}

int hipxCommCheckError(void) { return comm_err_check(); }

int hipxCommAllreduceSum(double *vals, int n)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  if (!c.active || c.nranks == 1 || n <= 0) return HIPX_SUCCESS;
  HIPX_ARG(n <= 64, "at most 64 scalars per all-reduce");
  hipStream_t s = rt().compute;
  memcpy(c.h_red, vals, sizeof(double) * (size_t)n);
  HIPX_HIP(hipMemcpyAsync(c.d_red, c.h_red, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
  {
    int ierr = allreduce_dev(c.d_red, n);
    if (ierr) return ierr;
  }
  HIPX_HIP(hipMemcpyAsync(c.h_red, c.d_red, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  HIPX_HIP(hipStreamSynchronize(s));
  memcpy(vals, c.h_red, sizeof(double) * (size_t)n);
  return comm_err_check();
}

int hipxVecMDotAllreduce(const double *x, hipx_int nv, const double *const *y, hipx_int n, double *results)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(nv >= 1 && nv <= 16, "1 <= nv <= 16");  // (16 sums = 32 pair words in exact mode: the staging line and the IPC slots hold 64)
  static const bool force = getenv("HIPX_FORCE_ALLREDUCE") != nullptr;  // test hook: run the chain on a 1-rank communicator too
  if (!c.active || (c.nranks == 1 && !force)) return hipxVecMDot(x, nv, y, n, results);
  const int slot = HIPX_MAX_RED_SLOTS - 2;  // reserved for this chain
  if (n > 0) {
    int ierr = launch_mdot_nosignal(x, (int)nv, y, n, slot, c.d_red);
    if (ierr) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 2 * (size_t)nv, rt().compute));  // rank without rows (sums or pairs)
  {
    int ierr = allreduce_dev(c.d_red, (int)nv, red_pairs());
    if (ierr) return ierr;
  }
  int ierr = red_signal(slot, c.d_red, (int)nv);
  if (ierr) return ierr;
  if ((ierr = red_wait(slot, (int)nv, results))) return ierr;
  return comm_err_check();
}

int hipxCGFusedUpdateAllreduce(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, double *sums2)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  static const bool force = getenv("HIPX_FORCE_ALLREDUCE") != nullptr;
  if (!c.active || (c.nranks == 1 && !force)) return hipxCGFusedUpdate(x, r, z, p, w, d, a, n, sums2);
  const int slot = HIPX_MAX_RED_SLOTS - 2;
  if (n > 0) {
    int ierr = launch_cg_fused_nosignal(x, r, z, p, w, d, a, n, slot, c.d_red);
    if (ierr) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 4, rt().compute));
  {
    int ierr = allreduce_dev(c.d_red, 2, red_pairs());
    if (ierr) return ierr;
  }
  int ierr = red_signal(slot, c.d_red, 2);
  if (ierr) return ierr;
  if ((ierr = red_wait(slot, 2, sums2))) return ierr;
  return comm_err_check();
}

// ---- launch-ahead CG on several ranks: the reductions complete ON THE STREAM (local kernel -> all-reduce -> publish to the host
// slot and to device memory), so the kernels of the next iteration, which read their scalars from device memory, can be queued
// before the host has seen anything.  Same arithmetic as hipxVecMDotAllreduce / hipxCGFusedUpdateAllreduce.
int hipxMatMultMPIDotBegin(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y, hipx_int n, int slot, double *dev_dot)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && dev_dot, "communicator not initialised / bad slot");
  int ierr = matmult_mpi(Ad, Bo, h, x, lvec, y, false);  // mpiaij.c:1047-1061 (the ghost buffer's acknowledgement rides in the all-reduce kernel below)
  if (ierr) return ierr;
  const double *ys[1] = {y};
  if (n > 0) {
    if ((ierr = launch_mdot_nosignal(x, 1, ys, n, slot, c.d_red))) return ierr;  // cg.c:258 VecXDot(P, W), local part
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 2, rt().compute));
  return allreduce_signal(c.d_red, 1, red_pairs(), slot, dev_dot, h);
}

// hipxVecMDotBegin with the sums all-reduced on the stream (the single-reduction CG's one 24-byte all-reduce per iteration, launch-ahead form)
int hipxVecMDotBeginAllreduce(const double *x, hipx_int nv, const double *const *y, hipx_int n, int slot, double *dev_results)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && nv >= 1 && nv <= 16 && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && dev_results, "communicator not initialised / bad slot / nv");
  int ierr;
  if (n > 0) {
    if ((ierr = launch_mdot_nosignal(x, (int)nv, y, n, slot, c.d_red))) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 2 * (size_t)nv, rt().compute));
  return allreduce_signal(c.d_red, (int)nv, red_pairs(), slot, dev_results, nullptr);
}

int hipxCGFusedUpdateBeginAllreduce(double *x, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi,
                                    hipx_int n, int slot, double *dev_sums2)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && dev_beta && dev_dpi && dev_sums2, "communicator not initialised / bad slot / null scalars");
  HIPX_ARG(z || !d, "z may only be omitted with a constant diagonal (d == NULL)");
  int ierr;
  if (n > 0) {
    if ((ierr = launch_cg_fused_dev_nosignal(x, r, z, p, w, d, dconst, dev_beta, dev_dpi, n, slot, c.d_red))) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 4, rt().compute));
  return allreduce_signal(c.d_red, 2, red_pairs(), slot, dev_sums2, nullptr);
}

// ---- split-phase all-reduce (round 6).  Begin: the local sums (or pairs) in c.d_red2 start travelling; End: the compute stream waits for the
// peers, folds in rank order, publishes to the host slot and to device memory.  Same arithmetic as allreduce_dev: same bits.
static int split_ensure()
{
  Comm &c = cm();
  if (!c.d_red2) HIPX_HIP(hipMalloc((void **)&c.d_red2, sizeof(double) * 64));
  if (!c.ipc && !c.sp_stream) {
    HIPX_HIP(hipStreamCreateWithFlags(&c.sp_stream, hipStreamNonBlocking));
    HIPX_HIP(hipEventCreateWithFlags(&c.sp_ev_begin, hipEventDisableTiming));
    HIPX_HIP(hipEventCreateWithFlags(&c.sp_ev_done, hipEventDisableTiming));
  }
  return HIPX_SUCCESS;
}
static int allreduce_begin(int n, bool pairs)
{
  Comm &c = cm();
  HIPX_ARG(c.sp_n == 0, "a split-phase all-reduce is already in flight (hipxAllreduceEnd first)");
  HIPX_ARG(n >= 1 && (pairs ? 2 * n : n) <= 64, "at most 64 words per all-reduce");
  int ierr;
  if ((ierr = prof_section(HIPX_PROF_ALLREDUCE, true, rt().compute))) return ierr;
  if (c.ipc) {
    ipc_allreduce_kernel<<<1, 64, 0, rt().compute>>>(c.d_red2, n, pairs ? 1 : 0, c.rank, c.nranks, c.d_peer, c.hdr, ++c.seq, c.d_err, ipc_wait_ticks(), PostAck{}, PostSignal{}, 1);
    HIPX_LAUNCH_CHECK();
  } else {
    HIPX_HIP(hipEventRecord(c.sp_ev_begin, rt().compute));
    HIPX_HIP(hipStreamWaitEvent(c.sp_stream, c.sp_ev_begin, 0));
    if (pairs) {
      HIPX_NCCL(ncclAllGather(c.d_red2, c.d_gather, (size_t)(2 * n), ncclDouble, c.rcomm, c.sp_stream));
      dd_fold_ranks_kernel<<<1, 64, 0, c.sp_stream>>>(c.d_gather, n, c.nranks, c.d_red2);
      HIPX_LAUNCH_CHECK();
    } else HIPX_NCCL(ncclAllReduce(c.d_red2, c.d_red2, (size_t)n, ncclDouble, ncclSum, c.rcomm, c.sp_stream));
    HIPX_HIP(hipEventRecord(c.sp_ev_done, c.sp_stream));
  }
  c.sp_n     = n;
  c.sp_pairs = pairs;
  return prof_section(HIPX_PROF_ALLREDUCE, false, rt().compute);
}

int hipxAllreduceEnd(int slot, int nvals, double *dev_out)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && c.sp_n > 0 && nvals == c.sp_n && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "no split-phase all-reduce of that size in flight / bad slot");
  Runtime &r = rt();
  int      ierr;
  c.sp_n = 0;
  if ((ierr = prof_section(HIPX_PROF_ALLREDUCE, true, r.compute))) return ierr;
  if (c.ipc) {
    PostSignal sig;
    sig.flag    = r.d_flags + slot;
    sig.seq     = ++r.seq[slot];
    sig.results = slot_results_dev(slot);
    sig.dres    = dev_out;
    ipc_allreduce_kernel<<<1, 64, 0, r.compute>>>(c.d_red2, nvals, c.sp_pairs ? 1 : 0, c.rank, c.nranks, c.d_peer, c.hdr, c.seq, c.d_err, ipc_wait_ticks(), PostAck{}, sig, 2);
    HIPX_LAUNCH_CHECK();
  } else {
    HIPX_HIP(hipStreamWaitEvent(r.compute, c.sp_ev_done, 0));
    if ((ierr = red_signal(slot, c.d_red2, nvals, dev_out))) return ierr;
  }
  return prof_section(HIPX_PROF_ALLREDUCE, false, r.compute);
}

int hipxPipeCGUpdateBeginAllreduce(const hipxPipeCGVecs *v, const double *d, double dconst, int normkind, int first, const double *dev_sums, const double *dev_sums_old,
                                   const double *dev_alpha_old, double *dev_alpha_out, hipx_int n, int slot)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && v && dev_sums && dev_alpha_out && (first || (dev_sums_old && dev_alpha_old)) && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "communicator not initialised / null argument / bad slot");
  int ierr;
  if ((ierr = split_ensure())) return ierr;
  if (n > 0) {
    RedOut o  = red_out(slot, false, nullptr);
    o.results = c.d_red2;
    o.pairs   = rt().red_exact;
    if ((ierr = launch_pipecg_update(v, d, dconst, normkind, first, dev_sums, dev_sums_old, dev_alpha_old, dev_alpha_out, n, o))) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red2, 0, sizeof(double) * 6, rt().compute));  // (a rank without rows: zero sums or pairs; alpha is not needed by anybody there)
  return allreduce_begin(3, red_pairs());
}

int hipxGroppCGDirectionBeginAllreduce(double *p, double *s, double *x, const double *z, const double *Z, const double *dev_gamma_new, const double *dev_gamma_old, const double *dev_alpha_old,
                                       hipx_int n, int slot, double *dev_t_out)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && dev_gamma_new && dev_gamma_old && dev_alpha_old && dev_t_out && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "communicator not initialised / null scalars / bad slot");
  int ierr;
  if (n > 0) {
    RedOut o  = red_out(slot, false, nullptr);
    o.results = c.d_red;
    o.pairs   = rt().red_exact;
    if ((ierr = launch_gropp_dir(p, s, x, z, Z, dev_gamma_new, dev_gamma_old, dev_alpha_old, n, o))) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 2, rt().compute));
  return allreduce_signal(c.d_red, 1, red_pairs(), slot, dev_t_out, nullptr);
}

int hipxGroppCGUpdateBeginAllreduce(double *r, double *z, const double *s, const double *d, double dconst, int normkind, const double *dev_gamma, const double *dev_t, double *dev_alpha_out,
                                    hipx_int n, int slot)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && dev_gamma && dev_t && dev_alpha_out && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "communicator not initialised / null scalars / bad slot");
  int ierr;
  if ((ierr = split_ensure())) return ierr;
  if (n > 0) {
    RedOut o  = red_out(slot, false, nullptr);
    o.results = c.d_red2;
    o.pairs   = rt().red_exact;
    if ((ierr = launch_gropp_update(r, z, s, d, dconst, normkind, dev_gamma, dev_t, dev_alpha_out, n, o))) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red2, 0, sizeof(double) * 4, rt().compute));
  return allreduce_begin(2, red_pairs());
}

int hipxVecMDotAllreduceBegin(const double *x, hipx_int nv, const double *const *y, hipx_int n, int slot)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(c.active && nv >= 1 && nv <= 16 && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "communicator not initialised / bad slot / nv");
  int ierr;
  if ((ierr = split_ensure())) return ierr;
  if (n > 0) {
    if ((ierr = launch_mdot_nosignal(x, (int)nv, y, n, slot, c.d_red2))) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red2, 0, sizeof(double) * 2 * (size_t)nv, rt().compute));
  return allreduce_begin((int)nv, red_pairs());
}

int hipxHaloCreate(int nsend, const int *send_ranks, const hipx_int *send_off, const hipx_int *send_idx, int nrecv, const int *recv_ranks, const hipx_int *recv_off,
                   hipxHalo *out)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(nsend >= 0 && nrecv >= 0 && out, "bad halo sizes");
  hipxHalo h = new hipxHalo_s;
  h->nsend   = nsend;
  h->nrecv   = nrecv;
  h->send_ranks.assign(send_ranks, send_ranks + nsend);
  h->recv_ranks.assign(recv_ranks, recv_ranks + nrecv);
  h->send_off.assign(send_off, send_off + nsend + 1);
  h->recv_off.assign(recv_off, recv_off + nrecv + 1);
  const hipx_int ns = h->send_off[nsend];
  HIPX_HIP(hipMalloc((void **)&h->d_send_idx, sizeof(hipx_int) * (size_t)(ns ? ns : 1)));
  HIPX_HIP(hipMalloc((void **)&h->d_sendbuf, sizeof(double) * (size_t)(ns ? ns : 1)));
  if (ns) HIPX_HIP(hipMemcpy(h->d_send_idx, send_idx, sizeof(hipx_int) * (size_t)ns, hipMemcpyHostToDevice));
  HIPX_HIP(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
  HIPX_HIP(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
  *out = h;
  return HIPX_SUCCESS;
}

int hipxHaloDestroy(hipxHalo *ph)
{
  if (!ph || !*ph) return HIPX_SUCCESS;
  hipxHalo h = *ph;
  HIPX_HIP(hipDeviceSynchronize());
  (void)hipFree(h->d_send_idx);
  (void)hipFree(h->d_sendbuf);
  for (void *q : h->opened) (void)hipIpcCloseMemHandle(q);
  (void)hipFree(h->arena);
  (void)hipFree(h->d_ticket);
  (void)hipFree(h->d_recv_ranks);
  (void)hipFree(h->d_ack_ptrs);
  if (h->h_err) (void)hipHostFree(h->h_err);  // (d_err is its device alias)
  (void)hipEventDestroy(h->ev_packed);
  (void)hipEventDestroy(h->ev_done);
  delete h;
  *ph = nullptr;
  return HIPX_SUCCESS;
}

static int halo_begin(hipxHalo h, const double *x, double *lvec, const PackCG *cgp);
int hipxHaloBegin(hipxHalo h, const double *x, double *lvec) { return halo_begin(h, x, lvec, nullptr); }

// cgp != NULL: what is sent is the new CG direction formed from x (= the old one) on the way out, see PackCG
static int halo_begin(hipxHalo h, const double *x, double *lvec, const PackCG *cgp)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(h, "null halo");
  if (h->nsend + h->nrecv == 0) return HIPX_SUCCESS;
  const PackCG cg = cgp ? *cgp : PackCG{};
  if (h->ipc) {
    if (*reinterpret_cast<volatile unsigned int *>(h->h_err))
      return fail(HIPX_ERR_GPU, "ghost exchange (IPC): a neighbour never published / acknowledged its data (wait limit reached)", __FILE__, __LINE__);
    const unsigned long long s = ++h->seq;
    const int                q = (int)(s & 1);
    h->ghost_cur = reinterpret_cast<double *>(h->arena + h->hdr_bytes) + (size_t)q * h->nghost;
    {
      int ierr = prof_section(HIPX_PROF_HALO, true, rt().compute);
      if (ierr) return ierr;
    }
    // round 6: the put kernel runs on the COMPUTE stream, in front of the product (see ipc_put_kernel); up to four neighbours per launch
    for (int r0 = 0; r0 < h->nsend; r0 += 4) {
      PutArgs  pa;
      hipx_int maxn2 = 0;
      int      ns    = 0;
      for (int r = r0; r < h->nsend && ns < 4; r++) {
        const hipx_int cnt = h->send_off[r + 1] - h->send_off[r];
        if (!cnt) continue;
        char *pb = h->send_base[(size_t)r];
        pa.seg[ns].idx      = h->d_send_idx + h->send_off[r];
        pa.seg[ns].n        = cnt;
        pa.seg[ns].dst      = reinterpret_cast<double *>(pb + h->hdr_bytes) + (size_t)q * (size_t)h->send_peer_nghost[(size_t)r] + h->send_peer_off[(size_t)r];
        pa.seg[ns].flag     = reinterpret_cast<unsigned long long *>(pb) + h->me;                                      // peer.data_seq[me]
        pa.seg[ns].ack      = reinterpret_cast<const unsigned long long *>(h->arena) + h->nranks + h->send_ranks[r];  // my ack_seq[peer]
        pa.seg[ns].need_ack = s >= 2 ? s - 2 : 0;
        pa.seg[ns].ticket   = h->d_ticket + r;
        maxn2               = std::max<hipx_int>(maxn2, std::max<hipx_int>(cnt >> 1, 1));
        ns++;
      }
      if (!ns) continue;
      hipx_int g = (maxn2 + 255) / 256;
      if (g > 128) g = 128;  // (every workgroup arrives on its segment's ticket word: ~12 ns per arrival -- 1024 workgroups spent 12 us there)
      const dim3 grid((unsigned)g, (unsigned)ns);
      if (cgp) ipc_put_kernel<true><<<grid, 256, 0, rt().compute>>>(x, pa, s, h->d_err, ipc_wait_ticks(), cg);
      else ipc_put_kernel<false><<<grid, 256, 0, rt().compute>>>(x, pa, s, h->d_err, ipc_wait_ticks(), cg);
    }
    HIPX_LAUNCH_CHECK();
    return prof_section(HIPX_PROF_HALO, false, rt().compute);
  }
  if (!c.active || c.ipc) return fail(HIPX_ERR_ORDER, "hipxCommInit() (RCCL) or hipxHaloIpcAttach() must precede a ghost exchange", __FILE__, __LINE__);
  const hipx_int ns = h->send_off[h->nsend];
  if (ns) {
    hipx_int g = (ns + 255) / 256;
    if (g > 2048) g = 2048;
    if (cgp) pack_kernel<true><<<(unsigned)g, 256, 0, rt().compute>>>(x, h->d_send_idx, h->d_sendbuf, ns, cg);
    else pack_kernel<false><<<(unsigned)g, 256, 0, rt().compute>>>(x, h->d_send_idx, h->d_sendbuf, ns, cg);
    HIPX_LAUNCH_CHECK();
  }
  // the comm stream may start once the pack kernel (and whatever produced x) has finished
  HIPX_HIP(hipEventRecord(h->ev_packed, rt().compute));
  HIPX_HIP(hipStreamWaitEvent(rt().comm, h->ev_packed, 0));
  {
    int ierr = prof_section(HIPX_PROF_HALO, true, rt().comm);
    if (ierr) return ierr;
  }
  HIPX_NCCL(ncclGroupStart());
  for (int r = 0; r < h->nrecv; r++) {
    const hipx_int cnt = h->recv_off[r + 1] - h->recv_off[r];
    if (cnt) HIPX_NCCL(ncclRecv(lvec + h->recv_off[r], (size_t)cnt, ncclDouble, h->recv_ranks[r], c.comm, rt().comm));
  }
  for (int r = 0; r < h->nsend; r++) {
    const hipx_int cnt = h->send_off[r + 1] - h->send_off[r];
    if (cnt) HIPX_NCCL(ncclSend(h->d_sendbuf + h->send_off[r], (size_t)cnt, ncclDouble, h->send_ranks[r], c.comm, rt().comm));
  }
  HIPX_NCCL(ncclGroupEnd());
  {
    int ierr = prof_section(HIPX_PROF_HALO, false, rt().comm);
    if (ierr) return ierr;
  }
  HIPX_HIP(hipEventRecord(h->ev_done, rt().comm));
  return HIPX_SUCCESS;
}

int hipxHaloEnd(hipxHalo h)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(h, "null halo");
  if (h->nsend + h->nrecv == 0) return HIPX_SUCCESS;
  if (!h->ipc) HIPX_HIP(hipStreamWaitEvent(rt().compute, h->ev_done, 0));  // RCCL: the receives landed (IPC: the put kernel ran on this stream)
  if (h->ipc && h->nrecv) {
    ipc_wait_kernel<<<1, 64, 0, rt().compute>>>(reinterpret_cast<const unsigned long long *>(h->arena), h->d_recv_ranks, h->nrecv, h->seq, h->d_err, ipc_wait_ticks());
    HIPX_LAUNCH_CHECK();
  }
  return HIPX_SUCCESS;
}

// hipxHaloEnd for a consumer kernel that waits for the neighbours' sequence flags ITSELF (offdiag_dot_kernel): no wait kernel (one launch, ~5 us, less on the
// critical path); *w says what to wait for -- nothing with RCCL (the event has ordered the receives) or with more than four neighbours (the wait kernel ran)
static int halo_end_inline(hipxHalo h, IpcWait *w)
{
  *w = IpcWait{};
  if (h->nsend + h->nrecv == 0) return HIPX_SUCCESS;
  if (!h->ipc || h->nrecv > 4) return hipxHaloEnd(h);
  for (int r = 0; r < h->nrecv; r++) w->flag[r] = reinterpret_cast<const unsigned long long *>(h->arena) + h->recv_ranks[(size_t)r];  // my data_seq[sender]
  w->n     = h->nrecv;
  w->want  = h->seq;
  w->err   = h->d_err;
  w->limit = ipc_wait_ticks();
  return HIPX_SUCCESS;
}

// IPC: the ghost values of the exchange have been consumed (after the off-diagonal product): tell the senders, so that the
// exchange after next may overwrite this buffer
static int halo_release(hipxHalo h)
{
  if (!h->ipc || h->nsend + h->nrecv == 0) return HIPX_SUCCESS;
  if (h->nrecv) {
    ipc_ack_kernel<<<1, 64, 0, rt().compute>>>(h->d_ack_ptrs, h->nrecv, h->seq);
    HIPX_LAUNCH_CHECK();
  }
  return HIPX_SUCCESS;  // (a wait that gave up has written the host-mapped error word: seen at the next call)
}

static int allreduce_signal(double *d_vals, int n, bool pairs, int slot, double *dres, hipxHalo h)
{
  Comm &c = cm();
  int   ierr;
  if (c.ipc) {
    Runtime   &r = rt();
    PostAck    ack;
    PostSignal sig;
    if (h && h->ipc && h->nrecv) {
      if (h->nrecv <= 64) {
        ack.ptrs = h->d_ack_ptrs;
        ack.n    = h->nrecv;
        ack.seq  = h->seq;
      } else if ((ierr = halo_release(h))) return ierr;
    }
    sig.flag    = r.d_flags + slot;
    sig.seq     = ++r.seq[slot];
    sig.results = slot_results_dev(slot);
    sig.dres    = dres;
    return allreduce_dev(d_vals, n, pairs, &ack, &sig);
  }
  if (h && (ierr = halo_release(h))) return ierr;
  if ((ierr = allreduce_dev(d_vals, n, pairs))) return ierr;
  return red_signal(slot, d_vals, n, dres);
}

int hipxHaloIpcExport(hipxHalo h, int rank, int nranks, void *blob)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(h && blob && nranks >= 1 && rank >= 0 && rank < nranks, "bad arguments");
  HIPX_ARG(h->nrecv <= IPC_MAXN && h->nsend <= IPC_MAXN, "IPC ghost exchange: at most 60 neighbours");
  h->me        = rank;
  h->nranks    = nranks;
  h->nghost    = (size_t)h->recv_off[h->nrecv];
  h->hdr_bytes = ((size_t)16 * (size_t)nranks + 255) & ~(size_t)255;
  const size_t bytes = h->hdr_bytes + 2 * sizeof(double) * std::max<size_t>(h->nghost, 1) + 16;  // (+ 16: offdiag_dot_kernel reads the ghost values in aligned pairs)
  if (!h->arena) {
    // fine-grained: flags are polled while another process / GPU writes them, and the ghost values are written by the peers
    HIPX_HIP(hipExtMallocWithFlags((void **)&h->arena, bytes, hipDeviceMallocFinegrained));
    HIPX_HIP(hipMemset(h->arena, 0, bytes));
    HIPX_HIP(hipHostMalloc((void **)&h->h_err, sizeof(unsigned int), hipHostMallocMapped));  // (host-mapped: a kernel that gives up writes it where the host reads it -- no copy on the data path)
    *h->h_err = 0;
    HIPX_HIP(hipHostGetDevicePointer((void **)&h->d_err, h->h_err, 0));
  }
  IpcBlob b;
  memset(&b, 0, sizeof(b));
  HIPX_HIP(hipIpcGetMemHandle(&b.handle, h->arena));
  b.rank   = rank;
  b.nrecv  = h->nrecv;
  b.nghost = (long long)h->nghost;
  for (int r = 0; r < h->nrecv; r++) b.recv_ranks[r] = h->recv_ranks[r];
  for (int r = 0; r <= h->nrecv; r++) b.recv_off[r] = h->recv_off[r];
  memset(blob, 0, HIPX_HALO_IPC_BLOB_BYTES);
  memcpy(blob, &b, sizeof(b));
  return HIPX_SUCCESS;
}

int hipxHaloIpcAttach(hipxHalo h, const void *all_blobs)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(h && all_blobs && h->arena, "hipxHaloIpcExport() first");
  const char *blobs = (const char *)all_blobs;
  std::vector<char *> base((size_t)h->nranks, nullptr);
  auto map_rank = [&](int r, char **out) -> int {
    if (!base[(size_t)r]) {
      if (r == h->me) base[(size_t)r] = h->arena;
      else {
        IpcBlob b;
        memcpy(&b, blobs + (size_t)r * HIPX_HALO_IPC_BLOB_BYTES, sizeof(b));
        void *q = nullptr;
        HIPX_HIP(hipIpcOpenMemHandle(&q, b.handle, hipIpcMemLazyEnablePeerAccess));
        h->opened.push_back(q);
        base[(size_t)r] = (char *)q;
      }
    }
    *out = base[(size_t)r];
    return HIPX_SUCCESS;
  };
  h->send_base.assign((size_t)h->nsend, nullptr);
  h->send_peer_off.assign((size_t)h->nsend, 0);
  h->send_peer_nghost.assign((size_t)h->nsend, 0);
  for (int i = 0; i < h->nsend; i++) {
    const int r = h->send_ranks[i];
    HIPX_ARG(r >= 0 && r < h->nranks, "send neighbour out of range");
    int ierr = map_rank(r, &h->send_base[(size_t)i]);
    if (ierr) return ierr;
    IpcBlob b;
    memcpy(&b, blobs + (size_t)r * HIPX_HALO_IPC_BLOB_BYTES, sizeof(b));
    int found = -1;
    for (int k = 0; k < b.nrecv; k++)
      if (b.recv_ranks[k] == h->me) found = k;
    HIPX_ARG(found >= 0 && b.recv_off[found + 1] - b.recv_off[found] == h->send_off[i + 1] - h->send_off[i], "IPC ghost exchange: send and receive lists of two ranks do not match");
    h->send_peer_off[(size_t)i]    = b.recv_off[found];
    h->send_peer_nghost[(size_t)i] = std::max<long long>(b.nghost, 1);
  }
  std::vector<unsigned long long *> ack((size_t)std::max(h->nrecv, 1), nullptr);
  for (int i = 0; i < h->nrecv; i++) {
    const int r = h->recv_ranks[i];
    HIPX_ARG(r >= 0 && r < h->nranks, "receive neighbour out of range");
    char *pb  = nullptr;
    int  ierr = map_rank(r, &pb);
    if (ierr) return ierr;
    ack[(size_t)i] = reinterpret_cast<unsigned long long *>(pb) + h->nranks + h->me;  // peer.ack_seq[me]
  }
  HIPX_HIP(hipMalloc((void **)&h->d_ticket, sizeof(unsigned int) * (size_t)std::max(h->nsend, 1)));
  HIPX_HIP(hipMemset(h->d_ticket, 0, sizeof(unsigned int) * (size_t)std::max(h->nsend, 1)));
  HIPX_HIP(hipMalloc((void **)&h->d_recv_ranks, sizeof(int) * (size_t)std::max(h->nrecv, 1)));
  HIPX_HIP(hipMalloc((void **)&h->d_ack_ptrs, sizeof(void *) * (size_t)std::max(h->nrecv, 1)));
  if (h->nrecv) {
    HIPX_HIP(hipMemcpy(h->d_recv_ranks, h->recv_ranks.data(), sizeof(int) * (size_t)h->nrecv, hipMemcpyHostToDevice));
    HIPX_HIP(hipMemcpy(h->d_ack_ptrs, ack.data(), sizeof(void *) * (size_t)h->nrecv, hipMemcpyHostToDevice));
  }
  h->ghost_cur = reinterpret_cast<double *>(h->arena + h->hdr_bytes);
  h->ipc       = true;
  return HIPX_SUCCESS;
}

static int matmult_mpi(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y, bool release);
int hipxMatMultMPI(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y) { return matmult_mpi(Ad, Bo, h, x, lvec, y, true); }
static int matmult_mpi(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y, bool release)
{
  HIPX_CHECK_INIT();
  int ierr;
  if ((ierr = hipxHaloBegin(h, x, lvec))) return ierr;  // VecScatterBegin   mpiaij.c:1056
  if ((ierr = hipxMatMult(Ad, x, y))) return ierr;      // A->ops->mult      mpiaij.c:1057 (overlaps the exchange)
  if ((ierr = prof_section(HIPX_PROF_OFFDIAG, true, rt().compute))) return ierr;
  if ((ierr = hipxHaloEnd(h))) return ierr;             // VecScatterEnd     mpiaij.c:1058
  const double *ghost = (h && h->ipc) ? h->ghost_cur : lvec;  // IPC transport: the neighbours wrote straight into this rank's ghost buffer
  if (Bo && (ierr = hipxMatMultAdd(Bo, ghost, y, y))) return ierr;  // B->ops->multadd   mpiaij.c:1059
  if ((ierr = prof_section(HIPX_PROF_OFFDIAG, false, rt().compute))) return ierr;
  return (h && release) ? halo_release(h) : HIPX_SUCCESS;
}

int hipxMatMultAddMPI(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, const double *y, double *z)
{
  HIPX_CHECK_INIT();
  int ierr;
  if ((ierr = hipxHaloBegin(h, x, lvec))) return ierr;   // VecScatterBegin        mpiaij.c:1078
  if ((ierr = hipxMatMultAdd(Ad, x, y, z))) return ierr; // A->ops->multadd        mpiaij.c:1079
  if ((ierr = hipxHaloEnd(h))) return ierr;              // VecScatterEnd          mpiaij.c:1080
  const double *ghost = (h && h->ipc) ? h->ghost_cur : lvec;
  if (Bo && (ierr = hipxMatMultAdd(Bo, ghost, z, z))) return ierr;  // B->ops->multadd   mpiaij.c:1081
  return h ? halo_release(h) : HIPX_SUCCESS;
}

// ---- the two-kernel CG iteration on a rank WITH an off-diagonal block (round 6) -----------------------------------------------------------------
// One rank runs [direction update + product + dot] as ONE kernel (hipxMatMultCGDirectionDotBegin).  Here the same for MatMult_MPIAIJ (mpiaij.c:1047-1061):
//   comm stream:     the new direction of the rows the neighbours need, formed on the way out (PackCG) -> put / send          VecScatterBegin  mpiaij.c:1056
//   compute stream:  p_new = z d + b p, x += a p, w = Ad p_new, dot partials of the rows WITHOUT off-diagonal entries           A->ops->mult     mpiaij.c:1057
//                    wait for the ghost values                                                                                   VecScatterEnd    mpiaij.c:1058
//                    w += Bo ghost on the boundary rows + THEIR p_i w_i + the fold of all partials (offdiag_dot_kernel)          B->ops->multadd  mpiaij.c:1059, cg.c:258
//                    all-reduce of the one sum -> host slot + device copy
// i.e. the separate direction kernel (5 vector passes) and the separate dot kernel (2 passes) of hipxCGAypxAxpyDev + hipxMatMultMPIDotBegin are gone.  The
// vectors are those kernels' bit for bit (same operations per element); the dot is the same set of products p_i w_i summed in another order (default
// reductions: to rounding; exact reductions: Dot2 over the complete vectors as before, order-free).  *fused = 0: nothing was enqueued (the pair of blocks is
// not a z-slab of a stencil grid the march kernel takes, hipxMatMPICGPlan_) -- the caller runs the separate kernels.
extern "C" int hipxMatMPICGPlan_(hipxMat A, hipxMat B, int *ok, int *skipmask);
extern "C" int hipxMatMultCGDirectionPartial_(hipxMat A, int skipmask, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a,
                                              const double *dev_beta_new, const double *dev_beta_old, const double *dev_dpi, double *w, int want_dot, int *fused, const double **dotpart,
                                              hipx_int *npart);
extern "C" int hipxMatMultAddDotFold_(hipxMat A, hipxMat B, const double *ghost, double *w, const double *p, const double *partA, hipx_int npartA, int slot, double *dst, const hipx::IpcWait *wt);

int hipxMatMultMPICGDirectionDotBegin(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a,
                                      const double *dev_beta_new, const double *dev_beta_old, const double *dev_dpi, double *lvec, double *w, hipx_int n, int slot, double *dev_dot, int *fused)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(Ad && fused && p_old && p_new && z && x && w, "null argument");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && dev_dot, "reduction slot out of range / no device copy of the dot");
  HIPX_ARG(p_old != p_new && p_new != w && p_old != w && x != w && x != p_new && z != w && z != p_new, "the vectors must be distinct");
  *fused = 0;
  static const bool off = getenv("HIPX_NO_CGFUSE") != nullptr || getenv("HIPX_NO_MPI_CGFUSE") != nullptr;
  if (off || !c.active || !Bo || !h || h->nsend + h->nrecv == 0 || n <= 0) return HIPX_SUCCESS;
  if ((reinterpret_cast<uintptr_t>(p_old) | reinterpret_cast<uintptr_t>(p_new) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return HIPX_SUCCESS;
  int ok = 0, skip = 0, ierr;
  if ((ierr = hipxMatMPICGPlan_(Ad, Bo, &ok, &skip))) return ierr;
  if (!ok) return HIPX_SUCCESS;
  PackCG cg;
  cg.z            = z;
  cg.dconst       = dconst;
  cg.b            = b;
  cg.dev_beta_new = dev_beta_new;
  cg.dev_beta_old = dev_beta_new ? dev_beta_old : nullptr;
  if ((ierr = halo_begin(h, p_old, lvec, &cg))) return ierr;
  const bool    exact = red_pairs();
  const double *dotpart = nullptr;
  hipx_int      npart   = 0;
  int           done    = 0;
  if ((ierr = hipxMatMultCGDirectionPartial_(Ad, skip, p_old, p_new, z, dconst, x, b, a, dev_beta_new, dev_beta_old, dev_dpi, w, exact ? 0 : 1, &done, &dotpart, &npart))) return ierr;
  if (!done) return fail(HIPX_ERR_ORDER, "MPIAIJ CG product: the plan said the diagonal block takes the fused kernel, the launch declined (ghost exchange already started)", __FILE__, __LINE__);
  if ((ierr = prof_section(HIPX_PROF_OFFDIAG, true, rt().compute))) return ierr;
  const double *ghost = h->ipc ? h->ghost_cur : lvec;
  if (exact) {
    if ((ierr = hipxHaloEnd(h))) return ierr;
    if ((ierr = hipxMatMultAdd(Bo, ghost, w, w))) return ierr;
    const double *ys[1] = {w};
    if ((ierr = launch_mdot_nosignal(p_new, 1, ys, n, slot, c.d_red))) return ierr;
  } else {
    IpcWait wt;
    if ((ierr = halo_end_inline(h, &wt))) return ierr;  // (IPC: the off-diagonal kernel waits for the neighbours' flags itself)
    if ((ierr = hipxMatMultAddDotFold_(Ad, Bo, ghost, w, p_new, dotpart, npart, slot, c.d_red, &wt))) return ierr;
  }
  if ((ierr = prof_section(HIPX_PROF_OFFDIAG, false, rt().compute))) return ierr;
  if ((ierr = allreduce_signal(c.d_red, 1, exact, slot, dev_dot, h))) return ierr;  // (IPC: the acknowledgement of the ghost buffer rides in the all-reduce kernel)
  *fused = 1;
  return HIPX_SUCCESS;
}

int hipxHaloGhost(hipxHalo h, const double *lvec, const double **ghost)
{
  HIPX_ARG(h && ghost, "null argument");
  *ghost = h->ipc ? h->ghost_cur : lvec;
  return HIPX_SUCCESS;
}

int hipxHaloRelease(hipxHalo h)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(h, "null halo");
  return halo_release(h);
}

int hipxHaloTransport(hipxHalo h, int *transport)
{
  HIPX_ARG(h && transport, "null argument");
  *transport = h->ipc ? 1 : ((cm().active && !cm().ipc) ? 2 : 0);
  return HIPX_SUCCESS;
}

}  // extern "C"
