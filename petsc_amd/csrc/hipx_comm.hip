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
  double    *h_red = nullptr;  // pinned
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

__global__ void pack_kernel(const double *__restrict__ x, const hipx_int *__restrict__ idx, double *__restrict__ buf, hipx_int n)
{
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (hipx_int)gridDim.x * blockDim.x) buf[i] = x[idx[i]];
}

}  // namespace

struct hipxHalo_s {
  int                   nsend = 0, nrecv = 0;
  std::vector<int>      send_ranks, recv_ranks;
  std::vector<hipx_int> send_off, recv_off;
  hipx_int             *d_send_idx = nullptr;
  double               *d_sendbuf  = nullptr;
  hipEvent_t            ev_packed = nullptr, ev_done = nullptr;
};

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
  HIPX_HIP(hipHostMalloc((void **)&c.h_red, sizeof(double) * 64, hipHostMallocDefault));
  c.active = true;
  return HIPX_SUCCESS;
}

int hipxCommFinalize(void)
{
  Comm &c = cm();
  if (!c.active) return HIPX_SUCCESS;
  HIPX_HIP(hipDeviceSynchronize());
  HIPX_NCCL(ncclCommDestroy(c.comm));
  HIPX_NCCL(ncclCommDestroy(c.rcomm));
  (void)hipFree(c.d_red);
  (void)hipHostFree(c.h_red);
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

int hipxCommAllreduceSum(double *vals, int n)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  if (!c.active || c.nranks == 1 || n <= 0) return HIPX_SUCCESS;
  HIPX_ARG(n <= 64, "at most 64 scalars per all-reduce");
  hipStream_t s = rt().compute;
  memcpy(c.h_red, vals, sizeof(double) * (size_t)n);
  HIPX_HIP(hipMemcpyAsync(c.d_red, c.h_red, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
  HIPX_NCCL(ncclAllReduce(c.d_red, c.d_red, (size_t)n, ncclDouble, ncclSum, c.rcomm, s));
  HIPX_HIP(hipMemcpyAsync(c.h_red, c.d_red, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  HIPX_HIP(hipStreamSynchronize(s));
  memcpy(vals, c.h_red, sizeof(double) * (size_t)n);
  return HIPX_SUCCESS;
}

int hipxVecMDotAllreduce(const double *x, hipx_int nv, const double *const *y, hipx_int n, double *results)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(nv >= 1 && nv <= 8, "1 <= nv <= 8");
  static const bool force = getenv("HIPX_FORCE_ALLREDUCE") != nullptr;  // test hook: run the chain on a 1-rank communicator too
  if (!c.active || (c.nranks == 1 && !force)) return hipxVecMDot(x, nv, y, n, results);
  const int slot = HIPX_MAX_RED_SLOTS - 2;  // reserved for this chain
  if (n > 0) {
    int ierr = launch_mdot_nosignal(x, (int)nv, y, n, slot, c.d_red);
    if (ierr) return ierr;
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * (size_t)nv, rt().compute));  // rank without rows
  HIPX_NCCL(ncclAllReduce(c.d_red, c.d_red, (size_t)nv, ncclDouble, ncclSum, c.rcomm, rt().compute));
  int ierr = red_signal(slot, c.d_red, (int)nv);
  if (ierr) return ierr;
  return red_wait(slot, (int)nv, results);
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
  } else HIPX_HIP(hipMemsetAsync(c.d_red, 0, sizeof(double) * 2, rt().compute));
  HIPX_NCCL(ncclAllReduce(c.d_red, c.d_red, 2, ncclDouble, ncclSum, c.rcomm, rt().compute));
  int ierr = red_signal(slot, c.d_red, 2);
  if (ierr) return ierr;
  return red_wait(slot, 2, sums2);
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
  (void)hipEventDestroy(h->ev_packed);
  (void)hipEventDestroy(h->ev_done);
  delete h;
  *ph = nullptr;
  return HIPX_SUCCESS;
}

int hipxHaloBegin(hipxHalo h, const double *x, double *lvec)
{
  HIPX_CHECK_INIT();
  Comm &c = cm();
  HIPX_ARG(h, "null halo");
  if (h->nsend + h->nrecv == 0) return HIPX_SUCCESS;
  if (!c.active) return fail(HIPX_ERR_ORDER, "hipxCommInit() must precede a ghost exchange", __FILE__, __LINE__);
  const hipx_int ns = h->send_off[h->nsend];
  if (ns) {
    hipx_int g = (ns + 255) / 256;
    if (g > 2048) g = 2048;
    pack_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(x, h->d_send_idx, h->d_sendbuf, ns);
    HIPX_LAUNCH_CHECK();
  }
  // the comm stream may start once the pack kernel (and whatever produced x) has finished
  HIPX_HIP(hipEventRecord(h->ev_packed, rt().compute));
  HIPX_HIP(hipStreamWaitEvent(rt().comm, h->ev_packed, 0));
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
  HIPX_HIP(hipEventRecord(h->ev_done, rt().comm));
  return HIPX_SUCCESS;
}

int hipxHaloEnd(hipxHalo h)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(h, "null halo");
  if (h->nsend + h->nrecv == 0) return HIPX_SUCCESS;
  HIPX_HIP(hipStreamWaitEvent(rt().compute, h->ev_done, 0));
  return HIPX_SUCCESS;
}

int hipxMatMultMPI(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y)
{
  HIPX_CHECK_INIT();
  int ierr;
  if ((ierr = hipxHaloBegin(h, x, lvec))) return ierr;  // VecScatterBegin   mpiaij.c:1056
  if ((ierr = hipxMatMult(Ad, x, y))) return ierr;      // A->ops->mult      mpiaij.c:1057 (overlaps the exchange)
  if ((ierr = hipxHaloEnd(h))) return ierr;             // VecScatterEnd     mpiaij.c:1058
  if (Bo) return hipxMatMultAdd(Bo, lvec, y, y);        // B->ops->multadd   mpiaij.c:1059
  return HIPX_SUCCESS;
}

}  // extern "C"
