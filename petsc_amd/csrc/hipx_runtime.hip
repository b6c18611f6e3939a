// hipx_runtime.hip -- device, streams, memory and events behind the C ABI (include/hipx.h).
#include "hipx_internal.h"
#include <cstring>
#include <cstdlib>
#include <vector>

namespace hipx {
Runtime &rt()
{
  static Runtime r;
  return r;
}
int fail(int code, const char *what, const char *file, int line)
{
  snprintf(rt().errmsg, sizeof(rt().errmsg), "hipx error %d: %s (%s:%d)", code, what, file, line);
  return code;
}
}  // namespace hipx
using namespace hipx;

// Completion of a blocking reduction: the last workgroup writes the sums into pinned host memory, fences at system
// scope and then stores the launch's sequence number; the host spins on that word instead of paying a
// hipStreamSynchronize round trip (3 such waits per CG iteration).  hipStreamQuery every few thousand spins surfaces
// a faulted kernel instead of hanging.
int hipx::red_wait(int slot, int nvals, double *out)
{
  Runtime                     &r    = rt();
  volatile unsigned long long *flag = r.h_flags + slot;
  const unsigned long long     want = r.seq[slot];
  unsigned long                spins = 0;
  while (*flag != want) {
    if ((++spins & 0x3fff) == 0) {
      hipError_t e = hipStreamQuery(r.compute);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(HIPX_ERR_HIP_BASE + (int)e, hipGetErrorString(e), __FILE__, __LINE__);
      if (e == hipSuccess && *flag != want) {  // stream drained but the flag never came: treat as a device fault
        HIPX_HIP(hipStreamSynchronize(r.compute));
        if (*flag != want) return fail(HIPX_ERR_GPU, "reduction kernel finished without signalling", __FILE__, __LINE__);
      }
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const volatile double *h = slot_results_host(slot);
  for (int v = 0; v < nvals; v++) out[v] = h[v];
  return HIPX_SUCCESS;
}

namespace {
struct SectionProf {
  bool                    on = false;
  std::vector<hipEvent_t> ev[HIPX_PROF_NSECTIONS];  // pairs
  size_t                  used[HIPX_PROF_NSECTIONS] = {0};
};
SectionProf &sprof()
{
  static SectionProf p;
  return p;
}
}  // namespace

int hipx::prof_section(int id, bool start, hipStream_t s)
{
  SectionProf &p = sprof();
  if (!p.on || id < 0 || id >= HIPX_PROF_NSECTIONS) return HIPX_SUCCESS;
  if (start && p.used[id] + 2 > p.ev[id].size()) {
    if (p.ev[id].size() >= 200000) return HIPX_SUCCESS;  // bounded: the tally stops growing, the run goes on
    for (int k = 0; k < 2; k++) {
      hipEvent_t e;
      HIPX_HIP(hipEventCreate(&e));
      p.ev[id].push_back(e);
    }
  }
  if (!start && (p.used[id] & 1) == 0) return HIPX_SUCCESS;  // a stop without its start (tally was full)
  HIPX_HIP(hipEventRecord(p.ev[id][p.used[id]++], s));
  return HIPX_SUCCESS;
}

extern "C" {

int hipxProfileSections(int enable)
{
  HIPX_CHECK_INIT();
  SectionProf &p = sprof();
  p.on = enable != 0;
  for (int k = 0; k < HIPX_PROF_NSECTIONS; k++) p.used[k] = 0;
  return HIPX_SUCCESS;
}

int hipxProfileSectionGet(int id, int *count, double *total_ms)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(id >= 0 && id < HIPX_PROF_NSECTIONS, "unknown profile section");
  SectionProf &p = sprof();
  HIPX_HIP(hipDeviceSynchronize());
  double tot = 0.0;
  for (size_t k = 0; k + 1 < p.used[id]; k += 2) {
    float ms = 0.f;
    HIPX_HIP(hipEventElapsedTime(&ms, p.ev[id][k], p.ev[id][k + 1]));
    tot += ms;
  }
  if (count) *count = (int)(p.used[id] / 2);
  if (total_ms) *total_ms = tot;
  p.used[id] = 0;
  return HIPX_SUCCESS;
}

int hipxInit(int device)
{
  Runtime &r = rt();
  if (r.initialized) return HIPX_SUCCESS;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) return fail(HIPX_ERR_GPU, "no HIP device visible: libhipx has no CPU fallback", __FILE__, __LINE__);
  HIPX_ARG(device >= 0 && device < count, "device ordinal out of range");
  HIPX_HIP(hipSetDevice(device));
  r.device = device;
  HIPX_HIP(hipStreamCreateWithFlags(&r.compute, hipStreamNonBlocking));
  {  // the exchange kernels are tiny and latency-critical: highest priority, so that they are dispatched ahead of the
     // (possibly persistent) SpMV workgroups of the compute stream they overlap with
    int lo = 0, hi = 0;
    HIPX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPX_HIP(hipStreamCreateWithPriority(&r.comm, hipStreamNonBlocking, hi));
  }
  HIPX_HIP(hipMalloc((void **)&r.d_partials, sizeof(double) * HIPX_MAX_RED_SLOTS * kMaxRedVals * kRedBlocks));
  HIPX_HIP(hipMalloc((void **)&r.d_tickets, sizeof(unsigned int) * HIPX_MAX_RED_SLOTS));
  HIPX_HIP(hipMemset(r.d_tickets, 0, sizeof(unsigned int) * HIPX_MAX_RED_SLOTS));
  HIPX_HIP(hipHostMalloc((void **)&r.h_results, sizeof(double) * HIPX_MAX_RED_SLOTS * kMaxRedVals, hipHostMallocMapped));
  memset(r.h_results, 0, sizeof(double) * HIPX_MAX_RED_SLOTS * kMaxRedVals);
  HIPX_HIP(hipHostGetDevicePointer((void **)&r.d_results, r.h_results, 0));
  HIPX_HIP(hipHostMalloc((void **)&r.h_flags, sizeof(unsigned long long) * HIPX_MAX_RED_SLOTS, hipHostMallocMapped));
  memset(r.h_flags, 0, sizeof(unsigned long long) * HIPX_MAX_RED_SLOTS);
  HIPX_HIP(hipHostGetDevicePointer((void **)&r.d_flags, r.h_flags, 0));
  HIPX_HIP(hipMalloc((void **)&r.d_scalars, sizeof(double) * 4096));
  HIPX_HIP(hipMalloc((void **)&r.d_ptrs, sizeof(void *) * 4096));
  HIPX_HIP(hipDeviceSynchronize());
  {
    const char *e = getenv("HIPX_REDUCTIONS");
    r.red_exact   = (e && (!strcmp(e, "exact") || !strcmp(e, "1"))) ? 1 : 0;
  }
  r.initialized = true;
  return HIPX_SUCCESS;
}

int hipxSetReductionMode(int mode)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(mode == HIPX_RED_FAST || mode == HIPX_RED_EXACT, "unknown reduction mode");
  HIPX_HIP(hipStreamSynchronize(rt().compute));  // nothing queued may see the switch half-way (pairs vs sums in the all-reduce chain)
  rt().red_exact = mode;
  return HIPX_SUCCESS;
}
int hipxGetReductionMode(int *mode)
{
  *mode = rt().red_exact;
  return HIPX_SUCCESS;
}

int hipxGetDeviceCount(int *count)
{
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) return fail(HIPX_ERR_GPU, "no HIP device visible: libhipx has no CPU fallback", __FILE__, __LINE__);
  return HIPX_SUCCESS;
}

int hipxFinalize(void)
{
  Runtime &r = rt();
  if (!r.initialized) return HIPX_SUCCESS;
  HIPX_HIP(hipDeviceSynchronize());
  (void)hipFree(r.d_partials);
  (void)hipFree(r.d_tickets);
  (void)hipHostFree(r.h_results);
  (void)hipHostFree(r.h_flags);
  (void)hipFree(r.d_scalars);
  (void)hipFree(r.d_ptrs);
  (void)hipStreamDestroy(r.compute);
  (void)hipStreamDestroy(r.comm);
  r = Runtime();
  return HIPX_SUCCESS;
}

int         hipxIsInitialized(void) { return rt().initialized ? 1 : 0; }
const char *hipxGetErrorString(void) { return rt().errmsg; }

int hipxDeviceName(char *buf, size_t len)
{
  HIPX_CHECK_INIT();
  hipDeviceProp_t p;
  HIPX_HIP(hipGetDeviceProperties(&p, rt().device));
  snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return HIPX_SUCCESS;
}

int hipxDeviceUID(unsigned long long *uid)
{
  HIPX_CHECK_INIT();
  char bus[64] = {0};
  HIPX_HIP(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), rt().device));
  unsigned long long h = 1469598103934665603ULL;  // FNV-1a of "domain:bus:device.function": the same GPU gives the same id in every process,
  for (const char *c = bus; *c; c++) h = (h ^ (unsigned char)*c) * 1099511628211ULL;  // whatever HIP_VISIBLE_DEVICES renumbering the launcher applied
  *uid = h;
  return HIPX_SUCCESS;
}

void *hipxComputeStream(void) { return (void *)rt().compute; }
void *hipxCommStream(void) { return (void *)rt().comm; }

int hipxStreamSynchronize(void)
{
  HIPX_CHECK_INIT();
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  return HIPX_SUCCESS;
}
int hipxDeviceSynchronize(void)
{
  HIPX_CHECK_INIT();
  HIPX_HIP(hipDeviceSynchronize());
  return HIPX_SUCCESS;
}

int hipxMalloc(void **dptr, size_t bytes)
{
  HIPX_CHECK_INIT();
  *dptr = nullptr;
  if (!bytes) return HIPX_SUCCESS;
  hipError_t e = hipMalloc(dptr, bytes);
  if (e == hipErrorOutOfMemory) return fail(HIPX_ERR_MEM, "hipMalloc: out of device memory", __FILE__, __LINE__);
  HIPX_HIP(e);
  return HIPX_SUCCESS;
}
int hipxFree(void *dptr)
{
  HIPX_CHECK_INIT();
  if (dptr) {
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    HIPX_HIP(hipFree(dptr));
  }
  return HIPX_SUCCESS;
}
int hipxMallocHost(void **hptr, size_t bytes)
{
  HIPX_CHECK_INIT();
  HIPX_HIP(hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault));
  return HIPX_SUCCESS;
}
int hipxFreeHost(void *hptr)
{
  HIPX_CHECK_INIT();
  if (hptr) HIPX_HIP(hipHostFree(hptr));
  return HIPX_SUCCESS;
}
int hipxMemcpyHtoD(void *dst, const void *src, size_t bytes)
{
  HIPX_CHECK_INIT();
  if (!bytes) return HIPX_SUCCESS;
  HIPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  return HIPX_SUCCESS;
}
int hipxMemcpyDtoH(void *dst, const void *src, size_t bytes)
{
  HIPX_CHECK_INIT();
  if (!bytes) return HIPX_SUCCESS;
  HIPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  return HIPX_SUCCESS;
}
int hipxMemcpyDtoD(void *dst, const void *src, size_t bytes)
{
  HIPX_CHECK_INIT();
  if (!bytes || dst == src) return HIPX_SUCCESS;
  HIPX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt().compute));
  return HIPX_SUCCESS;
}
int hipxMemset(void *dst, int value, size_t bytes)
{
  HIPX_CHECK_INIT();
  if (!bytes) return HIPX_SUCCESS;
  HIPX_HIP(hipMemsetAsync(dst, value, bytes, rt().compute));
  return HIPX_SUCCESS;
}

int hipxEventCreate(void **ev)
{
  HIPX_CHECK_INIT();
  hipEvent_t e;
  HIPX_HIP(hipEventCreate(&e));
  *ev = (void *)e;
  return HIPX_SUCCESS;
}
int hipxEventDestroy(void *ev)
{
  HIPX_CHECK_INIT();
  HIPX_HIP(hipEventDestroy((hipEvent_t)ev));
  return HIPX_SUCCESS;
}
int hipxEventRecord(void *ev)
{
  HIPX_CHECK_INIT();
  HIPX_HIP(hipEventRecord((hipEvent_t)ev, rt().compute));
  return HIPX_SUCCESS;
}
int hipxEventElapsedMs(void *start, void *stop, float *ms)
{
  HIPX_CHECK_INIT();
  HIPX_HIP(hipEventSynchronize((hipEvent_t)stop));
  HIPX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return HIPX_SUCCESS;
}

}  // extern "C"
