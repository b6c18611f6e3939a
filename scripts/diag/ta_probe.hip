// Diagnostic: how fast does a CU's address pipeline take agent-scope (sc1) loads whose lanes do not share cache lines?
//   mode 0: 8-byte loads, every lane its own line (4 KiB apart)           -- the loaders' polls before
//   mode 1: 16-byte loads, every lane its own line                          -- the loaders' polls now
//   mode 2: 16-byte loads, 4 adjacent lanes share one 64-byte line, lines 4 KiB apart (16 lines per instruction)
//   mode 3: 16-byte loads, 64 lanes contiguous (1 KiB per instruction)
//   mode 4: 8-byte STORES (sc1), every lane its own line;  mode 5: 16-byte stores, every lane its own line
// Two 64-lane workgroups per CU (512 workgroups), each issues REPS x 8 requests; lines are re-used (L2 hits): the rate is the
// request path's, not HBM's.  Build: hipcc --offload-arch=gfx950 -O2 -o ta_probe ta_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void probe(char *base, int reps, int *sink)
{
  const int lane = threadIdx.x;
  char     *w    = base + (size_t)blockIdx.x * (64 * 4096);
  size_t    off  = MODE == 2 ? (size_t)(lane >> 2) * 4096 + (lane & 3) * 16 : (MODE == 3 ? (size_t)lane * 16 : (size_t)lane * 4096);
  int       acc  = 0;
  for (int r = 0; r < reps; r++) {
    int4v v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const char *p = w + off + (size_t)(k * 64 + (r & 7) * 512);
      if (MODE == 0) {
        int2v t;
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(t) : "v"(p) : "memory");
        v[k].x = t.x; v[k].y = t.y; v[k].z = 0; v[k].w = 0;
      } else if (MODE <= 3) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v[k]) : "v"(p) : "memory");
      } else if (MODE == 4) {
        int2v t = {r, k};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
        v[k] = int4v{0, 0, 0, 0};
      } else {
        int4v t = {r, k, r, k};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
        v[k] = int4v{0, 0, 0, 0};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#pragma unroll
    for (int k = 0; k < 8; k++) acc += v[k].x;
  }
  if (acc == 0x12345678) sink[0] = acc;
}
template <int MODE>
static void run(char *d, int *sink, const char *what)
{
  const int reps = 2000, grid = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE><<<grid, 64>>>(d, 100, sink);
  hipEventRecord(e0);
  probe<MODE><<<grid, 64>>>(d, reps, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = 2.0 * reps * 8;  // wave-instructions per CU
  printf("mode %d %-58s %8.3f ms  %7.1f ns per wave-instruction per CU  (%.2f lane-requests per ns per CU)\n", MODE, what, ms, ms * 1e6 / per_cu, 64.0 * per_cu / (ms * 1e6));
}
int main()
{
  char *d;
  int  *sink;
  hipMalloc(&d, (size_t)512 * 64 * 4096);
  hipMalloc(&sink, 4);
  hipMemset(d, 0, (size_t)512 * 64 * 4096);
  run<0>(d, sink, "8 B loads, a line per lane");
  run<1>(d, sink, "16 B loads, a line per lane");
  run<2>(d, sink, "16 B loads, 4 lanes per 64 B line");
  run<3>(d, sink, "16 B loads, contiguous");
  run<4>(d, sink, "8 B stores, a line per lane");
  run<5>(d, sink, "16 B stores, a line per lane");
  return 0;
}
