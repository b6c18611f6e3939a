// Diagnostic: how fast does a CU's address pipeline take loads / stores whose lanes do not share cache lines?
//   mode 0: 8-byte sc1 loads, every lane its own line (4 KiB apart)            -- the loaders' polls of the first strand kernels
//   mode 1: 16-byte sc1 loads, every lane its own line                           -- the loaders' polls now
//   mode 2: 16-byte sc1 loads, 4 adjacent lanes share one 64-byte line, lines 4 KiB apart (16 lines per instruction)
//   mode 3: 16-byte sc1 loads, 64 lanes contiguous (1 KiB per instruction)
//   mode 4: 8-byte sc1 stores, every lane its own line;   mode 5: 16-byte sc1 stores, every lane its own line
//   mode 6: 16-byte PLAIN loads (no sc1: L1 allowed), every lane its own line, a new line every request
//   mode 7: 16-byte plain loads, every lane its own line, the four quarters of that line in four consecutive requests
//           (what the loaders' operand loads do: 1 miss + 3 hits in the CU's L1?)
//   mode 8: 16-byte sc1 loads, the four quarters of a lane's own line in four consecutive requests (a far poll of 8 rows)
//   mode 9: 16-byte sc1 stores, 4 adjacent lanes share one 64-byte line (a line-granular flush)
//   mode 10: 16-byte plain stores, every lane its own line (the t store)
//   mode 11 / 12: 16-byte sc1 / plain stores, the four quarters of a lane's own line in four consecutive requests (a line of 8 rows
//           kept in registers and stored at once)
// Two 64-lane workgroups per CU (512 workgroups), each issues REPS x 8 requests; lines are re-used across repetitions (L2 hits):
// the rate is the request path's, not HBM's.  Build: hipcc --offload-arch=gfx950 -O2 -o ta_probe ta_probe.hip  (petsc_amd/build.py)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void probe(char *base, int reps, int *sink)
{
  const int lane = threadIdx.x;
  char     *w    = base + (size_t)blockIdx.x * (64 * 4096);
  const bool quad = MODE == 2 || MODE == 9;
  size_t    off  = quad ? (size_t)(lane >> 2) * 4096 + (lane & 3) * 16 : (MODE == 3 ? (size_t)lane * 16 : (size_t)lane * 4096);
  int       acc  = 0;
  for (int r = 0; r < reps; r++) {
    int4v v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      // modes 7 / 8: requests k = 0..3 and 4..7 walk the four quarters of one line each; the others take a new line per request
      const size_t step = (MODE == 7 || MODE == 8 || MODE == 11 || MODE == 12) ? (size_t)((k & 3) * 16 + (k >> 2) * 64 + (r & 7) * 128) : (size_t)(k * 64 + (r & 7) * 512);
      const char  *p    = w + off + step;
      if (MODE == 0) {
        int2v t;
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(t) : "v"(p) : "memory");
        v[k].x = t.x; v[k].y = t.y; v[k].z = 0; v[k].w = 0;
      } else if (MODE <= 3 || MODE == 8) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v[k]) : "v"(p) : "memory");
      } else if (MODE == 6 || MODE == 7) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[k]) : "v"(p) : "memory");
      } else if (MODE == 4) {
        int2v t = {r, k};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
        v[k] = int4v{0, 0, 0, 0};
      } else if (MODE == 10 || MODE == 12) {
        int4v t = {r, k, r, k};
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
        v[k] = int4v{0, 0, 0, 0};
      } else {
        int4v t = {r, k, r, k};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
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
static void run(char *d, int *sink, const char *what, int grid = 512)
{
  const int reps = 2000;
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
  const double per_cu = (grid / 256.0) * reps * 8;  // wave-instructions per CU
  printf("mode %2d %-66s %d/CU %8.3f ms  %7.1f ns per wave-instruction per CU  (%.2f lane-requests per ns per CU)\n", MODE, what, grid / 256, ms, ms * 1e6 / per_cu,
         64.0 * per_cu / (ms * 1e6));
}
int main()
{
  char *d;
  int  *sink;
  hipMalloc(&d, (size_t)512 * 64 * 4096);
  hipMalloc(&sink, 4);
  hipMemset(d, 0, (size_t)512 * 64 * 4096);
  run<0>(d, sink, "8 B sc1 loads, a line per lane");
  run<1>(d, sink, "16 B sc1 loads, a line per lane");
  run<1>(d, sink, "16 B sc1 loads, a line per lane, ONE wave per CU", 256);
  run<2>(d, sink, "16 B sc1 loads, 4 lanes per 64 B line");
  run<3>(d, sink, "16 B sc1 loads, contiguous");
  run<4>(d, sink, "8 B sc1 stores, a line per lane");
  run<5>(d, sink, "16 B sc1 stores, a line per lane");
  run<6>(d, sink, "16 B plain loads, a line per lane, new line per request");
  run<7>(d, sink, "16 B plain loads, a line per lane, 4 quarters of the line in a row");
  run<8>(d, sink, "16 B sc1 loads, a line per lane, 4 quarters of the line in a row");
  run<9>(d, sink, "16 B sc1 stores, 4 lanes per 64 B line");
  run<10>(d, sink, "16 B plain stores, a line per lane");
  run<11>(d, sink, "16 B sc1 stores, a line per lane, 4 quarters of the line in a row");
  run<12>(d, sink, "16 B plain stores, a line per lane, 4 quarters of the line in a row");
  return 0;
}
