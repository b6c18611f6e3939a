// Diagnostic: where do the waves of co-resident workgroups land?  (HW_ID: wave slot, SIMD, CU, SE; XCC_ID)
// Build: hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void probe(unsigned *out, int spin)
{
  extern __shared__ char smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const unsigned xc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  const int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    out[2 * (blockIdx.x * nw + w)]     = hw;
    out[2 * (blockIdx.x * nw + w) + 1] = xc;
  }
  smem[threadIdx.x] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);  // keep every workgroup resident until all have started
}
__global__ void dpp_probe(double *o, const double *first, const double *v)
{
  const double f = first[threadIdx.x], x = v[threadIdx.x];
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(f), __double2loint(x), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(f), __double2hiint(x), 0x138, 0xf, 0xf, false);
  o[threadIdx.x] = __hiloint2double(hi, lo);
}
static void dpp_check()
{  // wave_shr:1 on a 64-lane wave: lane i gets lane i-1's value (across the 16-lane rows), lane 0 keeps its own `old`
  double h[64], f[64], r[64], *d;
  for (int i = 0; i < 64; i++) { h[i] = 100.0 + i; f[i] = -1.0 - i; }
  hipMalloc(&d, 3 * 64 * sizeof(double));
  hipMemcpy(d + 64, f, sizeof(f), hipMemcpyHostToDevice);
  hipMemcpy(d + 128, h, sizeof(h), hipMemcpyHostToDevice);
  dpp_probe<<<1, 64>>>(d, d + 64, d + 128);
  hipMemcpy(r, d, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; i++) bad += r[i] != (i ? h[i - 1] : f[0]);
  printf("== DPP wave_shr:1: %s (lane 0 %.0f, lane 1 %.0f, lane 16 %.0f, lane 32 %.0f, lane 63 %.0f)\n", bad ? "WRONG" : "ok", r[0], r[1], r[16], r[32], r[63]);
  hipFree(d);
}
static void run(int nt, int lds, int grid)
{
  const int nw = nt / 64;
  unsigned *d;
  hipMalloc(&d, sizeof(unsigned) * 2 * grid * nw);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  probe<<<grid, nt, lds>>>(d, 20000);  // 100 MHz wall clock: 200 us
  std::vector<unsigned> h(2 * grid * nw);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;  // CU key -> workgroups
  int hist[8][4] = {};
  for (int b = 0; b < grid; b++) {
    const unsigned hw = h[2 * b * nw], xc = h[2 * b * nw + 1] & 15;
    cu[(xc << 8) | ((hw >> 8) & 0xff)].push_back(b);
    for (int w = 0; w < nw; w++) hist[w][(h[2 * (b * nw + w)] >> 4) & 3]++;
  }
  printf("== %d threads, %d B LDS, %d workgroups: %zu distinct CUs\n", nt, lds, grid, cu.size());
  for (int w = 0; w < nw; w++) printf("  wave %d: SIMD histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  int npair = 0, same0 = 0, slotpar = 0, shown = 0;
  std::map<int, int> per;
  for (auto &kv : cu) {
    per[(int)kv.second.size()]++;
    if (kv.second.size() == 2) {
      const int a = kv.second[0], b = kv.second[1];
      npair++;
      if (((h[2 * a * nw] >> 4) & 3) == ((h[2 * b * nw] >> 4) & 3)) same0++;
      if (((h[2 * a * nw]) & 1) != ((h[2 * b * nw]) & 1)) slotpar++;
      if (shown++ < 4) {
        printf("  CU %03x: wg %d waves (simd:slot)", kv.first, a);
        for (int w = 0; w < nw; w++) printf(" %u:%u", (h[2 * (a * nw + w)] >> 4) & 3, h[2 * (a * nw + w)] & 15);
        printf(" | wg %d", b);
        for (int w = 0; w < nw; w++) printf(" %u:%u", (h[2 * (b * nw + w)] >> 4) & 3, h[2 * (b * nw + w)] & 15);
        printf("\n");
      }
    }
  }
  for (auto &kv : per) printf("  CUs with %d workgroups: %d\n", kv.first, kv.second);
  printf("  pairs %d: wave 0 of both on the same SIMD in %d; wave-slot parity differs in %d\n", npair, same0, slotpar);
  hipFree(d);
}
int main()
{
  dpp_check();
  run(256, 78 * 1024, 512);
  run(128, 78 * 1024, 512);
  return 0;
}
