// Diagnostic (round 5, plane-march PCSOR): what the recurrence's building blocks cost one wave on gfx950, in shader clocks (s_memtime) and in
// ns (s_memrealtime, 100 MHz): dependent v_add_f64 / v_mul_f64 chains, the 13-term "mul, subtract" chain of a 27-point row, LDS read latency,
// LDS write -> read of the same wave, the DPP wavefront shift, s_sleep 1, a uniform-address counter read.
// One workgroup of 64 threads (optionally NW more idle-polling waves on the CU: argv[1]).  Build: petsc_amd/build.py build_diag().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) double lds_double;
__global__ void probe(double *out, unsigned long long *clk, int reps, int nspin)
{
  __shared__ double sm[4096];
  __shared__ int    flag;
  lds_double *L = (lds_double *)sm;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) flag = 0;
  for (int q = threadIdx.x; q < 4096; q += blockDim.x) sm[q] = 1.0 + q * 1e-9;
  __syncthreads();
  if (wave > 0) {  // pollers: what the helper waves of the plane-march kernel do while they wait
    while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    return;
  }
  double x = 1.0 + lane * 1e-3, y = 0.999999, acc = 0.0;
  unsigned long long c0, c1, r0, r1;
  int slot = 0;
  auto rec = [&](unsigned long long a, unsigned long long b, unsigned long long ra, unsigned long long rb) {
    if (lane == 0) {
      clk[2 * slot]     = b - a;
      clk[2 * slot + 1] = rb - ra;
    }
    slot++;
  };
#define T0() r0 = __builtin_readcyclecounter(); c0 = wall_clock64()
#define T1() c1 = wall_clock64(); r1 = __builtin_readcyclecounter(); rec(r0, r1, c0, c1)
  // 0: dependent v_add_f64 chain
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = x + y;
  }
  T1();
  // 1: dependent v_mul_f64 chain
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = x * y;
  }
  T1();
  // 2: 13 x (independent mul, dependent subtract) + scale: one 27-point row
  double c[13], v[13];
#pragma unroll
  for (int k = 0; k < 13; k++) {
    c[k] = 0.01 * (k + 1);
    v[k] = 1.0 + 0.001 * k + lane;
  }
  T0();
  for (int r = 0; r < reps; r++) {
    double s = x;
#pragma unroll
    for (int k = 0; k < 13; k++) s = s - c[k] * v[k];
    x = s * y;
    v[12] = x;
  }
  T1();
  // 3: LDS read latency (dependent address chain)
  int idx = lane;
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const double d = L[idx];
      idx            = (idx + (int)d + 64) & 4095;
    }
  }
  T1();
  acc += idx;
  // 4: LDS write then read of the same slot by the same wave
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      L[lane + 64 * k] = x;
      asm volatile("" ::: "memory");
      x = L[lane + 64 * k] + 1e-9;
    }
  }
  T1();
  // 5: DPP wavefront shift chain
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int lo = __builtin_amdgcn_update_dpp(__double2loint(y), __double2loint(x), 0x138, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(__double2hiint(y), __double2hiint(x), 0x138, 0xf, 0xf, false);
      x            = __hiloint2double(hi, lo) + y;
    }
  }
  T1();
  // 6: s_sleep 1 x 16
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 16; k++) __builtin_amdgcn_s_sleep(1);
  }
  T1();
  // 7: uniform-address LDS counter read, waited for (what a readiness check costs)
  int cs = 0;
  T0();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      cs += __hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  T1();
  acc += cs;
  if (lane == 0) {
    __hip_atomic_store(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    out[0] = x + acc;
  }
}
int main(int argc, char **argv)
{
  const int nspin = argc > 1 ? atoi(argv[1]) : 0, reps = 2000;
  double *out;
  unsigned long long *clk, h[32];
  hipMalloc(&out, 64);
  hipMalloc(&clk, sizeof(h));
  const char *name[8] = {"dependent v_add_f64", "dependent v_mul_f64", "27-pt row: 13 x (mul, sub) + mul", "LDS read (dependent)", "LDS write -> read", "DPP wave_shr:1 (2 movs) + add", "s_sleep 1",
                         "LDS counter read + wait"};
  const int   per[8]  = {16, 16, 1, 8, 8, 16, 16, 16};
  for (int pass = 0; pass < 2; pass++) {
    hipMemset(clk, 0, sizeof(h));
    probe<<<1, 64 * (1 + nspin), 0, 0>>>(out, clk, reps, nspin);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  printf("one wave + %d polling waves on the CU (reps %d)\n", nspin, reps);
  for (int k = 0; k < 8; k++)
    printf("  %-34s %8.1f shader clocks  %8.2f ns  per op  (clock %.2f GHz)\n", name[k], (double)h[2 * k] / (reps * per[k]), (double)h[2 * k + 1] * 10.0 / (reps * per[k]),
           h[2 * k + 1] ? (double)h[2 * k] / ((double)h[2 * k + 1] * 10.0) : 0.0);
  return 0;
}
