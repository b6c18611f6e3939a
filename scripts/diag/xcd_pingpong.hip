// Diagnostic (round 6, dependency-driven PCSOR): how long a value takes from one workgroup to another -- agent-scope relaxed atomic store, agent-scope relaxed
// atomic load in a poll loop (what sor_publish / sor_poll of csrc/hipx_sor.hip do) -- when the two workgroups sit on the SAME XCD (blocks b and b + 8 of a launch:
// hardware block b runs on XCD b mod 8) and when they sit on DIFFERENT ones (blocks b and b + 1).  A ping-pong of N round trips; ns per one-way hop.
// Also with workgroup-scope (sc0) loads / stores for the same-XCD pair -- NOT a valid hand-off by the memory model (a load may hit the CU's vector L1), printed
// to see whether it is faster at all.  Build: petsc_amd/build.py build_diag().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SCOPE>
__global__ void pingpong(unsigned long long *flags, int partner_delta, int n, unsigned long long *out)
{
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  int role = -1;
  if (b == 0) role = 0;
  else if (b == partner_delta) role = 1;
  if (role < 0) return;
  unsigned long long *mine = flags + 32 * role, *theirs = flags + 32 * (1 - role);  // separate 256-byte lines
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= n; i++) {
    if (role == 0) {
      __hip_atomic_store(theirs, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)i) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 200000000ull) { if (role == 0) *out = ~0ull; return; }  // 2 s: gave up (a stale line that never refreshes)
      }
    } else {
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)i) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 200000000ull) return;
      }
      __hip_atomic_store(theirs, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
    }
  }
  if (role == 0) *out = wall_clock64() - t0;
}
int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long *flags, *out, h;
  hipMalloc(&flags, 512);
  hipMalloc(&out, 8);
  const struct { const char *what; int delta; int scope; } cases[] = {
    {"different XCDs (blocks 0, 1), agent scope", 1, 0}, {"same XCD (blocks 0, 8), agent scope", 8, 0}, {"same XCD (blocks 0, 16), agent scope", 16, 0},
    {"different XCDs (blocks 0, 3), agent scope", 3, 0}, {"same XCD (blocks 0, 8), workgroup scope [not a valid hand-off]", 8, 1}};
  for (auto &c : cases) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(flags, 0, 512);
      hipMemset(out, 0, 8);
      if (c.scope == 0) pingpong<__HIP_MEMORY_SCOPE_AGENT><<<32, 64>>>(flags, c.delta, n, out);
      else pingpong<__HIP_MEMORY_SCOPE_WORKGROUP><<<32, 64>>>(flags, c.delta, n, out);
      if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", c.what); return 1; }
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      if (rep) {
        if (h == ~0ull) printf("%-70s gave up after 2 s (the poll never saw the store)\n", c.what);
        else printf("%-70s %8.1f ns per one-way hop\n", c.what, (double)h * 10.0 / (2.0 * n));
        fflush(stdout);
      }
    }
  }
  return 0;
}
