// hipx_sorbox.hip -- PCSOR on box stencils as a PLANE MARCH (round 5): MatSOR_SeqAIJ's zero-guess sweeps (aij.c:1930-1958) on
// constant-coefficient 3-D box stencils in natural ordering (7-point ... 27-point: BASELINE config 3's operator), bit-identical to the
// reference's loop, with every operand of the recurrence in LDS or registers.
//
// Why another schedule.  The strand schedule (hipx_sor.hip) keeps ONE x-line per lane too, but every lane reaches its operands through
// global memory one cache line at a time (lanes are 8 nx bytes apart): its row period is bound by the CU's request path (~0.5 us per
// row), and every plane hop costs a loader hand-off through L2 (~5 us).  Here
//   * a workgroup owns P = 4 CONSECUTIVE PLANES of a block of 64 lines (wave w <-> plane P c + w, lane s <-> line 64 J - k + s): plane
//     k - 1's values reach plane k through an LDS ring, three ds_reads per row, never through memory;
//   * the block boundaries move ONE LINE PER PLANE (the line range of block J at plane k is [64 J - k, 64 J - k + 64)), so the line
//     j + 1 of plane k - 1 a row needs is always in its own block and every cross-workgroup dependency points to block J - 1 or to the
//     chunk of planes below: hand-offs through memory add pipeline-fill latency once per block / chunk, not per plane hop;
//   * right-hand side, results and the two west lines / the south plane a workgroup needs from its neighbours are moved by HELPER waves in
//     16-byte accesses, four lanes per 64-byte line segment, transposed through LDS rings -- the compute waves never touch memory;
//   * lanes run in lockstep with a fixed skew: in step t wave w lane s is at row i = t - 2 - 2 s - 4 w.  The previous line (lane s - 1)
//     is then two rows ahead, the lower plane's next line (wave w - 1, lane s) four: exactly what the 27-point stencil needs.
// Arithmetic: sum = rhs; sum -= a_e * x_e for the dependency-side entries e in CSR order (forward: (dk, dj, di) ascending; backward:
// the mirrored grid with the order reversed); x = sum * idiag -- products and differences rounded separately (-ffp-contract=off), the
// reference's operations in the reference's order.  Rows on the boundary of the grid have fewer entries: their missing neighbours read
// the ZERO ELEMENT z0 = -0.0 (all couplings negative) or +0.0 (all positive): a * z0 = +0.0 and sum - (+0.0) = sum for every sum, so the
// row is evaluated exactly as the reference evaluates its shorter list.  That needs: every row's entries = the interior row's entries
// present in the grid, with the interior row's values (verified per row at set-up), couplings of one sign, one diagonal value.
// Anything else keeps the strand / level schedules.  scripts/sor_box_model.py is the CPU model of this schedule (ring depths, wait
// conditions, index arithmetic) checked against the reference loop under a randomised scheduler.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "hipx_internal.h"

using namespace hipx;

namespace {

constexpr int BX_P = 4, BX_G = 8;                             // planes per workgroup, rows per staging group
constexpr int BX_RX = 16, BX_RB = 32, BX_RS = 32, BX_RW = 32;  // ring rows: own x lines / rhs / south plane / west lines
constexpr int BX_THREADS = 2 * BX_P * 64;                      // P compute waves + P helper waves
// LDS layout (doubles)
constexpr int BX_OX = 0;                                        // X [P][64][RX + 1]
constexpr int BX_OS = BX_OX + BX_P * 64 * (BX_RX + 1);          // S [66][RS + 1]
constexpr int BX_OW = BX_OS + 66 * (BX_RS + 1);                 // W [P][2][RW]
constexpr int BX_OB = BX_OW + BX_P * 2 * BX_RW;                 // B [P][64][RB + 1]
constexpr int BX_OT = BX_OB + BX_P * 64 * (BX_RB + 1);          // TR[P][64][RX + 1]   (forward sweeps only)
constexpr int BX_OC = BX_OT + BX_P * 64 * (BX_RX + 1);          // counters (ints): cprog[P + 1], hprog[P], flush[P], abort
constexpr int BX_LDS_BYTES = BX_OC * 8 + 64;
constexpr unsigned long long BX_SENTINEL = 0x7FF4DEADBEEF0001ULL;  // = SOR_SENTINEL of hipx_sor.hip (sor_fill_kernel fills x with it)
constexpr long long          BX_SPIN_TICKS = 400000000LL;           // 4 s of the 100 MHz wall clock

struct BoxParams {
  int          nx, ny, nz, nb, nch, T, ngroups, xfull;
  long long    m;
  double       coef[13];  // dependency-side couplings at the canonical positions e = 9 (dk + 1) + 3 (dj + 1) + (di + 1) of the LOGICAL grid
  double       idiag, z0, omw;  // omega / (d + shift) [1 / d when omega == 1, shift <= 0]; zero element; 1 - omega
  const double *rhs;            // forward: b; backward after forward: t; backward alone: b
  double       *xout, *tout;
  const int2   *order;          // (J, c) of ticket n
  unsigned int *ctl;            // [0] ticket, [1] error
};

typedef __attribute__((address_space(3))) double bx_lds_double;
typedef __attribute__((address_space(3))) int    bx_lds_int;
typedef double bx_double2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int bx_cnt_load(bx_lds_int *p) { return __hip_atomic_load((int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void bx_cnt_store(bx_lds_int *p, int v) { __hip_atomic_store((int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// wait until *p >= want (uniform over the wave); false when the launch was aborted
__device__ __forceinline__ bool bx_wait_ge(bx_lds_int *p, int want, bx_lds_int *abortw, unsigned int *gerr)
{
  int       spins = 0;
  long long t0    = 0;
  while (bx_cnt_load(p) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (bx_cnt_load(abortw)) return false;
    if ((++spins & 0x3ff) == 0) {
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      if (__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > BX_SPIN_TICKS) {
        __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bx_cnt_store(abortw, 1);
        return false;
      }
    }
  }
  return true;
}

// a pair of rows of another workgroup's line out of global x: 16-byte agent-scope load, repeated until neither half is the sentinel
__device__ __forceinline__ bx_double2 bx_poll2(const double *p, bx_lds_int *abortw, unsigned int *gerr)
{
  bx_double2 v;
  int        spins = 0;
  long long  t0    = 0;
  for (;;) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if ((unsigned long long)__double_as_longlong(v.x) != BX_SENTINEL && (unsigned long long)__double_as_longlong(v.y) != BX_SENTINEL) break;
    __builtin_amdgcn_s_sleep(2);
    if (bx_cnt_load(abortw)) break;
    if ((++spins & 0xff) == 0) {
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      if (__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > BX_SPIN_TICKS) {
        __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bx_cnt_store(abortw, 1);
        break;
      }
    }
  }
  return v;
}

__device__ __forceinline__ void bx_store2_sc1(double *p, bx_double2 v)
{
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");  // write-through: every 8-byte half is its own ready flag
}

// REV: the backward sweep = the forward schedule on the mirrored grid (logical row r <-> physical row m - 1 - r), entry order reversed.
// EM: bit e set = the interior row has the dependency-side entry at canonical position e.  KIND as in hipx_sor.hip: 0 forward zero-guess
// (t = sum, x = sum idiag), 1 backward after forward (x = (1 - w) (t idiag) + sum idiag: aij.c:1955 with the forward result x = t idiag
// re-formed from t), 2 backward zero-guess alone.
template <bool REV, int EM, int KIND>
__global__ __launch_bounds__(BX_THREADS, 1) void sor_box_kernel(const BoxParams Q)
{
  extern __shared__ double bx_smem[];
  bx_lds_double *L   = (bx_lds_double *)bx_smem;
  bx_lds_int    *cnt = (bx_lds_int *)(L + BX_OC);
  bx_lds_int    *cprog = cnt, *hprog = cnt + (BX_P + 1), *flushp = cnt + (2 * BX_P + 1), *abortw = cnt + (3 * BX_P + 1), *tick = cnt + (3 * BX_P + 2);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    for (int q = 0; q < BX_P; q++) {
      bx_cnt_store(cprog + q, 0);
      bx_cnt_store(hprog + q, 0);
      bx_cnt_store(flushp + q, 0);
    }
    bx_cnt_store(cprog + BX_P, 0x3fffffff);  // "the wave above the last one": never holds anybody back
    bx_cnt_store(abortw, 0);
    bx_cnt_store(tick, (int)atomicAdd(Q.ctl, 1u));  // workgroups take their (block, chunk) in ticket order: every dependency has an earlier ticket
  }
  __syncthreads();
  const int2 jc = Q.order[bx_cnt_load(tick)];
  const int  J = jc.x, c = jc.y, k0 = BX_P * c;
  const int  nx = Q.nx, ny = Q.ny, nz = Q.nz, T = Q.T;
  const double z0 = Q.z0;
  unsigned int *gerr = Q.ctl + 1;

  if (wave < BX_P) {
    // ------------------------------------------------------------------------------------------------ compute wave w: plane k0 + w
    const int  w = wave, s = lane, k = k0 + w;
    const int  j = 64 * J - k + s;
    const bool valid = j >= 0 && j < ny && k < nz;
    double Lr[3][2], Mr[2], xp = z0;  // lower plane lines j-1, j, j+1 at rows i-1, i; previous line at rows i-1, i; own x(i-1)
#pragma unroll
    for (int l = 0; l < 3; l++) Lr[l][0] = Lr[l][1] = z0;
    Mr[0] = Mr[1] = z0;
    // ring bases of this lane's four neighbour lines (element offsets; the row slot is added per step)
    int nb_base[4], nb_mask[4];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int q = s + d;  // line index in the lower plane's slot
      if (w == 0) {
        nb_base[d] = BX_OS + q * (BX_RS + 1);
        nb_mask[d] = BX_RS - 1;
      } else if (q < 2) {
        nb_base[d] = BX_OW + ((w - 1) * 2 + q) * BX_RW;
        nb_mask[d] = BX_RW - 1;
      } else {
        nb_base[d] = BX_OX + ((w - 1) * 64 + (q - 2)) * (BX_RX + 1);
        nb_mask[d] = BX_RX - 1;
      }
    }
    if (s == 0) {
      nb_base[3] = BX_OW + (w * 2 + 1) * BX_RW;
      nb_mask[3] = BX_RW - 1;
    } else {
      nb_base[3] = BX_OX + (w * 64 + (s - 1)) * (BX_RX + 1);
      nb_mask[3] = BX_RX - 1;
    }
    const int xo = BX_OX + (w * 64 + s) * (BX_RX + 1), bo = BX_OB + (w * 64 + s) * (BX_RB + 1), to = BX_OT + (w * 64 + s) * (BX_RX + 1);
    bool alive = true;
    for (int t = 0; t < T && alive; t++) {
      if (w > 0) alive = alive && bx_wait_ge(cprog + (w - 1), t - 2 < T ? t - 2 : T, abortw, gerr);  // the lower plane's row i + 1 of line j + 1: relaxed in wave w - 1's step t - 3
      alive = alive && bx_wait_ge(hprog + w, t + 1, abortw, gerr);                                    // this step's right-hand side and west / south rows are staged
      alive = alive && bx_wait_ge(cprog + (w + 1), t - 8, abortw, gerr);                              // x ring: plane k + 1 reads a row up to 7 steps after it was written
      alive = alive && bx_wait_ge(flushp + w, t - BX_RX + 1, abortw, gerr);                           // the rows this step overwrites have left for memory
      if (!alive) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int i = t - 2 - 2 * s - 4 * w;
      if (i >= -1 && i < nx) {
        double N[4];
        const bool inr = i + 1 < nx;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const double v = L[nb_base[d] + ((i + 1) & nb_mask[d])];
          N[d]           = inr ? v : z0;
        }
        if (i >= 0) {
          const double rhs = L[bo + (i & (BX_RB - 1))];
          double       sum = rhs;
          auto val = [&](int e) -> double {
            if (e < 9) {
              const int l = e / 3, cc = e % 3;
              return cc < 2 ? Lr[l][cc] : N[l];
            }
            if (e < 12) return (e - 9) < 2 ? Mr[e - 9] : N[3];
            return xp;
          };
#pragma unroll
          for (int q = 0; q < 13; q++) {
            const int e = REV ? 12 - q : q;
            if ((EM >> e) & 1) sum = sum - Q.coef[e] * val(e);
          }
          double xv;
          if (KIND == 1) xv = Q.omw * (rhs * Q.idiag) + sum * Q.idiag;
          else xv = sum * Q.idiag;
          if (!valid) xv = z0;  // a lane without a line publishes the zero element for its neighbours
          L[xo + (i & (BX_RX - 1))] = xv;
          if (KIND == 0) L[to + (i & (BX_RX - 1))] = sum;
          xp = xv;
        }
#pragma unroll
        for (int l = 0; l < 3; l++) {
          Lr[l][0] = Lr[l][1];
          Lr[l][1] = N[l];
        }
        Mr[0] = Mr[1];
        Mr[1] = N[3];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) bx_cnt_store(cprog + w, t + 1);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- helper wave of plane k0 + w
  const int  w = wave - BX_P, k = k0 + w;
  const bool plane_ok = k < nz;
  const long long nxl = nx, nyl = ny;
  auto phys = [&](int r, int jj, int kk) -> long long {  // element index of logical row (r, jj, kk); pairs (r, r + 1), r even, are 16-byte aligned
    const long long lr = (long long)r + nxl * ((long long)jj + nyl * (long long)kk);
    return REV ? Q.m - 2 - lr : lr;  // REV: the pair (r, r + 1) lies at m - 2 - lr, halves swapped
  };
  int  g = 0, fg = 0;  // next group to stage / to flush
  bool alive = true;
  // right-hand side: the loads of group g + 1 are in flight while group g's halo rows are polled and results are flushed
  bx_double2 pre[4];
  auto rhs_issue = [&](int gg) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int  s = 16 * p + (lane >> 2), qd = lane & 3, jj = 64 * J - k + s;
      const int  r = BX_G * gg - 2 - 2 * s - 4 * w + 2 * qd;
      const bool ok = plane_ok && r >= 0 && r < nx && jj >= 0 && jj < ny;
      pre[p]        = *reinterpret_cast<const bx_double2 *>(ok ? Q.rhs + phys(r, jj, k) : Q.rhs);  // (a lane without a row reads element 0: never stored)
    }
  };
  auto rhs_store = [&](int gg) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int s = 16 * p + (lane >> 2), qd = lane & 3, jj = 64 * J - k + s;
      const int r = BX_G * gg - 2 - 2 * s - 4 * w + 2 * qd;
      if (plane_ok && r >= 0 && r < nx && jj >= 0 && jj < ny) {
        const int bo = BX_OB + (w * 64 + s) * (BX_RB + 1);
        L[bo + (r & (BX_RB - 1))]       = REV ? pre[p].y : pre[p].x;
        L[bo + ((r + 1) & (BX_RB - 1))] = REV ? pre[p].x : pre[p].y;
      }
    }
  };
  if (Q.ngroups > 0) rhs_issue(0);
  while (alive && (g < Q.ngroups || fg * BX_G < T)) {
    bool did = false;
    // ---- stage group g: everything the compute wave reads in steps [8 g, 8 g + 8)
    if (g < Q.ngroups && bx_cnt_load(cprog + w) >= BX_G * g - (BX_RB - BX_G) && bx_cnt_load(cprog + w) >= BX_G * g - (BX_RW - 16) && bx_cnt_load(cprog + (w + 1)) >= BX_G * g - (BX_RW - 16)) {
      did = true;
      // (a) right-hand side rows [8 g - 2 - 2 s - 4 w, + 8) of the 64 lines: four lanes per line, a pair of rows each
      rhs_store(g);
      if (g + 1 < Q.ngroups) rhs_issue(g + 1);
      // (b) the two west lines of plane k (lines 64 J - k - 2, - 1: block J - 1's last lanes), rows [8 g - 4 w, + 8): task 0 of lanes 0-7;
      // (c) helper 0: the south plane k0 - 1 (66 lines: the chunk below and, there, block J - 1's last lanes), line index q rows
      //     [8 g - 2 max(q - 2, 0), + 8): tasks 1-5.  All loads of a group are issued together (16 bytes, agent scope), one wait; a half that still
      //     holds the sentinel is polled on its own afterwards
      constexpr int NT = 6;
      bx_double2    hv[NT];
      const double *hp[NT];
      bool          hin[NT], hmem[NT];  // the task exists for this lane / its rows come from memory (else: the zero element)
      int           hoff[NT], hmask[NT], hr[NT];
#pragma unroll
      for (int tk = 0; tk < NT; tk++) {
        int q, qd, jj, kk, r;
        if (tk == 0) {
          q = lane >> 2, qd = lane & 3, jj = 64 * J - k - 2 + q, kk = k;
          r         = BX_G * g - 4 * w + 2 * qd;
          hin[tk]   = lane < 8 && r >= 0 && r < nx;
          hmem[tk]  = hin[tk] && plane_ok && jj >= 0 && jj < ny;
          hoff[tk]  = BX_OW + (w * 2 + (q & 1)) * BX_RW;
          hmask[tk] = BX_RW - 1;
        } else {
          const int task = lane + 64 * (tk - 1);
          q = task >> 2, qd = task & 3, jj = 64 * J - (k0 - 1) + q - 2, kk = k0 - 1;
          r         = BX_G * g - 2 * (q > 2 ? q - 2 : 0) + 2 * qd;
          hin[tk]   = w == 0 && task < 66 * 4 && r >= 0 && r < nx;
          hmem[tk]  = hin[tk] && k0 > 0 && jj >= 0 && jj < ny;
          hoff[tk]  = BX_OS + (q < 66 ? q : 0) * (BX_RS + 1);
          hmask[tk] = BX_RS - 1;
        }
        hr[tk] = r;
        hp[tk] = hmem[tk] ? Q.xout + phys(r, jj, kk) : Q.xout;
      }
      if (w == 0) {
#pragma unroll
        for (int tk = 0; tk < NT; tk++) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(hv[tk]) : "v"(hp[tk]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5])::"memory");
      } else {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(hv[0]) : "v"(hp[0]) : "memory");
      }
#pragma unroll
      for (int tk = 0; tk < NT; tk++) {
        if (tk > 0 && w != 0) break;
        if (hin[tk]) {
          bx_double2 v = hv[tk];
          if (hmem[tk]) {
            if ((unsigned long long)__double_as_longlong(v.x) == BX_SENTINEL || (unsigned long long)__double_as_longlong(v.y) == BX_SENTINEL) v = bx_poll2(hp[tk], abortw, gerr);
            if (REV) {
              const double tmp = v.x;
              v.x              = v.y;
              v.y              = tmp;
            }
          } else v.x = v.y = z0;
          L[hoff[tk] + (hr[tk] & hmask[tk])]       = v.x;
          L[hoff[tk] + ((hr[tk] + 1) & hmask[tk])] = v.y;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      g++;
      if (lane == 0) bx_cnt_store(hprog + w, BX_G * g);
    }
    // ---- flush group fg: the results of steps [8 fg, 8 fg + 8) leave for memory
    if (fg * BX_G < T) {
      const int need = (fg + 1) * BX_G < T ? (fg + 1) * BX_G : T;
      if (bx_cnt_load(cprog + w) >= need) {
        did = true;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (plane_ok) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const int s = 16 * p + (lane >> 2), qd = lane & 3, jj = 64 * J - k + s;
            const int r = BX_G * fg - 2 - 2 * s - 4 * w + 2 * qd;
            if (r >= 0 && r < nx && jj >= 0 && jj < ny) {
              const int       xo = BX_OX + (w * 64 + s) * (BX_RX + 1);
              const long long e  = phys(r, jj, k);
              bx_double2      v;
              v.x = L[xo + (r & (BX_RX - 1))];
              v.y = L[xo + ((r + 1) & (BX_RX - 1))];
              if (REV) {
                const double tmp = v.x;
                v.x              = v.y;
                v.y              = tmp;
              }
              // forward sweep inside a symmetric application: only the lines other workgroups read go to memory (the last two lanes, the
              // chunk's top plane); the result of the application is the backward sweep's
              if (Q.xfull || s >= 62 || w == BX_P - 1) bx_store2_sc1(Q.xout + e, v);
              if (KIND == 0) {
                const int  to = BX_OT + (w * 64 + s) * (BX_RX + 1);
                bx_double2 tv;
                tv.x = L[to + (r & (BX_RX - 1))];
                tv.y = L[to + ((r + 1) & (BX_RX - 1))];
                *reinterpret_cast<bx_double2 *>(Q.tout + e) = tv;  // (forward: never REV)
              }
            }
          }
        }
        fg++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) bx_cnt_store(flushp + w, fg * BX_G < T ? fg * BX_G : T);
      }
    }
    if (!did) {
      __builtin_amdgcn_s_sleep(2);
      if (bx_cnt_load(abortw)) alive = false;
    }
  }
}

// expected presence of the 27 canonical positions at every row against the row's template: count of rows that differ
__global__ void box_verify_kernel(long long m, int nx, int ny, int nz, const unsigned char *__restrict__ tid, const unsigned int *__restrict__ tmask, unsigned int base_mask, unsigned int *bad)
{
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((long long)nx * ny));
    unsigned  want = 0;
    for (int p = 0; p < 27; p++) {
      const int di = p % 3 - 1, dj = (p / 3) % 3 - 1, dk = p / 9 - 1;
      if (i + di >= 0 && i + di < nx && j + dj >= 0 && j + dj < ny && k + dk >= 0 && k + dk < nz) want |= 1u << p;
    }
    want &= base_mask;
    if (tmask[tid[r]] != want) atomicAdd(bad, 1u);
  }
}

}  // namespace

struct hipxSorBox_s {
  int           nx = 0, ny = 0, nz = 0, nb = 0, nch = 0, T = 0, ngroups = 0, em = 0;
  long long     m = 0;
  double        coefF[13], coefB[13], diag = 0.0, z0 = 0.0;
  int2         *d_order = nullptr;
  unsigned int *d_ctl = nullptr;
};
typedef hipxSorBox_s *hipxSorBox;

extern "C" void hipxSorBoxFree_(void *p)
{
  hipxSorBox B = (hipxSorBox)p;
  if (!B) return;
  (void)hipFree(B->d_order);
  (void)hipFree(B->d_ctl);
  delete B;
}

// Looks at the row templates of a matrix (hipxMatTemplates_: per template the (column - row, value) list in CSR order and the position of the
// diagonal) and decides whether the plane march applies.  *out = NULL when it does not (no error).
extern "C" int hipxSorBoxBuild_(long long m, int ntmpl, const int *tstart, const int *toff, const double *tval, const int *tdiag, const int64_t *tcount, const unsigned char *d_tid, void **out)
{
  *out = nullptr;
  if (m < 64 || ntmpl < 1 || ntmpl > 256) return HIPX_SUCCESS;
  int best = 0;
  for (int t = 1; t < ntmpl; t++)
    if (tstart[t + 1] - tstart[t] > tstart[best + 1] - tstart[best] || (tstart[t + 1] - tstart[t] == tstart[best + 1] - tstart[best] && tcount[t] > tcount[best])) best = t;
  const int  blen = tstart[best + 1] - tstart[best];
  const int *boff = toff + tstart[best];
  if (blen < 3 || blen > 27 || tdiag[best] < 0) return HIPX_SUCCESS;
  // clusters of consecutive offsets -> centres: 0, L, {S - L, S, S + L} or {S}
  std::vector<long long> centres;
  for (int a = 0; a < blen;) {
    int e = a;
    while (e + 1 < blen && boff[e + 1] == boff[e] + 1) e++;
    const long long cc = ((long long)boff[a] + boff[e]) / 2;
    if (cc > 0) centres.push_back(cc);
    a = e + 1;
  }
  if (centres.empty()) return HIPX_SUCCESS;
  long long Lx = centres[0], S = 0;
  if (centres.size() == 1) S = m;  // one plane
  else if (centres.size() == 2) S = centres[1];
  else if (centres.size() == 4) S = centres[2];
  else return HIPX_SUCCESS;
  if (Lx < 4 || (Lx & 1) || S % Lx || m % S || S / Lx < 3) return HIPX_SUCCESS;
  const int nx = (int)Lx, ny = (int)(S / Lx), nz = (int)(m / S);
  auto pos_of = [&](long long off, int &p) -> bool {
    const long long dk = (long long)std::floor((double)off / (double)S + 0.5), rem = off - dk * S;
    const long long dj = (long long)std::floor((double)rem / (double)Lx + 0.5), di = rem - dj * Lx;
    if (dk < -1 || dk > 1 || dj < -1 || dj > 1 || di < -1 || di > 1) return false;
    if (nz == 1 && dk != 0) return false;
    p = (int)(9 * (dk + 1) + 3 * (dj + 1) + (di + 1));
    return true;
  };
  double   bval[27];
  unsigned bmask = 0;
  for (int a = 0; a < blen; a++) {
    int p;
    if (!pos_of(boff[a], p) || ((bmask >> p) & 1)) return HIPX_SUCCESS;
    bmask |= 1u << p;
    bval[p] = tval[tstart[best] + a];
  }
  if (!((bmask >> 13) & 1)) return HIPX_SUCCESS;
  // every template: a sub-list of the interior row's list with its values, the same diagonal
  std::vector<unsigned> tmask((size_t)ntmpl, 0u);
  for (int t = 0; t < ntmpl; t++) {
    if (tdiag[t] < 0) return HIPX_SUCCESS;
    for (int a = tstart[t]; a < tstart[t + 1]; a++) {
      int p;
      if (!pos_of(toff[a], p) || !((bmask >> p) & 1) || tval[a] != bval[p]) return HIPX_SUCCESS;
      tmask[(size_t)t] |= 1u << p;
    }
    if (!((tmask[(size_t)t] >> 13) & 1)) return HIPX_SUCCESS;
  }
  // dependency sides: structurally symmetric, couplings of one sign and nonzero
  const unsigned lowm = bmask & 0x1FFFu;
  unsigned       upm  = 0;
  for (int e = 0; e < 13; e++)
    if ((bmask >> (26 - e)) & 1) upm |= 1u << e;
  if (lowm != upm || !lowm) return HIPX_SUCCESS;
  if (lowm != 0x1FFFu && lowm != 0x1410u) return HIPX_SUCCESS;  // instantiated: the 27-point box and the 7-point star
  int sgn = 0;
  for (int p = 0; p < 27; p++)
    if (p != 13 && ((bmask >> p) & 1)) {
      if (bval[p] == 0.0 || bval[p] != bval[p]) return HIPX_SUCCESS;
      const int sg = bval[p] < 0 ? -1 : 1;
      if (sgn && sg != sgn) return HIPX_SUCCESS;
      sgn = sg;
    }
  hipxSorBox B = new hipxSorBox_s;
  B->nx = nx, B->ny = ny, B->nz = nz, B->m = m, B->em = (int)lowm;
  B->diag = bval[13];
  B->z0   = sgn < 0 ? -0.0 : 0.0;
  for (int e = 0; e < 13; e++) {
    B->coefF[e] = ((lowm >> e) & 1) ? bval[e] : 0.0;
    B->coefB[e] = ((lowm >> e) & 1) ? bval[26 - e] : 0.0;  // mirrored grid: logical position e is the physical position 26 - e
  }
  hipStream_t   st = rt().compute;
  unsigned int *d_tm = nullptr;
  auto          bail = [&](int ierr) {
    (void)hipFree(d_tm);
    hipxSorBoxFree_(B);
    return ierr;
  };
  if (hipMalloc((void **)&B->d_ctl, sizeof(unsigned int) * 4) != hipSuccess || hipMalloc((void **)&d_tm, sizeof(unsigned int) * (size_t)ntmpl) != hipSuccess) return bail(fail(HIPX_ERR_HIP_BASE, "hipMalloc", __FILE__, __LINE__));
  if (hipMemsetAsync(B->d_ctl, 0, sizeof(unsigned int) * 4, st) != hipSuccess || hipMemcpyAsync(d_tm, tmask.data(), sizeof(unsigned int) * (size_t)ntmpl, hipMemcpyHostToDevice, st) != hipSuccess)
    return bail(fail(HIPX_ERR_HIP_BASE, "hipMemcpy", __FILE__, __LINE__));
  box_verify_kernel<<<(unsigned)std::min<long long>((m + 255) / 256, 8192), 256, 0, st>>>(m, nx, ny, nz, d_tid, d_tm, bmask, B->d_ctl + 2);
  unsigned int nbad = 1;
  if (hipMemcpyAsync(&nbad, B->d_ctl + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return bail(fail(HIPX_ERR_HIP_BASE, "verify", __FILE__, __LINE__));
  (void)hipFree(d_tm);
  d_tm = nullptr;
  if (nbad) {  // some row is not "the interior row cut at the boundary": not a box in natural ordering
    hipxSorBoxFree_(B);
    return HIPX_SUCCESS;
  }
  B->nb      = (ny + nz - 2) / 64 + 1;
  B->nch     = (nz + BX_P - 1) / BX_P;
  B->T       = nx + 2 + 2 * 63 + 4 * (BX_P - 1);
  B->ngroups = (B->T + BX_G - 1) / BX_G;
  // ticket order: by estimated start (a chunk hop ~ 4 P steps + a hand-off, a block hop ~ 128 steps + a hand-off); every dependency of (J, c)
  // -- (J - 1, c), (J - 1, c - 1), (J, c - 1) -- starts earlier
  std::vector<int2> order;
  for (int c = 0; c < B->nch; c++)
    for (int J = 0; J < B->nb; J++) order.push_back(make_int2(J, c));
  std::stable_sort(order.begin(), order.end(), [](const int2 &a, const int2 &b) { return a.y * 5 + a.x * 16 < b.y * 5 + b.x * 16; });
  if (hipMalloc((void **)&B->d_order, sizeof(int2) * order.size()) != hipSuccess || hipMemcpy(B->d_order, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(HIPX_ERR_HIP_BASE, "hipMalloc", __FILE__, __LINE__));
  *out = B;
  return HIPX_SUCCESS;
}

extern "C" void hipxSorBoxShape_(void *p, int *nx, int *ny, int *nz)
{
  hipxSorBox B = (hipxSorBox)p;
  *nx = B->nx, *ny = B->ny, *nz = B->nz;
}

template <bool REV, int EM, int KIND>
static int box_launch(hipxSorBox B, const BoxParams &Q)
{
  static bool attr = false;
  auto        kern = &sor_box_kernel<REV, EM, KIND>;
  if (!attr) {
    HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BX_LDS_BYTES));
    attr = true;
  }
  kern<<<(unsigned)(B->nb * B->nch), BX_THREADS, BX_LDS_BYTES, rt().compute>>>(Q);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// One zero-guess sweep.  kind 0: forward (rhs = b; t and x written; xfull = 0 inside a symmetric application: only the lines the schedule itself
// hands between workgroups reach xout); kind 1: backward after forward (rhs = t); kind 2: backward alone (rhs = b).  xout must be filled with the
// sentinel (sor_fill_kernel) before the launch: a row of x is its own ready flag between workgroups.  Vectors 16-byte aligned.
extern "C" int hipxSorBoxRun_(void *p, int kind, const double *rhs, double *tout, double *xout, double omega, double shift, int xfull)
{
  hipxSorBox B = (hipxSorBox)p;
  BoxParams  Q;
  Q.nx = B->nx, Q.ny = B->ny, Q.nz = B->nz, Q.nb = B->nb, Q.nch = B->nch, Q.T = B->T, Q.ngroups = B->ngroups, Q.xfull = xfull;
  Q.m = B->m;
  const bool plain = (omega == 1.0 && shift <= 0.0);
  Q.idiag = plain ? 1.0 / B->diag : omega / (shift + B->diag);  // MatInvertDiagonalForSOR_SeqAIJ aij.c:1797-1840
  Q.z0    = B->z0;
  Q.omw   = 1.0 - omega;
  Q.rhs = rhs, Q.xout = xout, Q.tout = tout, Q.order = B->d_order, Q.ctl = B->d_ctl;
  memcpy(Q.coef, kind == 0 ? B->coefF : B->coefB, sizeof(Q.coef));
  HIPX_HIP(hipMemsetAsync(B->d_ctl, 0, sizeof(unsigned int), rt().compute));  // the ticket; the error word is sticky until read
  const bool box27 = B->em == 0x1FFF;
  if (kind == 0) return box27 ? box_launch<false, 0x1FFF, 0>(B, Q) : box_launch<false, 0x1410, 0>(B, Q);
  if (kind == 1) return box27 ? box_launch<true, 0x1FFF, 1>(B, Q) : box_launch<true, 0x1410, 1>(B, Q);
  return box27 ? box_launch<true, 0x1FFF, 2>(B, Q) : box_launch<true, 0x1410, 2>(B, Q);
}

extern "C" int hipxSorBoxError_(void *p, unsigned int *err)
{
  hipxSorBox B = (hipxSorBox)p;
  HIPX_HIP(hipMemcpyAsync(err, B->d_ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  if (*err) HIPX_HIP(hipMemsetAsync(B->d_ctl, 0, 2 * sizeof(unsigned int), rt().compute));
  return HIPX_SUCCESS;
}
