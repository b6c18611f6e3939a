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
//     16-byte accesses, transposed through LDS rings -- the compute waves never touch memory;
//   * (round 6) the hop between CHUNKS goes through a mailbox: the top plane of a chunk leaves pair of rows by pair of rows, the 64 lanes' pairs of
//     one flush side by side (1 KB), and a poller wave of the chunk above stages whole flushes as the chunk's "wave -1"; whoever reads an entry
//     puts the sentinel back, so the mailbox is filled once per matrix.  Hop = 16 steps of skew + 3.7 us (round 5: groups of eight rows through x
//     itself, 16 + 10 steps + 3.4 us);
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

constexpr int BX_P = 4;                                       // planes per workgroup (rows per staging / flushing group: the kernel's template parameter G)
constexpr int BX_RX = 16, BX_RB = 32, BX_RS = 32, BX_RW = 32;  // ring rows: own x lines / rhs / south plane / west lines
constexpr int BX_THREADS = 4 * BX_P * 64;                      // P compute waves + 2 P stagers (even / odd groups) + 2 flushers of planes 0 .. P - 2 + the top plane's flusher + the south plane's poller
// LDS layout (doubles)
constexpr int BX_OX = 0;                                        // X [P][64][RX + 1]
constexpr int BX_OS = BX_OX + BX_P * 64 * (BX_RX + 1);          // S [66][RS + 1]
constexpr int BX_OW = BX_OS + 66 * (BX_RS + 1);                 // W [P][2][RW]
constexpr int BX_OB = BX_OW + BX_P * 2 * BX_RW;                 // B [P][64][RB + 1]
constexpr int BX_OT = BX_OB + BX_P * 64 * (BX_RB + 1);          // TR[P][64][RX + 1]   (forward sweeps only)
constexpr int BX_OC = BX_OT + BX_P * 64 * (BX_RX + 1);          // counters (ints): (P + 2) records of 4, abort, ticket
constexpr int BX_LDS_BYTES = BX_OC * 8 + 2 * (8 * BX_P + 8) + 16;
static_assert(BX_RW == BX_RS, "the stager writes both halo rings with one row mask");
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
  double       *mbox;           // the chunks' top planes on their way up: [chunk c < nch - 1][block J][flush f < T / 2 + 8][lane] pairs of rows, sentinel = not there yet
  int          dbg;             // HIPX_SORBOX_DEBUG (timing probes, WRONG RESULTS): 1 = no staging / flushing (the compute waves alone), 2 = also no waiting for the lower plane;
                                // bits (helpers on): 4 = no south poller (chunks run without their hop; one-block boxes only: nobody mails), 32 = the flushers store nothing
  unsigned long long *trace;    // HIPX_SORBOX_STATS=3: [2 c][f] when workgroup (0, c) mailed flush f, [2 c + 1][f] when workgroup (0, c + 1) had it in its ring (100 MHz clock), c < 8
  unsigned long long *stats;    // HIPX_SORBOX_STATS: spin counts of the compute waves by unmet condition [4], stager ring waits [1], stager halo polls [1], steps [1]
};

typedef __attribute__((address_space(3))) double bx_lds_double;
typedef __attribute__((address_space(3))) int    bx_lds_int;
typedef double bx_double2 __attribute__((ext_vector_type(2)));

// (readfirstlane: a counter is the same for every lane -- as a VGPR value the compiler takes every branch on it for divergent and wraps the waits in
// exec-mask bookkeeping: measured ~680 clocks per readiness check that never had to wait)
__device__ __forceinline__ int bx_cnt_load(bx_lds_int *p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load((int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void bx_cnt_store(bx_lds_int *p, int v) { __hip_atomic_store((int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Ordering between the waves of ONE workgroup goes through LDS only: a wave's LDS operations complete in order, so "ring writes before the counter"
// is s_waitcnt lgkmcnt(0) and "counter before ring reads" is program order (the counter's value has been waited for when it is compared).  The
// __builtin_amdgcn_fence forms also wait for vmcnt(0) -- the stager would sit out the full latency of the loads it has just issued for the NEXT
// groups at every publish (measured: 424 ns per step in a workgroup without any dependency, the memory latency of one group divided by eight).
__device__ __forceinline__ void bx_lds_release() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void bx_lds_acquire() { asm volatile("" ::: "memory"); }

// wait until *p >= want (uniform over the wave); false when the launch was aborted
template <int SLEEP = 4>
__device__ __forceinline__ bool bx_wait_ge(bx_lds_int *p, int want, bx_lds_int *abortw, unsigned int *gerr)
{
  int       spins = 0;
  long long t0    = 0;
  while (bx_cnt_load(p) < want) {
    __builtin_amdgcn_s_sleep(SLEEP);  // (the helpers' waits: groups of eight steps, nothing is lost by looking every ~256 clocks)
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

__device__ __forceinline__ void bx_store2_sc1(double *p, bx_double2 v)
{
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");  // write-through: every 8-byte half is its own ready flag
}

// REV: the backward sweep = the forward schedule on the mirrored grid (logical row r <-> physical row m - 1 - r), entry order reversed.
// EM: bit e set = the interior row has the dependency-side entry at canonical position e.  KIND as in hipx_sor.hip: 0 forward zero-guess
// (t = sum, x = sum idiag), 1 backward after forward (x = (1 - w) (t idiag) + sum idiag: aij.c:1955 with the forward result x = t idiag
// re-formed from t), 2 backward zero-guess alone.
// Waves of a workgroup: w = 0..P-1 compute (plane k0 + w), then two stagers per plane (even / odd groups: right-hand side + west lines into the
// LDS rings), then two flushers for planes 0 .. P - 2 (results out of the rings to memory), the flusher of the top plane (pair by pair: x, t and
// the mailbox) and the poller of the south plane.  Counters in LDS are the only synchronisation inside the workgroup; between workgroups a row of
// x (a mailbox entry) in memory is its own ready flag.
//
// What a step costs is INSTRUCTIONS: a lone wave issues one instruction every 4-5 clocks whatever its kind, and the dependent arithmetic of a
// 27-point row (13 products and differences + the scaling) is 88 clocks by itself (scripts/diag/lat_probe.hip).  The first versions of this loop
// took ~240 instructions per step (register rotations as moves, four counter reads, selects for the rows beyond a line's ends, exec-mask
// bookkeeping around a divergent branch): 520 ns per step free-running.  Hence:
//   * the loop is unrolled by four: the three-row windows of the four neighbour lines are circular over FOUR registers (two more rows are in
//     flight: operands are requested two steps ahead), the slot of a row is (step & 3) -- no moves;
//   * a line has a row nx holding the zero element (written by its lane one step after its last row, staged for the west / south lines) and the
//     rings start out filled with the zero element: rows -1 and nx need no select;
//   * every lane does the arithmetic of every step; only the ring writes are predicated;
//   * a wave's four counters are 16-bit fields of ONE 8-byte record: one LDS read per step, requested before the arithmetic, looked at after it.
typedef __attribute__((address_space(3))) unsigned short     bx_lds_u16;
typedef __attribute__((address_space(3))) unsigned long long bx_lds_u64;
__device__ __forceinline__ void bx_put16(bx_lds_u16 *p, int v) { __hip_atomic_store((unsigned short *)p, (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int  bx_get16(bx_lds_u16 *p) { return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load((unsigned short *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
// wait until the 16-bit counter *p >= want (uniform over the wave); false when the launch was aborted.  Helper waves: groups of eight steps,
// nothing is lost by looking every ~256 clocks
__device__ __forceinline__ bool bx_wait16(bx_lds_u16 *p, int want, bx_lds_int *abortw, unsigned int *gerr)
{
  int       spins = 0;
  long long t0    = 0;
  while (bx_get16(p) < want) {
    __builtin_amdgcn_s_sleep(4);
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

// G: steps per staging / flushing group (8, 4 or 2).  A row leaves for memory when its group of G steps is complete and the workgroup above stages
// it with ITS group: every hop between workgroups waits out up to G + (14 mod G) steps beyond the recurrence's own skew.
template <bool REV, int EM, int KIND, int G>
__global__ __launch_bounds__(BX_THREADS, 1) void sor_box_kernel(const BoxParams Q)
{
  constexpr int BX_G = G;
  constexpr int PPL = G / 2, LPP = 64 / PPL, NPASS = 64 / LPP;  // lanes (pairs of rows) per line and group; lines per pass; passes over the 64 lines
  static_assert(G == 8, "groups of eight steps (the flushers count on two groups per ring)");
  extern __shared__ double bx_smem[];
  bx_lds_double *L = (bx_lds_double *)bx_smem;
  // Counters: 16-bit "steps done", one 8-byte record per READER so that a wave fetches everything it waits for with one LDS access:
  //   C[w] (compute wave w)         = {relaxed by plane w - 1 (w = 0: virtual steps of the south plane staged), staged for plane w, relaxed by plane w + 1, flushed of plane w}
  //   H[w] (stagers / flusher of w) = {relaxed by plane w, relaxed by plane w + 1, staged for plane w, w = P - 1: steps whose rows the mailbox wave has read}
  // A writer stores its counter into every record that holds it (one ds_write_b16, one lane per copy).  Planes that do not exist read 0xffff.
  bx_lds_u16 *c16 = (bx_lds_u16 *)(L + BX_OC);
  bx_lds_int *abortw = (bx_lds_int *)(c16 + 8 * BX_P + 8), *tick = abortw + 1;
  bx_lds_u16 *dummy = c16 + 8 * BX_P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto Crec = [&](int w_) -> bx_lds_u16 * { return c16 + 8 * w_; };
  auto Hrec = [&](int w_) -> bx_lds_u16 * { return c16 + 8 * w_ + 4; };
  const double z0 = Q.z0;
  for (int q = threadIdx.x; q < BX_OB; q += BX_THREADS) L[q] = z0;  // the x, south and west rings start out as the zero element (rows -1 of every line)
  if (threadIdx.x == 0) {
    for (int q = 0; q < 8 * BX_P + 8; q++) bx_put16(c16 + q, 0);
    if (Q.dbg & 7) bx_put16(Crec(0) + 0, 0xffff);  // (otherwise: the south plane's virtual steps, published by its poller)
    bx_put16(Crec(BX_P - 1) + 2, 0xffff);
    bx_put16(Hrec(BX_P - 1) + 1, 0xffff);
    bx_cnt_store(abortw, 0);
    bx_cnt_store(tick, (int)atomicAdd(Q.ctl, 1u));  // workgroups take their (block, chunk) in ticket order: every dependency has an earlier ticket
  }
  __syncthreads();
  const int2 jc = Q.order[bx_cnt_load(tick)];
  const int  J = jc.x, c = jc.y, k0 = BX_P * c;
  if (Q.stats && threadIdx.x == 0) Q.stats[8 + 2 * (c * Q.nb + J)] = wall_clock64();
  const int  nx = Q.nx, ny = Q.ny, nz = Q.nz, T = Q.T;  // T: a multiple of four
  unsigned int *gerr = Q.ctl + 1;

  if (wave < BX_P) {
    // ------------------------------------------------------------------------------------------------ compute wave w: plane k0 + w
    const int  w = wave, s = lane, k = k0 + w;
    const int  j = 64 * J - k + s;
    const bool valid = j >= 0 && j < ny && k < nz;
    const int  i0 = -2 - 2 * s - 4 * w;  // row of this lane in step t: t + i0 (-1: its neighbours' rows 0 arrive; nx: it publishes the line's zero element)
    // byte offsets of the rings this lane reads (the row slot is added per step) and writes
    int nb_base[3], nb_mask[3];
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
    const int wo = BX_OW + (w * 2 + 1) * BX_RW;
    const int xo = BX_OX + (w * 64 + s) * (BX_RX + 1), bo = BX_OB + (w * 64 + s) * (BX_RB + 1), to = BX_OT + (w * 64 + s) * (BX_RX + 1);
    bx_lds_u64 *crec = (bx_lds_u64 *)Crec(w);
    // where this wave's "steps relaxed" goes: C[w + 1][0], C[w - 1][2], H[w][0], H[w - 1][1] -- lanes 0-3 store one copy each
    bx_lds_u16 *pub = dummy + (lane & 7);
    if (lane == 0 && w + 1 < BX_P) pub = Crec(w + 1) + 0;
    if (lane == 1 && w > 0) pub = Crec(w - 1) + 2;
    if (lane == 2) pub = Hrec(w) + 0;
    if (lane == 3 && w > 0) pub = Hrec(w - 1) + 1;
    // what step t needs, checked before its operands are requested:
    //   lower plane: its row i + 1 of line j + 1 was relaxed in wave w - 1's step t - 3            -> relaxed(w - 1) >= t - 2   (w = 0: the south plane's
    //                virtual step t - 3 is staged -> its poller's count >= t)
    //   right-hand side / west / south rows of the step are staged                                  -> staged(w)      >= t + 1
    //   x ring (16 rows): plane k + 1 reads a row up to 7 steps after it was written                -> relaxed(w + 1) >= t - 8
    //   x / t rings: the rows this step overwrites have left for memory                             -> flushed(w)     >= t - 15
    unsigned long long raw = 0;
    int                c_low = 0, c_stg = 0, c_up = 0, c_fl = 0;
    auto look_issue = [&]() { raw = __hip_atomic_load((unsigned long long *)crec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto look_take = [&]() {
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)raw), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(raw >> 32));
      c_low = (int)(lo & 0xffffu), c_stg = (int)(lo >> 16), c_up = (int)(hi & 0xffffu), c_fl = (int)(hi >> 16);
    };
    const int staged_all = Q.ngroups * BX_G;
    // (readfirstlane: w comes from threadIdx -- as a VGPR value it makes every branch on ready() divergent for the compiler: 850 -> 995 instructions
    // in the loop of four steps, 304 -> 330 ns per step)
    const int low_off = __builtin_amdgcn_readfirstlane(w == 0 ? 2 : 0);  // the south plane's poller counts from virtual step -2 (rows 0, 1 of its first line)
    auto ready = [&](int t) -> bool {
      if (Q.dbg == 2) return true;
      const bool core = (c_low >= (t - 2 < T ? t - 2 : T) + low_off) & (c_up >= t - 8);
      if (Q.dbg == 1) return core;
      return core & (c_stg >= (t + 1 < staged_all ? t + 1 : staged_all)) & (c_fl >= t - (BX_RX - 1));
    };
    unsigned sp_low = 0, sp_stg = 0, sp_up = 0, sp_fl = 0;  // spins by the first unmet condition (HIPX_SORBOX_STATS)
    auto wait_ready = [&](int t) -> bool {  // (the record has just been looked at)
      int       spins = 0;
      long long t0    = 0;
      while (!ready(t)) {
        if (Q.stats) {
          if (c_low < (t - 2 < T ? t - 2 : T) + low_off) sp_low++;
          else if (c_stg < (t + 1 < staged_all ? t + 1 : staged_all)) sp_stg++;
          else if (c_up < t - 8) sp_up++;
          else sp_fl++;
        }
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
        look_issue();
        look_take();
      }
      return true;
    };
    // registers of the recurrence: Wn[line][slot], line 0-2 = the lower plane's lines j - 1, j, j + 1, line 3 = the previous line of this plane;
    // in step t slot (t - 1) & 3 holds row i - 1, t & 3 row i, (t + 1) & 3 row i + 1 and (t + 2) & 3 the row arriving for step t + 1
    double Wn[4][4], Rr[2], Ww[2], xcur = z0;
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
      for (int q = 0; q < 4; q++) Wn[l][q] = z0;
    Rr[0] = Rr[1] = Ww[0] = Ww[1] = z0;
    bool alive = true;
    // the operands of step u: rows i + 1 of the lower plane's lines into window slot (u + 1) & 3, the right-hand side of row i and the west line's
    // row i + 1 into slot u & 1
#define BX_REQUEST(U, S4, S2) \
  { \
    const int r1_ = (U) + i0 + 1; \
    Wn[0][S4] = L[nb_base[0] + (r1_ & nb_mask[0])]; \
    Wn[1][S4] = L[nb_base[1] + (r1_ & nb_mask[1])]; \
    Wn[2][S4] = L[nb_base[2] + (r1_ & nb_mask[2])]; \
    Rr[S2]    = L[bo + ((r1_ - 1) & (BX_RB - 1))]; \
    Ww[S2]    = L[wo + (r1_ & (BX_RW - 1))]; \
  }
    // one step (PH = t & 3): the record is requested; row i is relaxed out of registers; the results go to the rings and the step is published with
    // NO wait in between (a wave's LDS operations are executed in order: whoever sees the counter sees the rows); the record decides whether step
    // t + 2 may be prepared; its operands are requested
#define BX_STEP(PH, GUARD) \
  { \
    const int t_ = t + (PH), i = t_ + i0; \
    constexpr int M1 = ((PH) + 3) & 3, C0 = (PH), P1 = ((PH) + 1) & 3, S2 = (PH) & 1; \
    if ((GUARD) != 1 || t_ + 2 < T) look_issue(); \
    { /* the previous line's row i + 1: lane s - 1 relaxed it in the step before (wavefront shift); lane 0 takes the west line's */ \
      const int nlo = __builtin_amdgcn_update_dpp(__double2loint(Ww[S2]), __double2loint(xcur), 0x138 /* wave_shr:1 */, 0xf, 0xf, false); \
      const int nhi = __builtin_amdgcn_update_dpp(__double2hiint(Ww[S2]), __double2hiint(xcur), 0x138, 0xf, 0xf, false); \
      Wn[3][P1]     = __hiloint2double(nhi, nlo); \
    } \
    const double rhs = Rr[S2]; \
    double       sum = rhs; \
    _Pragma("unroll") for (int q = 0; q < 13; q++) \
    { \
      const int e = REV ? 12 - q : q; \
      if ((EM >> e) & 1) { \
        const int    l_ = e < 12 ? e / 3 : 0, cc = e % 3; \
        const double v_ = e == 12 ? xcur : (cc == 0 ? Wn[l_][M1] : (cc == 1 ? Wn[l_][C0] : Wn[l_][P1])); \
        sum             = sum - Q.coef[e] * v_; \
      } \
    } \
    double xv; \
    if (KIND == 1) xv = Q.omw * (rhs * Q.idiag) + sum * Q.idiag; \
    else xv = sum * Q.idiag; \
    xcur = (valid && (unsigned)i < (unsigned)nx) ? xv : z0; /* beyond the line, and a lane without a line: the zero element */ \
    bool rdy_ = true; \
    if ((GUARD) >= 2) { /* the four comparisons as ONE 64-bit subtraction (fields below 0x8000, no borrow across fields) */ \
      const unsigned long long R_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(raw >> 32)) << 32) | \
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)raw); \
      rdy_ = ((((R_ | 0x8000800080008000ull) - needN) & 0x8000800080008000ull) == 0x8000800080008000ull); \
      if ((GUARD) == 2) needN += 0x0001000100010001ull; \
      else needN += 0x0000000000010001ull + ((unsigned long long)(t_ >= 6) << 32) + ((unsigned long long)(t_ >= 13) << 48); /* (needs below zero stay 0) */ \
    } else if ((GUARD) != 1 || t_ + 2 < T) { /* the record is looked at BEFORE the ring writes are issued: behind them the wait for it is a wait for them too */ \
      look_take(); \
      rdy_ = ready(t_ + 2); \
    } \
    if ((unsigned)i <= (unsigned)nx) L[xo + (i & (BX_RX - 1))] = xcur; \
    if (KIND == 0 && (unsigned)i < (unsigned)nx) L[to + (i & (BX_RX - 1))] = sum; \
    asm volatile("" ::: "memory"); /* ring writes, then the counter: program order = LDS order */ \
    if (lane < 4) bx_put16(pub, t_ + 1); \
    if ((GUARD) != 1 || t_ + 2 < T) { \
      if (!rdy_) { \
        look_issue(); \
        look_take(); \
        alive = wait_ready(t_ + 2); \
        if (!alive) break; \
      } \
      bx_lds_acquire(); /* (asking for the operands before the ring writes -- they come out of other waves' rings -- was measured: 279 -> 297 ns per step) */ \
      BX_REQUEST(t_ + 2, M1, S2) \
    } \
  }
    look_issue();
    look_take();
    alive = wait_ready(1);
    if (alive) {
      bx_lds_acquire();
      BX_REQUEST(0, 1, 0)
      BX_REQUEST(1, 2, 1)
    }
    // The four readiness comparisons of a step as one 64-bit subtraction: need = {t - 2 (+ 2: plane 0), t + 1, t - 8, t - 15} for the step t = t_ + 2
    // being prepared, in the record's field order, needs below zero held at 0.  Three loops: the first sixteen steps (the needs below zero: the
    // increment is put together per step), the steps in between (+ 1 per field and step), the last ones with every check spelled out (the ends of the
    // lines).  (The first sixteen steps spelled out cost ~80 ns each: 1.3 us at the start of every workgroup, that is on every hop between chunks --
    // a chunk never catches up with the chunk below, it runs at the same pace.)
    const bool swar = Q.dbg == 0 && staged_all < 0x8000;
    unsigned long long needN = (unsigned long long)(unsigned)low_off | (3ull << 16);
    int                t = 0;
    if (swar) {
      for (; t < 16 && alive; t += 4) {  // (T >= 148: every one of these steps prepares a step t + 2 < T)
        BX_STEP(0, 3)
        BX_STEP(1, 3)
        BX_STEP(2, 3)
        BX_STEP(3, 3)
      }
      for (; t + 8 <= T && alive; t += 4) {
        BX_STEP(0, 2)
        BX_STEP(1, 2)
        BX_STEP(2, 2)
        BX_STEP(3, 2)
      }
    }
    for (; t < T && alive; t += 4) {
      BX_STEP(0, 1)
      BX_STEP(1, 1)
      BX_STEP(2, 1)
      BX_STEP(3, 1)
    }
#undef BX_STEP
#undef BX_REQUEST
    if (Q.stats && lane == 0) {
      if (w == BX_P - 1) Q.stats[8 + 2 * (c * Q.nb + J) + 1] = wall_clock64();
      atomicAdd(Q.stats + 0, (unsigned long long)sp_low);
      atomicAdd(Q.stats + 1, (unsigned long long)sp_stg);
      atomicAdd(Q.stats + 2, (unsigned long long)sp_up);
      atomicAdd(Q.stats + 3, (unsigned long long)sp_fl);
      atomicAdd(Q.stats + 6, (unsigned long long)T);
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------------------------- helper waves
  // P .. 3 P - 1: two stagers per plane (even / odd groups of G steps: right-hand side + west lines into the LDS rings); 3 P: the flusher of planes
  // 0 .. P - 2 (groups of G steps); 3 P + 1: the flusher of the chunk's TOP plane, pair by pair; 3 P + 2: the poller of the SOUTH plane, pair by pair.
  // The last two are the hop between chunks: a row pair of plane k0 + P - 1 leaves for memory two steps after its first row was relaxed and is in
  // the ring of the chunk above one memory round trip later (round 5 / 6a: groups of eight steps on both sides, misaligned by two, the south plane
  // spread over the stagers of all planes and polled task after task: 6.6 us per hop beyond the recurrence's own sixteen steps).
  if (Q.dbg & 3) return;
  const int       hw  = wave - BX_P;
  const long long nxl = nx, nyl = ny;
  auto phys = [&](int r, int jj, int kk) -> long long {  // element index of logical row (r, jj, kk); pairs (r, r + 1), r even, are 16-byte aligned
    const long long lr = (long long)r + nxl * ((long long)jj + nyl * (long long)kk);
    return REV ? Q.m - 2 - lr : lr;  // REV: the pair (r, r + 1) lies at m - 2 - lr, halves swapped
  };
  // agent-scope (sc1) loads the compiler can see: the values are waited for where they are used
  auto ld = [&](const double *q) -> bx_double2 {
    const unsigned long long *u = reinterpret_cast<const unsigned long long *>(q);
    bx_double2                v;
    v.x = __longlong_as_double((long long)__hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    v.y = __longlong_as_double((long long)__hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return v;
  };
  auto unset = [](const bx_double2 &v) -> bool {
    return (unsigned long long)__double_as_longlong(v.x) == BX_SENTINEL || (unsigned long long)__double_as_longlong(v.y) == BX_SENTINEL;
  };
  // a helper's poll went on for too long (or another workgroup failed): the launch is given up
  auto give_up = [&](int &spins, long long &t0) -> bool {
    if ((++spins & 0xf) == 0 && bx_cnt_load(abortw)) return true;
    if ((spins & 0xff) == 0) {
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      if (__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > BX_SPIN_TICKS) {
        __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bx_cnt_store(abortw, 1);
        return true;
      }
    }
    return false;
  };

  if (hw >= 2 * BX_P) {
    if (hw == 2 * BX_P || hw == 2 * BX_P + 3) {
      // ------------------------------------------------------------------------------------------ flushers of planes k0 .. k0 + P - 1
      // Two waves for the P planes' results: x and t to memory in groups of G steps, 64-byte pieces.  The wave shares its SIMD with a compute wave
      // and gets an issue slot every 5-8 clocks: a group of one plane may cost ~70 instructions, not 400 -- ring offsets (two phases: a group is
      // half the 16-row ring), row numbers and addresses are formed once and advanced by a group per pass.  (With the index arithmetic of the first
      // version a wave with two planes took longer than the eight steps it has, and its planes -- then everybody -- ran at 360-460 ns per step
      // instead of 305.)
      static_assert(BX_G == 8 && BX_RX == 16, "two phases of the ring per group");
      constexpr int NPL = (BX_P + 1) / 2;  // planes of the first wave (the second has the rest)
      const int     w_lo = hw == 2 * BX_P ? 0 : NPL, w_hi = hw == 2 * BX_P ? NPL : BX_P;
      bool          lok[NPL][NPASS], xst[NPL][NPASS], wt[NPL][NPASS];
      int           rr[NPL][NPASS], la[NPL][NPASS], lb[NPL][NPASS];
      double       *xp[NPL][NPASS], *tp[NPL][NPASS];
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
#pragma unroll
        for (int p = 0; p < NPASS; p++) {
          const int w = w_lo + pl, s = LPP * p + lane / PPL, qd = lane % PPL, k = k0 + w, jj = 64 * J - k + s;
          lok[pl][p] = w < w_hi && k < nz && jj >= 0 && jj < ny;
          xst[pl][p] = Q.xfull || s >= 62;  // forward sweep inside a symmetric application: only the lines other workgroups read in x itself go to memory
                                            // (the chunk's top plane travels through the mailbox)
          wt[pl][p]  = s >= 62;  // (the lines other workgroups poll in x itself -- block J + 1's west lines: write-through, a row is its own ready flag;
                                 //  everything else only has to be in memory when the kernel ends)
          rr[pl][p]  = -2 - 2 * s - 4 * w + 2 * qd;  // first row of the pair in group 0 (even: the pair does not wrap in the ring)
          la[pl][p]  = (w * 64 + s) * (BX_RX + 1) + (rr[pl][p] & (BX_RX - 1));
          lb[pl][p]  = (w * 64 + s) * (BX_RX + 1) + ((rr[pl][p] + BX_G) & (BX_RX - 1));
          const long long e0 = lok[pl][p] ? phys(0, jj, k) : 0;
          xp[pl][p]  = Q.xout + (REV ? e0 - rr[pl][p] : e0 + rr[pl][p]);
          tp[pl][p]  = KIND == 0 ? Q.tout + e0 + rr[pl][p] : nullptr;
        }
      for (int fg = 0; fg * BX_G < T; fg++) {
        const int need = (fg + 1) * BX_G < T ? (fg + 1) * BX_G : T;
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
          const int w = w_lo + pl;
          if (w >= w_hi) break;
          if (!bx_wait16(Hrec(w), need, abortw, gerr)) return;
          bx_lds_acquire();
          bx_double2 xv[NPASS], tv[NPASS];
#pragma unroll
          for (int p = 0; p < NPASS; p++) {
            const int off = (fg & 1) ? lb[pl][p] : la[pl][p];
            xv[p].x = L[BX_OX + off];
            xv[p].y = L[BX_OX + off + 1];
            if (KIND == 0) {
              tv[p].x = L[BX_OT + off];
              tv[p].y = L[BX_OT + off + 1];
            }
          }
#pragma unroll
          for (int p = 0; p < NPASS; p++) {
            if (lok[pl][p] && (unsigned)rr[pl][p] < (unsigned)nx && !(Q.dbg & 32)) {
              if (xst[pl][p]) {
                bx_double2 v = xv[p];
                if (REV) v.x = xv[p].y, v.y = xv[p].x;
                if (wt[pl][p]) bx_store2_sc1(xp[pl][p], v);
                else *reinterpret_cast<bx_double2 *>(xp[pl][p]) = v;
              }
              if (KIND == 0) *reinterpret_cast<bx_double2 *>(tp[pl][p]) = tv[p];  // (forward: never REV)
            }
            rr[pl][p] += BX_G;
            xp[pl][p] += REV ? -BX_G : BX_G;
            if (KIND == 0) tp[pl][p] += BX_G;
          }
          bx_lds_release();  // (the ring reads are done; the stores may still be on their way)
          if (w == BX_P - 1 && !bx_wait16(Hrec(w) + 3, need, abortw, gerr)) return;  // (the top plane's rows are also read by the mailbox wave)
          if (lane == 0) bx_put16(Crec(w) + 3, need);
        }
      }
    } else if (hw == 2 * BX_P + 1) {
      // ------------------------------------------------------------------------------------------ flusher of the top plane k0 + P - 1
      // The chunk's top plane on its way up, pair by pair: the 64 lanes' rows of the last two steps into the MAILBOX of the chunk above (flush f = 1 KB
      // in one store instruction; through x itself the pairs of a flush lie nx rows apart: 64 partial lines for the poller above to collect).  Every
      // lane writes its entry of every flush from 7 on, the zero element where it has no row (and eight flushes of zero elements past the end of the
      // lines): the poller checks and stages whole flushes without a per-lane notion of what it needs.  Nothing else goes through this wave's memory
      // queue: with the plane's x and t in it (pair by pair: 64 partial lines per store instruction in front of the mailbox store, 0.3 ms per
      // application wherever the vectors lay unfavourably; group by group: every fourth flush late by the group's round trips) the hop grew by 3-8 us.
      constexpr int w = BX_P - 1;
      const int     k = k0 + w;
      const bool    lok = k < nz && 64 * J - k + lane >= 0 && 64 * J - k + lane < ny;
      // H[P - 1][3] = the steps whose rows this wave has read out of the x ring: the plane's flusher lets go of a group only when it is past it
      if (!(c + 1 < Q.nch && __any(lok))) {  // (no chunk above, or a plane without a line in the grid: nothing is sent, nothing is waited for)
        if (lane == 0) bx_put16(Hrec(w) + 3, 0xffff);
        return;
      }
      if (lane == 0) bx_put16(Hrec(w) + 3, 14);  // (the flushes before the seventh hold no row of the grid)
      double       *mb = Q.mbox + ((size_t)c * Q.nb + J) * (size_t)(T / 2 + 8) * 128 + 2 * lane;
      const int     xo = BX_OX + (w * 64 + lane) * (BX_RX + 1);
      bx_lds_u16   *rel = Hrec(w);
      for (int need = 16; need <= T + 16; need += 2) {  // (T: a multiple of four)
        int       spins = 0;
        long long t0    = 0;
        const int upto  = need < T ? need : T;
        while (bx_get16(rel) < upto) {
          __builtin_amdgcn_s_sleep(1);
          if (give_up(spins, t0)) return;
        }
        bx_lds_acquire();
        const int  r = need - 4 - 2 * lane - 4 * w;  // the rows lane s relaxed in steps need - 2, need - 1: still in the ring (16 rows) -- the plane's
        bx_double2 v;                               // flusher lets go of a group of eight only eight steps after its last row
        v.x = v.y = z0;
        if (lok && r >= 0 && r < nx) {
          v.x = L[xo + (r & (BX_RX - 1))];
          v.y = L[xo + ((r + 1) & (BX_RX - 1))];
        }
        bx_store2_sc1(mb + (size_t)(need / 2 - 1) * 128, v);  // (logical order: row r first)
        if (Q.trace && J == 0 && c < 8 && lane == 0 && need / 2 - 1 < 4096) Q.trace[(2 * c) * 4096 + need / 2 - 1] = wall_clock64();
        bx_lds_release();
        if (lane == 0) bx_put16(Hrec(w) + 3, upto);
      }
    } else if (hw == 2 * BX_P + 2) {
      // ------------------------------------------------------------------------------------------ poller of the south plane k0 - 1
      // The south plane is "wave -1" of the chunk: its VIRTUAL step v holds row v + 2 - 2 s of line s = 0 .. 63 of block J there (lane <-> line; the
      // two lines of block J - 1 in front of them are early and come with plane 0's stagers), and plane 0 waits for it exactly as plane w waits for
      // plane w - 1: staged virtual steps >= t - 2.  Pair n = virtual steps 2 n - 2, 2 n - 1 = rows 2 n - 2 s, + 1.
      bx_lds_u16 *sc = Crec(0) + 0;
      if (Q.dbg & 4) return;
      if (k0 == 0) {  // no plane below: the ring keeps the zero element
        if (lane == 0) bx_put16(sc, 0xffff);
        return;
      }
      // Pair n of line s = rows 2 n - 2 s, + 1 = what lane s of the chunk below relaxed in its steps 2 n + 14, 2 n + 15 = entry s of its flush n + 7 in
      // the mailbox.  A pass asks for the next FOUR flushes (4 x 1 KB, one 16-byte load per lane and flush), stages the leading ones that are
      // complete and puts the sentinel back into what it has read; it keeps no state from pass to pass.  (A window of eight pairs with per-slot
      // state compiled to ~950 instructions per pass: with a compute wave on the same SIMD a pass took longer than the sixteen steps of lead the
      // ring allows, and the chunk ran at the poller's pace, 1.5 x slower than the chunk below.)
      constexpr int WIN = 4;
      const int     npairs = T / 2 + 1;
      const int     jj = 64 * J - (k0 - 1) + lane;
      if (!__any(jj >= 0 && jj < ny)) {  // a plane without any line in the grid: nothing was sent, and the ring keeps the zero element
        if (lane == 0) bx_put16(sc, 0xffff);
        return;
      }
      const int     so = BX_OS + (lane + 2) * (BX_RS + 1);
      double       *mb  = Q.mbox + ((size_t)(c - 1) * Q.nb + J) * (size_t)(T / 2 + 8) * 128 + 2 * lane;
      bx_double2    sentinel2;
      sentinel2.x = sentinel2.y = __longlong_as_double((long long)BX_SENTINEL);
      int       base = 0, spins = 0, rcur = -2 * lane;  // rcur: this lane's first row of pair base, 2 base - 2 lane
      long long t0   = 0;
      unsigned  n_it = 0, n_empty = 0, n_lim = 0;  // HIPX_SORBOX_STATS: passes, passes that staged nothing, waits for ring space
      bool      probe = true;
      unsigned long long tk_rt = 0, tk_pass = 0, n_staged = 0;
      // The wave shares its SIMD with a compute wave that keeps the vector unit ~70 % busy: what a pass costs is its VECTOR instructions (~100 in
      // the first version, 0.7 us; the memory round trip is 0.4).  Hence: ballots straight from the comparisons, the row number kept running, no
      // per-lane notion of need (the chunk below sends every lane's entry, zero elements included).
      // rows r, r + 1 of this lane's line into the ring (row nx is the line's zero element: the chunk below sent it as such); the entry goes back
      auto put = [&](int r, const bx_double2 &v, double *e) {
        if ((unsigned)r <= (unsigned)nx) {
          const int o = so + (r & (BX_RS - 2));  // (r is even: the pair does not wrap in the ring)
          L[o]     = v.x;
          L[o + 1] = v.y;
        }
        *reinterpret_cast<bx_double2 *>(e) = sentinel2;  // (read once: the entry is ready for the next sweep)
      };
      while (base < npairs) {
        // ring space (32 rows): pair n overwrites the rows plane 0 asked for up to step 2 n - 28
        int lim = (bx_get16(Hrec(0)) + 19) >> 1;
        lim     = lim < npairs ? lim : npairs;
        if (lim <= base) {  // nothing may be staged yet: no memory traffic while plane 0 catches up
          n_lim++;
          __builtin_amdgcn_s_sleep(2);
          if (give_up(spins, t0)) return;
          continue;
        }
        double *ent = mb + (size_t)(base + 7) * 128;
        if (probe) {
          // Nothing had arrived on the last pass: the wave asks for the NEXT flush alone (1 KB in one piece: sixteen requests per round trip of a
          // waiting workgroup) and stages it the moment it is there -- the first pair of a chunk is in the ring one round trip after it became visible
          // (a one-lane probe followed by a pass over four flushes: two round trips and the pass's work, on every hop between chunks).
          bx_double2 pv;
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(pv) : "v"(ent) : "memory");
          if (__builtin_amdgcn_ballot_w64(unset(pv))) {
            __builtin_amdgcn_s_sleep(1);
            if (give_up(spins, t0)) return;
            continue;
          }
          probe = false;
          spins = 0, t0 = 0;
          put(rcur, pv, ent);
          bx_lds_release();
          base += 1;
          rcur += 2;
          if (lane == 0) bx_put16(sc, 2 * base);
          continue;
        }
        n_it++;
        const unsigned long long tk0 = Q.stats ? wall_clock64() : 0;
        bx_double2 v0, v1, v2, v3;
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                     "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                     : "v"(ent)
                     : "memory");
        if (Q.stats) tk_rt += wall_clock64() - tk0;
        int nlead = 0;
        if (!__builtin_amdgcn_ballot_w64(unset(v0))) {
          nlead = 1;
          if (base + 1 < lim && !__builtin_amdgcn_ballot_w64(unset(v1))) {
            nlead = 2;
            if (base + 2 < lim && !__builtin_amdgcn_ballot_w64(unset(v2))) {
              nlead = 3;
              if (base + 3 < lim && !__builtin_amdgcn_ballot_w64(unset(v3))) nlead = 4;
            }
          }
        }
        if (!nlead) {
          n_empty++;
          probe = true;
          if (give_up(spins, t0)) return;
          continue;
        }
        spins = 0, t0 = 0;
        put(rcur, v0, ent);
        if (nlead > 1) put(rcur + 2, v1, ent + 128);
        if (nlead > 2) put(rcur + 4, v2, ent + 256);
        if (nlead > 3) put(rcur + 6, v3, ent + 384);
        bx_lds_release();
        base += nlead;
        rcur += 2 * nlead;
        if (lane == 0) bx_put16(sc, 2 * base);  // (virtual steps -2 .. 2 base - 3 are staged)
        if (Q.stats) tk_pass += wall_clock64() - tk0, n_staged += nlead;
        if (Q.trace && J == 0 && c <= 8 && lane == 0) {
          const unsigned long long now = wall_clock64();
          for (int q = base - nlead; q < base; q++)
            if (q + 7 < 4096) Q.trace[(2 * (c - 1) + 1) * 4096 + q + 7] = now;
        }
      }
      if (Q.stats && lane == 0) {
        atomicAdd(Q.stats + 7, (unsigned long long)n_it);
        atomicAdd(Q.stats + 8 + 2 * Q.nb * Q.nch, (unsigned long long)n_empty);
        atomicAdd(Q.stats + 9 + 2 * Q.nb * Q.nch, (unsigned long long)n_lim);
        atomicAdd(Q.stats + 10 + 2 * Q.nb * Q.nch, tk_rt);
        atomicAdd(Q.stats + 11 + 2 * Q.nb * Q.nch, tk_pass);
        atomicAdd(Q.stats + 12 + 2 * Q.nb * Q.nch, n_staged);
      }
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------------------- stager of plane k0 + w
  // Everything of a group -- right-hand side (a), west lines (b) -- is requested from memory two groups before it is staged: 16-byte loads, PPL lanes
  // per line segment, the halo loads at agent scope (a row of x another workgroup has not written yet reads as the sentinel and is asked for again
  // when its turn comes).
  // A plane has TWO stagers, one for the even and one for the odd groups: a stager's loads for its next group (two groups on) are requested right
  // after it has published the current one and are the only ones it has in flight when it needs them.  (One stager with two groups in flight
  // waits, at every group, for the loads it has JUST issued -- the compiler's vmcnt(0) -- and the whole workgroup settles at memory latency / 8 per
  // step.)
  const int  role = hw / BX_P;  // groups of this parity
  const int  w = hw % BX_P, k = k0 + w;
  const bool plane_ok = k < nz;
  bx_lds_u16 *hrec = Hrec(w);  // {relaxed(w), relaxed(w + 1), staged(w), -}
  bx_double2  pre[NPASS], hv[2];
  // halo task tk of group g for this lane (lanes 0 .. 2 PPL - 1), both out of block J - 1's last two lanes: 0 = the two west lines of plane k (lines
  // 64 J - k - 2, - 1), rows [G g - 4 w, + G); 1 (plane 0's stagers) = the two lines of the south plane k0 - 1 in front of block J's, rows [G g, + G)
  auto halo_desc = [&](int g, const int tk, bool &in, bool &mem, int &off, int &r, const double *&ptr) {
    const int q = lane / PPL, qd = lane % PPL, kk = tk == 0 ? k : k0 - 1, jj = 64 * J - kk - 2 + q;
    r   = BX_G * g - (tk == 0 ? 4 * w : 0) + 2 * qd;
    in  = lane < 2 * PPL && r >= 0 && r <= nx;  // (row nx: the line's zero element)
    mem = in && r < nx && kk >= 0 && kk < nz && jj >= 0 && jj < ny;
    off = (tk == 0 ? BX_OW + (w * 2 + (q & 1)) * BX_RW : BX_OS + (q & 1) * (BX_RS + 1)) + (r & (BX_RW - 1));
    ptr = mem ? Q.xout + phys(r, jj, kk) : Q.xout;
  };
  auto issue = [&](int g) {
#pragma unroll
    for (int p = 0; p < NPASS; p++) {  // (a) rows [G g - 2 - 2 s - 4 w, + G) of the 64 lines
      const int  s = LPP * p + lane / PPL, qd = lane % PPL, jj = 64 * J - k + s;
      const int  r = BX_G * g - 2 - 2 * s - 4 * w + 2 * qd;
      const bool ok = plane_ok && r >= 0 && r < nx && jj >= 0 && jj < ny;
      pre[p]        = *reinterpret_cast<const bx_double2 *>(ok ? Q.rhs + phys(r, jj, k) : Q.rhs);  // (a lane without a row reads element 0: never stored)
    }
#pragma unroll
    for (int tk = 0; tk < 2; tk++) {
      if (tk == 1 && w != 0) break;
      bool          in, mem;
      int           off, r;
      const double *ptr;
      halo_desc(g, tk, in, mem, off, r, ptr);
      hv[tk].x = hv[tk].y = z0;
      if (mem) hv[tk] = ld(ptr);  // (no dummy address for the other lanes: every waiting workgroup asking for xout[0] is a hot spot on one channel)
    }
  };
  unsigned npoll = 0, nring = 0;
  auto stage = [&](int g) -> bool {
    if (Q.stats && (bx_get16(hrec) < BX_G * g - (BX_RW - 16) || bx_get16(hrec + 1) < BX_G * g - (BX_RW - 16))) nring++;
    // ring space: rhs ring (32 rows) -> the wave is past step G g - 24; west / south rings (32 rows, read up to 13 steps after staging) -> waves w and w + 1 past G g - 16
    if (!bx_wait16(hrec, BX_G * g - (BX_RW - 16), abortw, gerr) || !bx_wait16(hrec + 1, BX_G * g - (BX_RW - 16), abortw, gerr)) return false;
    bx_lds_acquire();
#pragma unroll
    for (int p = 0; p < NPASS; p++) {
      const int s = LPP * p + lane / PPL, qd = lane % PPL, jj = 64 * J - k + s;
      const int r = BX_G * g - 2 - 2 * s - 4 * w + 2 * qd;
      if (plane_ok && r >= 0 && r < nx && jj >= 0 && jj < ny) {
        const int bo = BX_OB + (w * 64 + s) * (BX_RB + 1);
        L[bo + (r & (BX_RB - 1))]       = REV ? pre[p].y : pre[p].x;
        L[bo + ((r + 1) & (BX_RB - 1))] = REV ? pre[p].x : pre[p].y;
      }
    }
#pragma unroll
    for (int tk = 0; tk < 2; tk++) {
      if (tk == 1 && w != 0) break;
      bool          in, mem;
      int           off, r;
      const double *ptr;
      halo_desc(g, tk, in, mem, off, r, ptr);
      if (mem && unset(hv[tk])) {  // (block J - 1 is a block hop ahead: seen on the first groups of a workgroup at most)
        npoll++;
        int       spins = 0;
        long long t0    = 0;
        for (;;) {
          hv[tk] = ld(ptr);
          if (!unset(hv[tk])) break;
          __builtin_amdgcn_s_sleep(1);
          if (give_up(spins, t0)) break;
        }
      }
      if (in) {
        bx_double2 v = hv[tk];
        if (mem) {
          if (REV) {
            const double tmp = v.x;
            v.x              = v.y;
            v.y              = tmp;
          }
        } else v.x = v.y = z0;
        L[off] = v.x;
        L[off - (r & (BX_RW - 1)) + ((r + 1) & (BX_RW - 1))] = v.y;  // (RW == RS: the same row mask for both rings)
      }
    }
    bx_lds_release();
    if (!bx_wait16(hrec + 2, BX_G * g, abortw, gerr)) return false;  // (groups are published in order: the other stager's group g - 1 first)
    if (lane < 2) bx_put16(lane == 0 ? Crec(w) + 1 : hrec + 2, BX_G * (g + 1));
    return true;
  };
  if (role < Q.ngroups) issue(role);
  for (int g = role; g < Q.ngroups; g += 2) {
    if (!stage(g)) return;
    if (g + 2 < Q.ngroups) issue(g + 2);
  }
  if (Q.stats) {
    atomicAdd(Q.stats + 4, (unsigned long long)(lane == 0 ? nring : 0));
    atomicAdd(Q.stats + 5, (unsigned long long)npoll);
  }
}

__global__ void box_fill_kernel(double *x, size_t n)
{
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) x[q] = __longlong_as_double((long long)BX_SENTINEL);
}

// The rows other workgroups poll in x itself are the last two lines of every block (lanes 62, 63: the west lines of block J + 1, and the two lines of
// the south plane in front of its lines): LOGICAL line j of plane k with (j + k) mod 64 in {62, 63}; rev: the mirrored grid.  One wave per line.
__global__ void box_fill_lines_kernel(double *x, int nx, int ny, int nz, int rev)
{
  const int       lane = threadIdx.x & 63;
  const long long nl = (long long)ny * nz, wpg = blockDim.x >> 6;
  for (long long l = (long long)blockIdx.x * wpg + (threadIdx.x >> 6); l < nl; l += (long long)gridDim.x * wpg) {
    const int j = (int)(l % ny), k = (int)(l / ny);
    const int jl = rev ? ny - 1 - j : j, kl = rev ? nz - 1 - k : k;
    if (((jl + kl) & 63) < 62) continue;
    double *row = x + l * nx;
    for (int i = lane; i < nx; i += 64) row[i] = __longlong_as_double((long long)BX_SENTINEL);
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
  int           nx = 0, ny = 0, nz = 0, nb = 0, nch = 0, T = 0, em = 0;
  long long     m = 0;
  double        coefF[13], coefB[13], diag = 0.0, z0 = 0.0;
  int2         *d_order = nullptr;
  unsigned int *d_ctl = nullptr;
  unsigned long long *d_stats = nullptr, *d_trace = nullptr;
  double       *d_mbox = nullptr;
  size_t        mbox_len = 0;
  unsigned int *h_err = nullptr;  // pinned: the error word of the application before
  hipEvent_t    ev_err = nullptr;
  bool          err_pending = false;
};
typedef hipxSorBox_s *hipxSorBox;

extern "C" void hipxSorBoxFree_(void *p)
{
  hipxSorBox B = (hipxSorBox)p;
  if (!B) return;
  (void)hipFree(B->d_order);
  (void)hipFree(B->d_ctl);
  (void)hipFree(B->d_stats);
  (void)hipFree(B->d_trace);
  (void)hipFree(B->d_mbox);
  if (B->h_err) (void)hipHostFree(B->h_err);
  if (B->ev_err) (void)hipEventDestroy(B->ev_err);
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
  if (Lx < 4 || Lx > 60000 || (Lx & 1) || S % Lx || m % S || S / Lx < 3) return HIPX_SUCCESS;  // (16-bit step counters: lines of up to 60000 rows)
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
  B->T       = (nx + 3 + 2 * 63 + 4 * (BX_P - 1) + 3) & ~3;  // rows -2 .. nx of the last lane of the last wave, rounded up to the unrolled loop's four
  // ticket order: by estimated start (a chunk hop ~ 4 P steps + a hand-off, a block hop ~ 128 steps + a hand-off); every dependency of (J, c)
  // -- (J - 1, c), (J - 1, c - 1), (J, c - 1) -- starts earlier
  std::vector<int2> order;
  for (int c = 0; c < B->nch; c++)
    for (int J = 0; J < B->nb; J++) order.push_back(make_int2(J, c));
  std::stable_sort(order.begin(), order.end(), [](const int2 &a, const int2 &b) { return a.y * 5 + a.x * 16 < b.y * 5 + b.x * 16; });
  if (hipMalloc((void **)&B->d_order, sizeof(int2) * order.size()) != hipSuccess || hipMemcpy(B->d_order, order.data(), sizeof(int2) * order.size(), hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(HIPX_ERR_HIP_BASE, "hipMalloc", __FILE__, __LINE__));
  // the mailbox of the chunks' top planes: filled with the sentinel once -- whoever reads an entry puts the sentinel back
  B->mbox_len = (size_t)B->nb * (size_t)(B->nch > 1 ? B->nch - 1 : 0) * (size_t)(B->T / 2 + 8) * 128;  // (eight flushes of padding per workgroup: the poller reads four at a time)
  if (B->mbox_len) {
    // (+ 4 KB: a poller reads four flushes at a time, the last pass of the last workgroup up to three flushes past its region)
    if (hipMalloc((void **)&B->d_mbox, sizeof(double) * (B->mbox_len + 4 * 128)) != hipSuccess) return bail(fail(HIPX_ERR_HIP_BASE, "hipMalloc", __FILE__, __LINE__));
    box_fill_kernel<<<4096, 256, 0, st>>>(B->d_mbox, B->mbox_len);
  }
  *out = B;
  return HIPX_SUCCESS;
}

extern "C" void hipxSorBoxShape_(void *p, int *nx, int *ny, int *nz)
{
  hipxSorBox B = (hipxSorBox)p;
  *nx = B->nx, *ny = B->ny, *nz = B->nz;
}

template <bool REV, int EM, int KIND, int G>
static int box_launch_g(hipxSorBox B, BoxParams &Q)
{
  static bool attr = false;
  auto        kern = &sor_box_kernel<REV, EM, KIND, G>;
  Q.ngroups        = (B->T + G - 1) / G;
  if (!attr) {
    HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BX_LDS_BYTES));
    attr = true;
  }
  kern<<<(unsigned)(B->nb * B->nch), BX_THREADS, BX_LDS_BYTES, rt().compute>>>(Q);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

template <bool REV, int EM, int KIND>
static int box_launch(hipxSorBox B, BoxParams &Q)
{
  return box_launch_g<REV, EM, KIND, 8>(B, Q);  // (groups of 4 and 2 steps were measured in round 6: the stagers cannot keep up -- 2.88 / 3.42 ms against 2.80)
}

// One zero-guess sweep.  kind 0: forward (rhs = b; t and x written; xfull = 0 inside a symmetric application: only the lines the schedule itself
// hands between workgroups reach xout); kind 1: backward after forward (rhs = t); kind 2: backward alone (rhs = b).  The polled rows of xout must hold
// the sentinel (hipxSorBoxFill_) before the launch: a row of x is its own ready flag between workgroups.  Vectors 16-byte aligned.
extern "C" int hipxSorBoxRun_(void *p, int kind, const double *rhs, double *tout, double *xout, double omega, double shift, int xfull)
{
  hipxSorBox B = (hipxSorBox)p;
  BoxParams  Q;
  Q.nx = B->nx, Q.ny = B->ny, Q.nz = B->nz, Q.nb = B->nb, Q.nch = B->nch, Q.T = B->T, Q.ngroups = 0, Q.xfull = xfull;
  Q.m = B->m;
  const bool plain = (omega == 1.0 && shift <= 0.0);
  Q.idiag = plain ? 1.0 / B->diag : omega / (shift + B->diag);  // MatInvertDiagonalForSOR_SeqAIJ aij.c:1797-1840
  Q.z0    = B->z0;
  Q.omw   = 1.0 - omega;
  Q.rhs = rhs, Q.xout = xout, Q.tout = tout, Q.order = B->d_order, Q.ctl = B->d_ctl, Q.mbox = B->d_mbox;
  static const bool want_stats = getenv("HIPX_SORBOX_STATS") != nullptr;  // developer switch: where the waves wait
  static const int  dbg        = getenv("HIPX_SORBOX_DEBUG") ? atoi(getenv("HIPX_SORBOX_DEBUG")) : 0;
  Q.dbg = dbg;
  Q.stats = nullptr;
  if (want_stats) {
    const size_t nst = 16 + 2 * (size_t)B->nb * B->nch;
    if (!B->d_stats) HIPX_HIP(hipMalloc((void **)&B->d_stats, nst * sizeof(unsigned long long)));
    HIPX_HIP(hipMemsetAsync(B->d_stats, 0, nst * sizeof(unsigned long long), rt().compute));
    Q.stats = B->d_stats;
  }
  Q.trace = nullptr;
  static const int stats_level = getenv("HIPX_SORBOX_STATS") ? atoi(getenv("HIPX_SORBOX_STATS")) : 0;
  if (stats_level >= 3) {
    if (!B->d_trace) HIPX_HIP(hipMalloc((void **)&B->d_trace, 16 * 4096 * sizeof(unsigned long long)));
    HIPX_HIP(hipMemsetAsync(B->d_trace, 0, 16 * 4096 * sizeof(unsigned long long), rt().compute));
    Q.trace = B->d_trace;
  }
  memcpy(Q.coef, kind == 0 ? B->coefF : B->coefB, sizeof(Q.coef));
  HIPX_HIP(hipMemsetAsync(B->d_ctl, 0, sizeof(unsigned int), rt().compute));  // the ticket; the error word is sticky until read
  const bool box27 = B->em == 0x1FFF;
  int        ierr;
  if (kind == 0) ierr = box27 ? box_launch<false, 0x1FFF, 0>(B, Q) : box_launch<false, 0x1410, 0>(B, Q);
  else if (kind == 1) ierr = box27 ? box_launch<true, 0x1FFF, 1>(B, Q) : box_launch<true, 0x1410, 1>(B, Q);
  else ierr = box27 ? box_launch<true, 0x1FFF, 2>(B, Q) : box_launch<true, 0x1410, 2>(B, Q);
  if (!ierr && want_stats) {
    const size_t                    nst = 16 + 2 * (size_t)B->nb * B->nch;
    std::vector<unsigned long long> hvv(nst);
    unsigned long long             *h = hvv.data();
    HIPX_HIP(hipMemcpyAsync(h, B->d_stats, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost, rt().compute));
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    {  // time stamps (100 MHz): when workgroup (J, c) started and when its top plane finished, relative to the first start
      unsigned long long t0 = ~0ull;
      for (size_t q = 0; q < (size_t)B->nb * B->nch; q++)
        if (h[8 + 2 * q] && h[8 + 2 * q] < t0) t0 = h[8 + 2 * q];
      auto us = [&](int J, int c, int e) { return (double)(long long)(h[8 + 2 * ((size_t)c * B->nb + J) + e] - t0) * 0.01; };
      const int cl = B->nch - 1, Jl = B->nb - 1, cm = B->nch / 2;
      fprintf(stderr, "[sorbox stamps us] (J,c): start..end  (0,0) %.1f..%.1f  (0,1) %.1f..%.1f  (0,2) %.1f..%.1f  (1,0) %.1f..%.1f  (2,0) %.1f..%.1f  (1,1) %.1f..%.1f  (0,%d) %.1f..%.1f  (%d,%d) %.1f..%.1f  (%d,%d) %.1f..%.1f\n",
              us(0, 0, 0), us(0, 0, 1), us(0, cl > 0 ? 1 : 0, 0), us(0, cl > 0 ? 1 : 0, 1), us(0, cl > 1 ? 2 : 0, 0), us(0, cl > 1 ? 2 : 0, 1), us(Jl > 0 ? 1 : 0, 0, 0), us(Jl > 0 ? 1 : 0, 0, 1),
              us(Jl > 1 ? 2 : 0, 0, 0), us(Jl > 1 ? 2 : 0, 0, 1), us(Jl > 0 ? 1 : 0, cl > 0 ? 1 : 0, 0), us(Jl > 0 ? 1 : 0, cl > 0 ? 1 : 0, 1), cm, us(0, cm, 0), us(0, cm, 1), Jl, 0, us(Jl, 0, 0), us(Jl, 0, 1), Jl, cl,
              us(Jl, cl, 0), us(Jl, cl, 1));
    }
    if (getenv("HIPX_SORBOX_STATS") && atoi(getenv("HIPX_SORBOX_STATS")) >= 2) {  // every workgroup's start / end (us): row J, column c
      unsigned long long t0 = ~0ull;
      for (size_t q = 0; q < (size_t)B->nb * B->nch; q++)
        if (h[8 + 2 * q] && h[8 + 2 * q] < t0) t0 = h[8 + 2 * q];
      for (int e = 0; e < 2; e++) {
        fprintf(stderr, "[sorbox %s us, kind %d, %d blocks x %d chunks]\n", e ? "end" : "start", kind, B->nb, B->nch);
        for (int J = 0; J < B->nb; J++) {
          fprintf(stderr, "  J=%d:", J);
          for (int c = 0; c < B->nch; c++) fprintf(stderr, " %.0f", (double)(long long)(h[8 + 2 * ((size_t)c * B->nb + J) + e] - t0) * 0.01);
          fprintf(stderr, "\n");
        }
      }
    }
    if (Q.trace) {  // how long a flush of workgroup (0, c) takes to the ring of workgroup (0, c + 1)
      std::vector<unsigned long long> tr(16 * 4096);
      HIPX_HIP(hipMemcpy(tr.data(), B->d_trace, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      for (int cc = 0; cc < 8 && cc + 1 < B->nch; cc++) {
        const unsigned long long *a = tr.data() + (size_t)(2 * cc) * 4096, *b = a + 4096;
        std::vector<double>       d;
        for (int f = 0; f < 4096; f++)
          if (a[f] && b[f]) d.push_back((double)(long long)(b[f] - a[f]) * 0.01);
        if (d.empty()) continue;
        std::vector<double> e = d;
        std::sort(e.begin(), e.end());
        unsigned long long f0 = 0, f1 = 0;
        int                nf = 0;
        for (int f = 0; f < 4096; f++)
          if (a[f]) {
            if (!f0) f0 = a[f];
            f1 = a[f];
            nf++;
          }
        fprintf(stderr, "[sorbox trace kind %d] (0, %d) mailed -> in the ring of (0, %d), us: min %.2f median %.2f max %.2f; flush 2: %.2f, 20: %.2f, 100: %.2f, 300: %.2f, 600: %.2f, last: %.2f; a flush every %.3f us\n", kind, cc, cc + 1,
                e.front(), e[e.size() / 2], e.back(), d.size() > 2 ? d[2] : 0.0, d.size() > 20 ? d[20] : 0.0, d.size() > 100 ? d[100] : 0.0, d.size() > 300 ? d[300] : 0.0, d.size() > 600 ? d[600] : 0.0, d.back(),
                nf > 1 ? (double)(long long)(f1 - f0) * 0.01 / (nf - 1) : 0.0);
      }
    }
    const double nw = (double)B->nb * B->nch * BX_P;
    fprintf(stderr, "[sorbox kind %d %dx%dx%d] per compute wave: steps %.0f, spins waiting for lower plane %.1f, stager %.1f, upper plane %.1f, flusher %.1f; per stager: ring waits %.1f, halo polls %.1f\n", kind,
            B->nx, B->ny, B->nz, (double)h[6] / nw, (double)h[0] / nw, (double)h[1] / nw, (double)h[2] / nw, (double)h[3] / nw, (double)h[4] / nw, (double)h[5] / nw);
    {
      const double np = (double)B->nb * (B->nch - 1 > 0 ? B->nch - 1 : 1);
      const size_t o  = 8 + 2 * (size_t)B->nb * B->nch;
      fprintf(stderr, "[sorbox south pollers] per poller: window passes %.1f, of them empty %.1f, waits for ring space %.1f, for %d pairs\n", (double)h[7] / np, (double)h[o] / np, (double)h[o + 1] / np, B->T / 2 + 1);
      fprintf(stderr, "[sorbox south pollers] per pass: loads %.2f us; a pass that staged something %.2f us, %.2f pairs\n", (double)h[o + 2] * 0.01 / (double)(h[7] ? h[7] : 1),
              (double)h[o + 3] * 0.01 / (double)((h[7] - h[o]) ? (h[7] - h[o]) : 1), (double)h[o + 4] / (double)((h[7] - h[o]) ? (h[7] - h[o]) : 1));
    }
  }
  return ierr;
}

// Before a sweep: the sentinel into the rows of xout that are their own ready flags (rev: the backward sweeps, kinds 1 and 2).  1 / 32 of the
// vector -- until round 6 the whole vector was filled, 2 x 134 MB per application at 256^3.
extern "C" int hipxSorBoxFill_(void *p, double *xout, int rev)
{
  hipxSorBox B = (hipxSorBox)p;
  box_fill_lines_kernel<<<2048, 256, 0, rt().compute>>>(xout, B->nx, B->ny, B->nz, rev);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// The error word of the applications so far (a dependency that was never published: the waits give up after 4 s).  The word of the application
// just enqueued is copied to pinned host memory behind it and looked at when the NEXT application is set up (or the matrix is destroyed): a host
// round trip per application cost 25-40 us of idle GPU (1.2 % of PCApply_SOR at 256^3).  A launch that gave up leaves sentinels (NaN) in its
// result, so a solve does not run on unnoticed; sync != 0 waits for the word of the application just enqueued (tests, HIPX_SORBOX_SYNC_CHECK=1).
extern "C" int hipxSorBoxError_(void *p, unsigned int *err, int sync)
{
  hipxSorBox B = (hipxSorBox)p;
  static const bool always = getenv("HIPX_SORBOX_SYNC_CHECK") != nullptr;
  *err = 0;
  if (!B->h_err) {
    HIPX_HIP(hipHostMalloc((void **)&B->h_err, 2 * sizeof(unsigned int), hipHostMallocDefault));
    B->h_err[0] = B->h_err[1] = 0;
    HIPX_HIP(hipEventCreateWithFlags(&B->ev_err, hipEventDisableTiming));
  }
  if (B->err_pending) {  // the previous application's word: long there
    HIPX_HIP(hipEventSynchronize(B->ev_err));
    *err |= B->h_err[0];
    B->err_pending = false;
  }
  HIPX_HIP(hipMemcpyAsync(B->h_err, B->d_ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipEventRecord(B->ev_err, rt().compute));
  B->err_pending = true;
  if (sync || always) {
    HIPX_HIP(hipEventSynchronize(B->ev_err));
    *err |= B->h_err[0];
    B->err_pending = false;
  }
  if (*err) {
    HIPX_HIP(hipMemsetAsync(B->d_ctl, 0, 2 * sizeof(unsigned int), rt().compute));
    if (B->mbox_len) box_fill_kernel<<<4096, 256, 0, rt().compute>>>(B->d_mbox, B->mbox_len);  // (a launch that gave up leaves entries behind)
  }
  return HIPX_SUCCESS;
}
