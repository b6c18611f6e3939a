// hipx_pipe.hip -- the vector work of the pipelined CG variants (KSPPIPECG pipecg.c:20-160, KSPGROPPCG groppcg.c:23-150) as ONE pass per iteration.
//
// Those loops end every iteration with a block of VecAYPX / VecAXPY calls on up to ten vectors followed by the sums of the next iteration
// (VecNormBegin / VecDotBegin ... PetscCommSplitReductionBegin, comb.c:338,379).  Run one kernel per call that block moves 24 + 6 vector
// passes; every operation of it is elementwise, so element i of the whole block depends on element i of its operands only:
//
//   batch_axpy_kernel<PID, COMP>   a RECORDED batch (the drop-in's lazy queue, plugin/vechipx.c): each operand loaded once, the recorded
//                                  operations applied to the registers in their order (y + a x / x + b y: product and sum rounded separately,
//                                  as the separate kernels and the reference's loops bvec1.c:70-83, dvec2.c:753-780 do -- the vectors come out
//                                  bit-identical), each changed vector stored once, and the sums the callers ask for next (u.u, r.r, r.u, w.u, ...)
//                                  accumulated from the registers.  The batches are compile-time PROGRAMS (kBatchProgs): operand slots numbered
//                                  by first appearance; a recorded queue that matches none runs as separate kernels.
//   pipecg_update_kernel<...>      the host layer's launch-ahead PIPECG (host/hipx_ksp.c: HipxKSPSolve_PIPECG): the same block with alpha and
//                                  beta formed ON THE DEVICE from the sums of the kernel queued before (the IEEE quotients of pipecg.c:132,
//                                  138-139), the x update deferred by one iteration (x += alpha_{i-1} p_{i-1} before p is overwritten: what is
//                                  enqueued ahead when the loop stops is the update that was due), m = B w (PCJACOBI / PCNONE) for the product
//                                  that follows, and m of the iteration before RE-FORMED from w instead of read (the same product).
//
// Both walk the vectors with all workgroups together, round by round (the launch shape that measured best for cg_fused_kernel), 16-byte
// accesses, and end in the shared reduction epilogue (hipx_reduce.h: plain or compensated sums, last workgroup folds).
#include "hipx_internal.h"
#include "hipx_reduce.h"
#include <cstdlib>

using namespace hipx;

namespace {

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

inline unsigned pipe_grid(hipx_int n)
{
  static const hipx_int cap = [] {
    const char *e = getenv("HIPX_RED_BLOCKS");
    const int   v = e ? atoi(e) : 256;
    return (hipx_int)((v >= 1 && v <= kRedBlocks) ? v : 256);
  }();
  hipx_int g = (n + kRedThreads * 2 - 1) / (kRedThreads * 2);
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

// ---------------------------------------------------------------- recorded batches
constexpr int kBatchMaxOps = HIPX_BATCH_MAX_OPS, kBatchMaxVecs = HIPX_BATCH_MAX_VECS, kBatchMaxDots = HIPX_BATCH_MAX_DOTS;
struct BatchProg {
  int nops, nvec, ndots;
  int kind[kBatchMaxOps], y[kBatchMaxOps], x[kBatchMaxOps];  // kind 1: y += s x (VecAXPY_Seq), 2: y = x + s y (VecAYPX_Seq)
  int da[kBatchMaxDots], db[kBatchMaxDots];                  // sums a . b of the vectors AFTER the batch
};
// slots: for each operation in order, y then x; a vector gets the next number when it appears first
constexpr BatchProg kBatchProgs[] = {
  // 0: KSPSolve_PIPECG, i > 0 (pipecg.c:140-150): z = n + b z; q = m + b q; p = u + b p; s = w + b s; x += a p; u -= a q; w -= a z; r -= a s
  //    slots z0 n1 q2 m3 p4 u5 s6 w7 x8 r9; then (pipecg.c:106-113) VecNormBegin(U | R), VecDotBegin(R, U), VecDotBegin(W, U)
  {8, 10, 4, {2, 2, 2, 2, 1, 1, 1, 1}, {0, 2, 4, 6, 8, 5, 7, 9}, {1, 3, 5, 7, 4, 2, 0, 6}, {5, 9, 9, 7}, {5, 9, 5, 5}},
  // 1: KSPSolve_PIPECG, i == 0 (pipecg.c:147-150 behind four VecCopy): x0 p1 u2 q3 w4 z5 r6 s7
  {4, 8, 4, {1, 1, 1, 1}, {0, 2, 4, 6}, {1, 3, 5, 7}, {2, 6, 6, 4}, {2, 6, 2, 2}},
  // 2: KSPSolve_GROPPCG (groppcg.c:98-100): x += a p; r -= a s; z -= a S; then VecNormBegin(z | r), VecDotBegin(r, z): x0 p1 r2 s3 z4 S5
  {3, 6, 3, {1, 1, 1}, {0, 2, 4}, {1, 3, 5}, {4, 2, 2}, {4, 2, 4}},
  // 3: KSPSolve_GROPPCG (groppcg.c:135-136): p = z + b p; s = Z + b s; then VecDotBegin(p, s) (groppcg.c:87): p0 z1 s2 Z3
  {2, 4, 1, {2, 2}, {0, 2}, {1, 3}, {0}, {2}},
  // 4: KSPSolve_PIPECR (pipecr.c:113-121): z = n + b z; q = m + b q; p = u + b p; x += a p; u -= a q; w -= a z  (then VecNormBegin(U), VecDotBegin(W, U),
  //    VecDotBegin(M, W) -- M comes out of the PCApply that follows: only the first two here): z0 n1 q2 m3 p4 u5 x6 w7
  {6, 8, 2, {2, 2, 2, 1, 1, 1}, {0, 2, 4, 6, 5, 7}, {1, 3, 5, 4, 2, 0}, {5, 7}, {5, 5}},
};
constexpr int kNumBatchProgs = (int)(sizeof(kBatchProgs) / sizeof(kBatchProgs[0]));

struct BatchArgs {
  double *v[kBatchMaxVecs];
  double  s[kBatchMaxOps];
};

template <int PID>
struct BatchInfo {
  static constexpr BatchProg P = kBatchProgs[PID];
  static constexpr bool written(int v)
  {
    for (int k = 0; k < P.nops; k++)
      if (P.y[k] == v) return true;
    return false;
  }
};

template <int PID, bool COMP>
__global__ __launch_bounds__(kRedThreads) void batch_axpy_kernel(const BatchArgs a, hipx_int n, bool vec, RedOut out)
{
  using I = BatchInfo<PID>;
  constexpr BatchProg P  = I::P;
  constexpr int       NV = P.nvec, NO = P.nops, ND = P.ndots;
  Acc<COMP>           acc[ND];
  const auto          ops = [&](double(&r)[NV]) {
#pragma unroll
    for (int k = 0; k < NO; k++) {
      if (P.kind[k] == 1) r[P.y[k]] = r[P.y[k]] + a.s[k] * r[P.x[k]];
      else r[P.y[k]] = r[P.x[k]] + a.s[k] * r[P.y[k]];
    }
#pragma unroll
    for (int d = 0; d < ND; d++) acc[d].prod(r[P.da[d]], r[P.db[d]]);
  };
  if (vec) {
    const hipx_int n2 = n >> 1, stride = (hipx_int)gridDim.x * kRedThreads;
    for (hipx_int q = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; q < n2; q += stride) {
      double2 t[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) t[v] = reinterpret_cast<const double2 *>(a.v[v])[q];
      double r0[NV], r1[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) {
        r0[v] = t[v].x;
        r1[v] = t[v].y;
      }
      ops(r0);
      ops(r1);
#pragma unroll
      for (int v = 0; v < NV; v++)
        if (I::written(v)) reinterpret_cast<double2 *>(a.v[v])[q] = make_double2(r0[v], r1[v]);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      double r[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) r[v] = a.v[v][n - 1];
      ops(r);
#pragma unroll
      for (int v = 0; v < NV; v++)
        if (I::written(v)) a.v[v][n - 1] = r[v];
    }
  } else {
    const hipx_int stride = (hipx_int)gridDim.x * kRedThreads;
    for (hipx_int i = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += stride) {
      double r[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) r[v] = a.v[v][i];
      ops(r);
#pragma unroll
      for (int v = 0; v < NV; v++)
        if (I::written(v)) a.v[v][i] = r[v];
    }
  }
  finish_sums<ND, COMP>(acc, out);
}

template <int PID>
int batch_launch(const BatchArgs &a, hipx_int n, bool vec, int slot)
{
  const unsigned g = pipe_grid(n);
  if (rt().red_exact) batch_axpy_kernel<PID, true><<<g, kRedThreads, 0, rt().compute>>>(a, n, vec, red_out(slot));
  else batch_axpy_kernel<PID, false><<<g, kRedThreads, 0, rt().compute>>>(a, n, vec, red_out(slot));
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// ---------------------------------------------------------------- PIPECG update, scalars on the device
// One pass of pipecg.c:132-150 (+ the deferred x update of the iteration before, + m = B w for the product that follows):
//   FIRST (i == 0):  alpha = gamma / delta;                 z = n; q = m; p = u; s = w                     (pipecg.c:132-136)
//   otherwise:       beta = gamma / gammaold; alpha = gamma / (delta - beta / alphaold * gamma);           (pipecg.c:138-139)
//                    x += alphaold p (the update iteration i-1 left behind); z = n + beta z; q = m + beta q; p = u + beta p; s = w + beta s
//   then             u -= alpha q; w -= alpha z; r -= alpha s;   m = B w   (PC: 0 PCNONE m = w, 1 constant Jacobi diagonal, 2 streamed diagonal)
//   sums of the new state: [0] u.u (NRM 1) | r.r (NRM 2) | nothing (0), [1] gamma = r.u, [2] delta = w.u               (pipecg.c:106-113)
// m of iteration i (read by "q = m + beta q") is w * d of the w this kernel reads: re-formed, not read.
struct PipeCGArgs {
  double       *z, *q, *p, *s, *x, *u, *w, *r, *m;
  const double *nv;             // n = A m of this iteration
  const double *d;              // PC 2: inverse diagonal
  double        dconst;         // PC 1
  const double *sums;           // device: [dp-sum, gamma, delta] of this iteration (all-reduced)
  const double *sums_old;       // device: the same of the iteration before (gamma_old = sums_old[1]); unused when FIRST
  const double *alpha_old;      // device: alpha of the iteration before; unused when FIRST
  double       *alpha_out;      // device: alpha of this iteration (one thread writes it)
};

template <bool FIRST, int PC, int NRM, bool COMP, int U>
__global__ __launch_bounds__(kRedThreads) void pipecg_update_kernel(const PipeCGArgs a, hipx_int n, bool vec, RedOut out, const int nt)
{
  const double gamma = a.sums[1], delta = a.sums[2];
  double       alpha, beta = 0.0, aold = 0.0;
  if (FIRST) alpha = gamma / delta;
  else {
    aold  = *a.alpha_old;
    beta  = gamma / a.sums_old[1];
    alpha = gamma / (delta - beta / aold * gamma);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.alpha_out = alpha;
  const double ma = -alpha;
  Acc<COMP>    acc[3];
  const auto   one = [&](double &z, double &q, double &p, double &s, double &x, double &u, double &w, double &r, double &m, const double nn, const double dd) {
    const double mo = (PC == 0) ? w : w * dd;  // m of this iteration: what the kernel before stored for the product
    if (FIRST) {
      z = nn;
      q = mo;
      p = u;
      s = w;
    } else {
      x = x + aold * p;
      z = nn + beta * z;
      q = mo + beta * q;
      p = u + beta * p;
      s = w + beta * s;
    }
    u = u + ma * q;
    w = w + ma * z;
    r = r + ma * s;
    m = (PC == 0) ? w : w * dd;
    if (NRM == 1) acc[0].prod(u, u);
    else if (NRM == 2) acc[0].prod(r, r);
    acc[1].prod(r, u);
    acc[2].prod(w, u);
  };
  if (vec) {
    // U pairs per stream in flight per thread (all loads of a round issued before the first use); nt bit 0: non-temporal loads, bit 1: non-temporal stores
    // (every vector is touched once per iteration and the working set is ten vectors: nothing read here is in a cache by the time it is wanted again)
    const hipx_int n2 = n >> 1, stride = (hipx_int)gridDim.x * kRedThreads;
    const double2  z0 = {0.0, 0.0};
    const auto     ld = [&](const double *b, hipx_int i) -> double2 {
      if (nt & 1) {
        double2 v;
        v.x = __builtin_nontemporal_load(b + 2 * i);
        v.y = __builtin_nontemporal_load(b + 2 * i + 1);
        return v;
      }
      return reinterpret_cast<const double2 *>(b)[i];
    };
    const auto st = [&](double *b, hipx_int i, const double2 v) {
      if (nt & 2) {
        __builtin_nontemporal_store(v.x, b + 2 * i);
        __builtin_nontemporal_store(v.y, b + 2 * i + 1);
      } else reinterpret_cast<double2 *>(b)[i] = v;
    };
    for (hipx_int i0 = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i0 < n2; i0 += U * stride) {
      double2 z[U], q[U], p[U], s[U], x[U], u[U], w[U], r[U], nn[U], dd[U], m[U];
#pragma unroll
      for (int k = 0; k < U; k++) {
        const hipx_int i  = i0 + k * stride;
        const hipx_int ic = i < n2 ? i : i0;  // (beyond the end: re-read the first pair of the round, results dropped)
        z[k]  = FIRST ? z0 : ld(a.z, ic);
        q[k]  = FIRST ? z0 : ld(a.q, ic);
        p[k]  = FIRST ? z0 : ld(a.p, ic);
        s[k]  = FIRST ? z0 : ld(a.s, ic);
        x[k]  = FIRST ? z0 : ld(a.x, ic);
        u[k]  = ld(a.u, ic);
        w[k]  = ld(a.w, ic);
        r[k]  = ld(a.r, ic);
        nn[k] = ld(a.nv, ic);
        dd[k] = (PC == 2) ? ld(a.d, ic) : make_double2(a.dconst, a.dconst);
      }
#pragma unroll
      for (int k = 0; k < U; k++) {
        const hipx_int i = i0 + k * stride;
        if (i < n2) {
          one(z[k].x, q[k].x, p[k].x, s[k].x, x[k].x, u[k].x, w[k].x, r[k].x, m[k].x, nn[k].x, dd[k].x);
          one(z[k].y, q[k].y, p[k].y, s[k].y, x[k].y, u[k].y, w[k].y, r[k].y, m[k].y, nn[k].y, dd[k].y);
          st(a.z, i, z[k]);
          st(a.q, i, q[k]);
          st(a.p, i, p[k]);
          st(a.s, i, s[k]);
          if (!FIRST) st(a.x, i, x[k]);
          st(a.u, i, u[k]);
          st(a.w, i, w[k]);
          st(a.r, i, r[k]);
          if (PC != 0) st(a.m, i, m[k]);  // (PCNONE: m IS w -- the product reads w)
        }
      }
    }
  }
  {  // rows beyond the 16-byte pairs (odd n), or every row of vectors that are not 16-byte aligned
    const hipx_int lo = vec ? (n & ~(hipx_int)1) : 0, stride = (hipx_int)gridDim.x * kRedThreads;
    for (hipx_int i = lo + (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += stride) {
      double z = FIRST ? 0.0 : a.z[i], q = FIRST ? 0.0 : a.q[i], p = FIRST ? 0.0 : a.p[i], s = FIRST ? 0.0 : a.s[i], x = FIRST ? 0.0 : a.x[i];
      double u = a.u[i], w = a.w[i], r = a.r[i], m;
      one(z, q, p, s, x, u, w, r, m, a.nv[i], (PC == 2) ? a.d[i] : a.dconst);
      a.z[i] = z;
      a.q[i] = q;
      a.p[i] = p;
      a.s[i] = s;
      if (!FIRST) a.x[i] = x;
      a.u[i] = u;
      a.w[i] = w;
      a.r[i] = r;
      if (PC != 0) a.m[i] = m;
    }
  }
  finish_sums<3, COMP>(acc, out);
}

template <bool FIRST, int PC, int NRM>
void pipecg_go2(const PipeCGArgs &a, hipx_int n, bool vec, const RedOut &o)
{
  const unsigned   g  = pipe_grid(n);
  static const int nt = getenv("HIPX_PIPECG_NT") ? atoi(getenv("HIPX_PIPECG_NT")) : 0;   // developer switches (timing experiments): non-temporal loads (1) / stores (2) / both (3)
  static const int u2 = getenv("HIPX_PIPECG_U") ? atoi(getenv("HIPX_PIPECG_U")) : 1;     // pairs per stream in flight per thread
  if (rt().red_exact) pipecg_update_kernel<FIRST, PC, NRM, true, 1><<<g, kRedThreads, 0, rt().compute>>>(a, n, vec, o, nt);
  else if (u2 == 2) pipecg_update_kernel<FIRST, PC, NRM, false, 2><<<g, kRedThreads, 0, rt().compute>>>(a, n, vec, o, nt);
  else pipecg_update_kernel<FIRST, PC, NRM, false, 1><<<g, kRedThreads, 0, rt().compute>>>(a, n, vec, o, nt);
}
template <bool FIRST, int PC>
void pipecg_go1(const PipeCGArgs &a, int nrm, hipx_int n, bool vec, const RedOut &o)
{
  if (nrm == 1) pipecg_go2<FIRST, PC, 1>(a, n, vec, o);
  else if (nrm == 2) pipecg_go2<FIRST, PC, 2>(a, n, vec, o);
  else pipecg_go2<FIRST, PC, 0>(a, n, vec, o);
}
template <bool FIRST>
void pipecg_go0(const PipeCGArgs &a, int pc, int nrm, hipx_int n, bool vec, const RedOut &o)
{
  if (pc == 0) pipecg_go1<FIRST, 0>(a, nrm, n, vec, o);
  else if (pc == 1) pipecg_go1<FIRST, 1>(a, nrm, n, vec, o);
  else pipecg_go1<FIRST, 2>(a, nrm, n, vec, o);
}

// ---------------------------------------------------------------- Gropp's CG (groppcg.c:23-140), scalars on the device
// The reference's iteration i:  t = p.s [reduction 1] || S = B s;  alpha = gamma / t;  x += alpha p; r -= alpha s; z -= alpha S;
//                               dp, gammaNew = r.z [reduction 2] || Z = A z;  test;  beta = gammaNew / gamma;  p = z + beta p; s = Z + beta s.
// Two passes per iteration here (+ the product Z = A z between them):
//   gropp_dir_kernel     D(i), i > 1: x += alpha_{i-1} p (the update iteration i-1 left behind, applied before p changes); p = z + beta p; s = Z + beta s with
//                        beta = gammaNew_{i-1} / gamma_{i-2} (groppcg.c:132: formed on the device); partial sums of t_i = p.s.  5 reads + 3 writes.
//   gropp_update_kernel  U(i): alpha_i = gamma_{i-1} / t_i (groppcg.c:96, on the device; one thread stores it); r -= alpha s; z -= alpha (s .* d) -- S = B s of groppcg.c:92
//                        re-formed per element, never stored (PC 0: PCNONE S = s, 1: constant Jacobi diagonal, 2: streamed) --; sums dp (z.z | r.r | none), gammaNew = r.z.
//                        3 reads + 2 writes (+ d).
// Element by element the operations and their order are the reference's (VecAXPY y + a x, VecAYPX x + b y, VecPointwiseMult x * y): the vectors are bit-identical.
struct GroppDirArgs {
  double       *p, *s, *x;
  const double *z, *Z;
  const double *gnew, *gold, *alpha_old;  // device scalars: gammaNew_{i-1}, gamma_{i-2}, alpha_{i-1}
};
template <bool COMP>
__global__ __launch_bounds__(kRedThreads) void gropp_dir_kernel(const GroppDirArgs a, hipx_int n, bool vec, RedOut out)
{
  const double beta = *a.gnew / *a.gold, aold = *a.alpha_old;
  Acc<COMP>    acc[1];
  const auto   one = [&](double &p, double &s, double &x, const double z, const double Zv) {
    x = x + aold * p;
    p = z + beta * p;
    s = Zv + beta * s;
    acc[0].prod(p, s);
  };
  const hipx_int stride = (hipx_int)gridDim.x * kRedThreads;
  if (vec) {
    constexpr int  U  = 2;  // pairs per stream in flight per thread (all loads of a round issued before the first use: 10 x 16 B per thread)
    const hipx_int n2 = n >> 1;
    for (hipx_int i0 = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i0 < n2; i0 += U * stride) {
      double2 p[U], s[U], x[U], z[U], Zv[U];
#pragma unroll
      for (int k = 0; k < U; k++) {
        const hipx_int i = i0 + k * stride < n2 ? i0 + k * stride : i0;
        p[k]  = reinterpret_cast<const double2 *>(a.p)[i];
        s[k]  = reinterpret_cast<const double2 *>(a.s)[i];
        x[k]  = reinterpret_cast<const double2 *>(a.x)[i];
        z[k]  = reinterpret_cast<const double2 *>(a.z)[i];
        Zv[k] = reinterpret_cast<const double2 *>(a.Z)[i];
      }
#pragma unroll
      for (int k = 0; k < U; k++) {
        const hipx_int i = i0 + k * stride;
        if (i < n2) {
          one(p[k].x, s[k].x, x[k].x, z[k].x, Zv[k].x);
          one(p[k].y, s[k].y, x[k].y, z[k].y, Zv[k].y);
          reinterpret_cast<double2 *>(a.x)[i] = x[k];
          reinterpret_cast<double2 *>(a.p)[i] = p[k];
          reinterpret_cast<double2 *>(a.s)[i] = s[k];
        }
      }
    }
  }
  for (hipx_int i = (vec ? (n & ~(hipx_int)1) : 0) + (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += stride) {
    double p = a.p[i], s = a.s[i], x = a.x[i];
    one(p, s, x, a.z[i], a.Z[i]);
    a.x[i] = x;
    a.p[i] = p;
    a.s[i] = s;
  }
  finish_sums<1, COMP>(acc, out);
}

struct GroppUpdArgs {
  double       *r, *z;
  const double *s, *d;
  double        dconst;
  const double *gamma, *t;  // device scalars: gamma_{i-1}, t_i
  double       *alpha_out;
};
template <int PC, int NRM, bool COMP>
__global__ __launch_bounds__(kRedThreads) void gropp_update_kernel(const GroppUpdArgs a, hipx_int n, bool vec, RedOut out)
{
  const double alpha = *a.gamma / *a.t, ma = -alpha;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.alpha_out = alpha;
  Acc<COMP>  acc[2];
  const auto one = [&](double &r, double &z, const double s, const double dd) {
    const double S = (PC == 0) ? s : s * dd;
    r = r + ma * s;
    z = z + ma * S;
    if (NRM == 1) acc[0].prod(z, z);
    else if (NRM == 2) acc[0].prod(r, r);
    acc[1].prod(r, z);
  };
  const hipx_int stride = (hipx_int)gridDim.x * kRedThreads;
  if (vec) {
    constexpr int  U  = (PC == 2) ? 2 : 4;  // pairs per stream in flight per thread (12 / 8 x 16 B)
    const hipx_int n2 = n >> 1;
    for (hipx_int i0 = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i0 < n2; i0 += U * stride) {
      double2 r[U], z[U], s[U], dd[U];
#pragma unroll
      for (int k = 0; k < U; k++) {
        const hipx_int i = i0 + k * stride < n2 ? i0 + k * stride : i0;
        r[k]  = reinterpret_cast<const double2 *>(a.r)[i];
        z[k]  = reinterpret_cast<const double2 *>(a.z)[i];
        s[k]  = reinterpret_cast<const double2 *>(a.s)[i];
        dd[k] = (PC == 2) ? reinterpret_cast<const double2 *>(a.d)[i] : make_double2(a.dconst, a.dconst);
      }
#pragma unroll
      for (int k = 0; k < U; k++) {
        const hipx_int i = i0 + k * stride;
        if (i < n2) {
          one(r[k].x, z[k].x, s[k].x, dd[k].x);
          one(r[k].y, z[k].y, s[k].y, dd[k].y);
          reinterpret_cast<double2 *>(a.r)[i] = r[k];
          reinterpret_cast<double2 *>(a.z)[i] = z[k];
        }
      }
    }
  }
  for (hipx_int i = (vec ? (n & ~(hipx_int)1) : 0) + (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += stride) {
    double r = a.r[i], z = a.z[i];
    one(r, z, a.s[i], (PC == 2) ? a.d[i] : a.dconst);
    a.r[i] = r;
    a.z[i] = z;
  }
  finish_sums<2, COMP>(acc, out);
}

template <int PC>
void gropp_update_go(const GroppUpdArgs &a, int nrm, hipx_int n, bool vec, const RedOut &o)
{
  const unsigned g = pipe_grid(n);
  hipStream_t    st = rt().compute;
  if (rt().red_exact) {
    if (nrm == 1) gropp_update_kernel<PC, 1, true><<<g, kRedThreads, 0, st>>>(a, n, vec, o);
    else if (nrm == 2) gropp_update_kernel<PC, 2, true><<<g, kRedThreads, 0, st>>>(a, n, vec, o);
    else gropp_update_kernel<PC, 0, true><<<g, kRedThreads, 0, st>>>(a, n, vec, o);
  } else {
    if (nrm == 1) gropp_update_kernel<PC, 1, false><<<g, kRedThreads, 0, st>>>(a, n, vec, o);
    else if (nrm == 2) gropp_update_kernel<PC, 2, false><<<g, kRedThreads, 0, st>>>(a, n, vec, o);
    else gropp_update_kernel<PC, 0, false><<<g, kRedThreads, 0, st>>>(a, n, vec, o);
  }
}

}  // namespace

int hipx::launch_pipecg_update(const hipxPipeCGVecs *v, const double *d, double dconst, int normkind, int first, const double *dev_sums, const double *dev_sums_old, const double *dev_alpha_old,
                               double *dev_alpha_out, hipx_int n, const RedOut &o)
{
  PipeCGArgs a;
  a.z = v->z; a.q = v->q; a.p = v->p; a.s = v->s; a.x = v->x; a.u = v->u; a.w = v->w; a.r = v->r; a.m = v->m;
  a.nv        = v->n;
  a.d         = d;
  a.dconst    = dconst;
  a.sums      = dev_sums;
  a.sums_old  = dev_sums_old;
  a.alpha_old = dev_alpha_old;
  a.alpha_out = dev_alpha_out;
  const bool vec = n >= 2 && aligned16(a.z) && aligned16(a.q) && aligned16(a.p) && aligned16(a.s) && aligned16(a.x) && aligned16(a.u) && aligned16(a.w) && aligned16(a.r) && aligned16(a.m) &&
                   aligned16(a.nv) && aligned16(a.d);
  const int  pc  = d ? 2 : (dconst == 1.0 ? 0 : 1);  // (a constant diagonal of exactly 1.0 is PCNONE's copy: w * 1.0 returns w bit for bit either way)
  if (first) pipecg_go0<true>(a, pc, normkind, n, vec, o);
  else pipecg_go0<false>(a, pc, normkind, n, vec, o);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

static int batch_find(int nops, const int *kind, const int *yslot, const int *xslot, int nvec)
{
  for (int p = 0; p < kNumBatchProgs; p++) {
    const BatchProg &P = kBatchProgs[p];
    if (P.nops != nops || P.nvec != nvec) continue;
    bool same = true;
    for (int k = 0; k < nops && same; k++) same = P.kind[k] == kind[k] && P.y[k] == yslot[k] && P.x[k] == xslot[k];
    if (same) return p;
  }
  return -1;
}

int hipx::launch_gropp_dir(double *p, double *s, double *x, const double *z, const double *Z, const double *dev_gamma_new, const double *dev_gamma_old, const double *dev_alpha_old, hipx_int n,
                           const RedOut &o)
{
  GroppDirArgs a;
  a.p = p; a.s = s; a.x = x; a.z = z; a.Z = Z;
  a.gnew = dev_gamma_new; a.gold = dev_gamma_old; a.alpha_old = dev_alpha_old;
  const bool vec = n >= 2 && aligned16(p) && aligned16(s) && aligned16(x) && aligned16(z) && aligned16(Z);
  if (rt().red_exact) gropp_dir_kernel<true><<<pipe_grid(n), kRedThreads, 0, rt().compute>>>(a, n, vec, o);
  else gropp_dir_kernel<false><<<pipe_grid(n), kRedThreads, 0, rt().compute>>>(a, n, vec, o);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipx::launch_gropp_update(double *r, double *z, const double *s, const double *d, double dconst, int normkind, const double *dev_gamma, const double *dev_t, double *dev_alpha_out, hipx_int n,
                              const RedOut &o)
{
  GroppUpdArgs a;
  a.r = r; a.z = z; a.s = s; a.d = d; a.dconst = dconst; a.gamma = dev_gamma; a.t = dev_t; a.alpha_out = dev_alpha_out;
  const bool vec = n >= 2 && aligned16(r) && aligned16(z) && aligned16(s) && aligned16(d);
  if (d) gropp_update_go<2>(a, normkind, n, vec, o);
  else if (dconst == 1.0) gropp_update_go<0>(a, normkind, n, vec, o);
  else gropp_update_go<1>(a, normkind, n, vec, o);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

extern "C" {

int hipxVecBatchProgramKnown(int nops, const int *kind, const int *yslot, const int *xslot, int nvec)
{
  if (nops < 1 || nops > kBatchMaxOps || nvec < 2 || nvec > kBatchMaxVecs || !kind || !yslot || !xslot) return 0;
  return batch_find(nops, kind, yslot, xslot, nvec) >= 0 ? 1 : 0;
}

int hipxVecBatchAXPYDotsBegin(int nops, const int *kind, const int *yslot, const int *xslot, const double *s, int nvec, double *const *vec, hipx_int n, int slot, int *ndots, int *da, int *db)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(ndots && da && db, "null output");
  *ndots = -1;
  HIPX_ARG(nops >= 1 && nops <= kBatchMaxOps && nvec >= 2 && nvec <= kBatchMaxVecs && kind && yslot && xslot && s && vec, "batch shape");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "reduction slot out of range");
  if (n <= 0) return HIPX_SUCCESS;
  const int pid = batch_find(nops, kind, yslot, xslot, nvec);
  if (pid < 0) return HIPX_SUCCESS;  // not a known program: nothing enqueued, *ndots = -1
  BatchArgs a;
  bool      al = n >= 2;
  for (int v = 0; v < kBatchMaxVecs; v++) {
    a.v[v] = vec[v < nvec ? v : 0];
    HIPX_ARG(a.v[v], "null vector");
    al = al && aligned16(a.v[v]);
  }
  for (int v = 0; v < nvec; v++)
    for (int u = 0; u < v; u++) HIPX_ARG(vec[u] != vec[v], "the vectors of a batch must be distinct");
  for (int k = 0; k < kBatchMaxOps; k++) a.s[k] = k < nops ? s[k] : 0.0;
  int ierr;
  switch (pid) {
  case 0: ierr = batch_launch<0>(a, n, al, slot); break;
  case 1: ierr = batch_launch<1>(a, n, al, slot); break;
  case 2: ierr = batch_launch<2>(a, n, al, slot); break;
  case 3: ierr = batch_launch<3>(a, n, al, slot); break;
  case 4: ierr = batch_launch<4>(a, n, al, slot); break;
  default: return fail(HIPX_ERR_ARG, "batch program table", __FILE__, __LINE__);
  }
  if (ierr) return ierr;
  const BatchProg &P = kBatchProgs[pid];
  *ndots             = P.ndots;
  for (int d = 0; d < P.ndots; d++) {
    da[d] = P.da[d];
    db[d] = P.db[d];
  }
  return HIPX_SUCCESS;
}

int hipxGroppCGDirectionBegin(double *p, double *s, double *x, const double *z, const double *Z, const double *dev_gamma_new, const double *dev_gamma_old, const double *dev_alpha_old, hipx_int n,
                              int slot, double *dev_t_out)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(p && s && x && z && Z && dev_gamma_new && dev_gamma_old && dev_alpha_old && dev_t_out, "null argument");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && n > 0, "bad slot / empty vector");
  return launch_gropp_dir(p, s, x, z, Z, dev_gamma_new, dev_gamma_old, dev_alpha_old, n, red_out(slot, true, dev_t_out));
}

int hipxGroppCGUpdateBegin(double *r, double *z, const double *s, const double *d, double dconst, int normkind, const double *dev_gamma, const double *dev_t, double *dev_alpha_out, hipx_int n,
                           int slot, double *dev_sums2_out)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(r && z && s && dev_gamma && dev_t && dev_alpha_out && dev_sums2_out, "null argument");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && n > 0, "bad slot / empty vector");
  return launch_gropp_update(r, z, s, d, dconst, normkind, dev_gamma, dev_t, dev_alpha_out, n, red_out(slot, true, dev_sums2_out));
}

int hipxPipeCGUpdateBegin(const hipxPipeCGVecs *v, const double *d, double dconst, int normkind, int first, const double *dev_sums, const double *dev_sums_old, const double *dev_alpha_old,
                          double *dev_alpha_out, hipx_int n, int slot, double *dev_sums_out)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(v && dev_sums && dev_alpha_out && dev_sums_out && (first || (dev_sums_old && dev_alpha_old)), "null argument");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && n > 0, "bad slot / empty vector");
  return launch_pipecg_update(v, d, dconst, normkind, first, dev_sums, dev_sums_old, dev_alpha_old, dev_alpha_out, n, red_out(slot, true, dev_sums_out));
}

}  // extern "C"
