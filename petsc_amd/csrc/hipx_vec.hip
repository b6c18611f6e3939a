// hipx_vec.hip -- Vec BLAS-1 kernels for gfx950 (wave64), behind include/hipx.h.
//
// Elementwise ops: HBM-bound streams.  Each thread moves 4 x 16 B per operand per launch (all loads
// issued before the first use), blocks of 256 threads, enough blocks to cover the vector: at N = 16.7 M
// that is 8192 workgroups, >> 256 CUs.  No FMA contraction (-ffp-contract=off): y + a*x rounds the
// product and the sum separately, exactly like the reference's scalar loops / reference BLAS, so every
// elementwise result is bit-identical to the CPU path (src/vec/vec/impls/seq/{bvec1,bvec2,dvec2}.c).
//
// Reductions: one launch, fixed grid (kRedBlocks x 256), per-thread register accumulation in a fixed
// element order -> wave64 __shfl_down tree -> LDS across the 4 waves -> per-block partial -> the last
// block to arrive (agent-scope release/acquire around a ticket counter) folds the partials in fixed
// order and writes the scalar into pinned, device-mapped host memory.  The host only waits on the
// stream: no D2H memcpy call on the dot/norm path (3 such waits per CG iteration, cg.c:258,309,344).
#include "hipx_internal.h"
#include "hipx_reduce.h"
#include <cmath>
#include <cstring>

using namespace hipx;

namespace {

constexpr int EW_UNROLL = 4;

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- elementwise, 1..3 operands
template <class F>
__global__ __launch_bounds__(kEwThreads) void ew1_kernel(double *y, hipx_int n, F f, bool vec)
{
  const hipx_int base = (hipx_int)blockIdx.x * (kEwThreads * EW_UNROLL) + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    double2       *y2 = reinterpret_cast<double2 *>(y);
    double2        v[EW_UNROLL];
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) v[k] = y2[p];
    }
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) {
        v[k].x = f(v[k].x);
        v[k].y = f(v[k].y);
        y2[p]  = v[k];
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = f(y[n - 1]);
  } else {
    for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads) y[i] = f(y[i]);
  }
}

template <class F>
__global__ __launch_bounds__(kEwThreads) void ew2_kernel(double *y, const double *x, hipx_int n, F f, bool vec)
{
  const hipx_int base = (hipx_int)blockIdx.x * (kEwThreads * EW_UNROLL) + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    double2       *y2 = reinterpret_cast<double2 *>(y);
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    double2        vy[EW_UNROLL], vx[EW_UNROLL];
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) {
        vy[k] = y2[p];
        vx[k] = x2[p];
      }
    }
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) {
        vy[k].x = f(vy[k].x, vx[k].x);
        vy[k].y = f(vy[k].y, vx[k].y);
        y2[p]   = vy[k];
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = f(y[n - 1], x[n - 1]);
  } else {
    for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads) y[i] = f(y[i], x[i]);
  }
}

template <class F>
__global__ __launch_bounds__(kEwThreads) void ew3_kernel(double *w, const double *x, const double *y, hipx_int n, F f, bool vec)
{
  const hipx_int base = (hipx_int)blockIdx.x * (kEwThreads * EW_UNROLL) + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    double2       *w2 = reinterpret_cast<double2 *>(w);
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *y2 = reinterpret_cast<const double2 *>(y);
    double2        vw[EW_UNROLL], vx[EW_UNROLL], vy[EW_UNROLL];
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) {
        if (F::reads_w) vw[k] = w2[p];
        vx[k] = x2[p];
        vy[k] = y2[p];
      }
    }
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) {
        double2 o;
        o.x   = f(vw[k].x, vx[k].x, vy[k].x);
        o.y   = f(vw[k].y, vx[k].y, vy[k].y);
        w2[p] = o;
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) w[n - 1] = f(F::reads_w ? w[n - 1] : 0.0, x[n - 1], y[n - 1]);
  } else {
    for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads)
      w[i] = f(F::reads_w ? w[i] : 0.0, x[i], y[i]);
  }
}

inline unsigned ew_grid(hipx_int n, bool vec)
{
  if (vec) {
    hipx_int n2 = n >> 1;
    hipx_int g  = (n2 + kEwThreads * EW_UNROLL - 1) / (kEwThreads * EW_UNROLL);
    return (unsigned)(g < 1 ? 1 : g);
  }
  hipx_int g = (n + kEwThreads - 1) / kEwThreads;
  if (g > kEwMaxBlocks) g = kEwMaxBlocks;
  return (unsigned)(g < 1 ? 1 : g);
}

template <class F>
int launch_ew1(double *y, hipx_int n, F f)
{
  if (n <= 0) return HIPX_SUCCESS;
  bool vec = aligned16(y) && n >= 2;
  ew1_kernel<F><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(y, n, f, vec);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}
template <class F>
int launch_ew2(double *y, const double *x, hipx_int n, F f)
{
  if (n <= 0) return HIPX_SUCCESS;
  bool vec = aligned16(y) && aligned16(x) && n >= 2;
  ew2_kernel<F><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(y, x, n, f, vec);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}
template <class F>
int launch_ew3(double *w, const double *x, const double *y, hipx_int n, F f)
{
  if (n <= 0) return HIPX_SUCCESS;
  bool vec = aligned16(w) && aligned16(x) && aligned16(y) && n >= 2;
  ew3_kernel<F><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(w, x, y, n, f, vec);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// functors (each mirrors one reference loop; see the ABI comments in hipx.h)
struct FSet { double a; __device__ double operator()(double) const { return a; } };
struct FScale { double a; __device__ double operator()(double y) const { return y * a; } };
struct FShift { double a; __device__ double operator()(double y) const { return y + a; } };
struct FRecip { __device__ double operator()(double y) const { return y != 0.0 ? 1.0 / y : y; } };  // vinv.c:1208-1229
struct FAbs { __device__ double operator()(double y) const { return fabs(y); } };
struct FAxpy { double a; __device__ double operator()(double y, double x) const { return y + a * x; } };            // daxpy
struct FAypx { double b; __device__ double operator()(double y, double x) const { return x + b * y; } };            // dvec2.c:774
struct FXmy { __device__ double operator()(double y, double x) const { return x - y; } };                           // dvec2.c:767
struct FAxpby { double a, b; __device__ double operator()(double y, double x) const { return a * x + b * y; } };    // bvec1.c:111
struct FAx { double a; __device__ double operator()(double, double x) const { return a * x; } };                    // bvec1.c:108
struct FWaxpy { static constexpr bool reads_w = false; double a; __device__ double operator()(double, double x, double y) const { return y + a * x; } };
struct FWadd { static constexpr bool reads_w = false; __device__ double operator()(double, double x, double y) const { return y + x; } };
struct FWsub { static constexpr bool reads_w = false; __device__ double operator()(double, double x, double y) const { return y - x; } };
struct FPmult { static constexpr bool reads_w = false; __device__ double operator()(double, double x, double y) const { return x * y; } };
struct FPdiv { static constexpr bool reads_w = false; __device__ double operator()(double, double x, double y) const { return y == 0.0 ? (x == 0.0 ? 1.0 : 0.0) : x / y; } };
// bvec1.c:120-147: four association orders
struct FAbc0 { static constexpr bool reads_w = true; double b, c; __device__ double operator()(double z, double x, double y) const { return x + b * y + c * z; } };
struct FAbc1 { static constexpr bool reads_w = true; double a, b; __device__ double operator()(double z, double x, double y) const { return a * x + b * y + z; } };
struct FAbc2 { static constexpr bool reads_w = false; double a, b; __device__ double operator()(double, double x, double y) const { return a * x + b * y; } };
struct FAbc3 { static constexpr bool reads_w = true; double a, b, c; __device__ double operator()(double z, double x, double y) const { return a * x + b * y + c * z; } };

// CG: p = z + b p (cg.c:249, general-beta loop dvec2.c:774) and the x += a p_old (cg.c:305) left over from the previous
// iteration in one pass: p is read once.  Same arithmetic per element as the two separate kernels.
// ZR = true: z is not stored anywhere; it is re-formed as r * dconst (constant Jacobi diagonal: the product cg_fused_kernel
// would have written), so the iteration has no z stream at all.  UPX = false: no pending x update (first pass after a flush).
template <bool DEVS, bool ZR, bool UPX>
__global__ __launch_bounds__(kEwThreads) void cg_aypx_axpy_kernel(double *p, double *x, const double *z, double dconst, double b_arg, double a_arg, const double *dev_beta_new,
                                                                   const double *dev_beta_old, const double *dev_dpi, hipx_int n, bool vec)
{
  // DEVS: b = beta_new / beta_old (cg.c:248) and a = beta_old / dpi (cg.c:288) from device-resident results
  const double b = DEVS ? (*dev_beta_new / *dev_beta_old) : b_arg;
  const double a = DEVS ? (*dev_beta_old / *dev_dpi) : a_arg;
  const hipx_int base = (hipx_int)blockIdx.x * (kEwThreads * EW_UNROLL) + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    double2       *p2 = reinterpret_cast<double2 *>(p), *x2 = reinterpret_cast<double2 *>(x);
    const double2 *z2 = reinterpret_cast<const double2 *>(z);
    double2        vp[EW_UNROLL], vx[EW_UNROLL], vz[EW_UNROLL];
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int q = base + k * kEwThreads;
      if (q < n2) {
        vp[k] = p2[q];
        if (UPX) vx[k] = x2[q];
        vz[k] = z2[q];
      }
    }
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int q = base + k * kEwThreads;
      if (q < n2) {
        if (ZR) {
          vz[k].x = vz[k].x * dconst;
          vz[k].y = vz[k].y * dconst;
        }
        if (UPX) {
          vx[k].x = vx[k].x + a * vp[k].x;
          vx[k].y = vx[k].y + a * vp[k].y;
          x2[q]   = vx[k];
        }
        vp[k].x = vz[k].x + b * vp[k].x;
        vp[k].y = vz[k].y + b * vp[k].y;
        p2[q]   = vp[k];
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const double po = p[n - 1], zv = ZR ? z[n - 1] * dconst : z[n - 1];
      if (UPX) x[n - 1] = x[n - 1] + a * po;
      p[n - 1] = zv + b * po;
    }
  } else {
    for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads) {
      const double po = p[i], zv = ZR ? z[i] * dconst : z[i];
      if (UPX) x[i] = x[i] + a * po;
      p[i] = zv + b * po;
    }
  }
}

// Single-reduction CG (KSPSolve_CG_SingleReduction, cg.c:364-534): the five vector updates between two reductions in ONE pass:
//   p = z + b p (cg.c:470, VecAYPX)   w = s + b w (cg.c:477: w = A p by recurrence)   x += a p (cg.c:490)   r -= a w (cg.c:491)   z = r .* d (PCApply_Jacobi,
//   cg.c:493) | z = r (PCNONE: d == NULL)
// element by element the operations of the five reference loops in their order (z's old value feeds p before the new one is stored).
__global__ __launch_bounds__(kEwThreads) void cg_sr_update_kernel(double *p, double *w, double *x, double *r, double *z, const double *s, const double *d, double b, double a, hipx_int n)
{
  const double ma = -a;
  for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads) {
    const double pn = z[i] + b * p[i];
    const double wn = s[i] + b * w[i];
    p[i] = pn;
    w[i] = wn;
    x[i] = x[i] + a * pn;
    const double rn = r[i] + ma * wn;
    r[i] = rn;
    z[i] = d ? rn * d[i] : rn;
  }
}

// The same with its scalars formed ON THE DEVICE (round 5: launch-ahead single-reduction CG): sums = {z.z, z.s, z.r} of the reduction queued before this
// kernel (delta = sums[1], beta = sums[2]), st_old = {beta, dpi, a} of the iteration before, st_new <- {beta, dpi, a} of this one (one thread writes it: the
// next iteration's kernel reads it).  b = beta / betaold (cg.c:464), dpi = delta - beta * beta * dpiold / (betaold * betaold) (cg.c:478), a = beta / dpi
// (cg.c:488): the host's expressions in the host's order -- the same doubles.  The x update is the one the iteration BEFORE left behind (x += a_old p_old,
// cg.c:490 of that iteration, applied before p changes), so that whatever has been queued ahead when the host stops the loop has applied exactly the
// updates of completed iterations; the host applies the last one (HipxKSPCGFlush).
__global__ __launch_bounds__(kEwThreads) void cg_sr_update_dev_kernel(double *p, double *w, double *x, double *r, double *z, const double *s, const double *d, const double *sums,
                                                                      const double *st_old, double *st_new, hipx_int n)
{
  const double beta = sums[2], delta = sums[1], betaold = st_old[0], dpiold = st_old[1], aold = st_old[2];
  const double b    = beta / betaold;
  const double dpi  = delta - beta * beta * dpiold / (betaold * betaold);
  const double a    = beta / dpi;
  const double ma   = -a;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st_new[0] = beta;
    st_new[1] = dpi;
    st_new[2] = a;
  }
  for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads) {
    const double po = p[i];
    x[i]            = x[i] + aold * po;
    const double pn = z[i] + b * po;
    const double wn = s[i] + b * w[i];
    p[i] = pn;
    w[i] = wn;
    const double rn = r[i] + ma * wn;
    r[i] = rn;
    z[i] = d ? rn * d[i] : rn;
  }
}

// ---------------------------------------------------------------- swap
__global__ __launch_bounds__(kEwThreads) void swap_kernel(double *x, double *y, hipx_int n)
{
  for (hipx_int i = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; i < n; i += (hipx_int)gridDim.x * kEwThreads) {
    double t = x[i];
    x[i]     = y[i];
    y[i]     = t;
  }
}

// ---------------------------------------------------------------- MAXPY: y (= beta y) += sum_j a_j x_j, up to 8 vectors per pass
constexpr int MAXPY_B = 8;
struct MaxpyArgs {
  const double *x[MAXPY_B];
  double        a[MAXPY_B];
};
// mode: 0 y += ..., 1 y = beta*y + ..., 2 y = 0 + ... (beta == 0: VecSet(y,0) then MAXPY, rvector.c:1434)
template <int NV>
__global__ __launch_bounds__(kEwThreads) void maxpy_kernel(double *y, MaxpyArgs args, double beta, int mode, hipx_int n, bool vec)
{
  const hipx_int base = (hipx_int)blockIdx.x * (kEwThreads * EW_UNROLL) + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    double2       *y2 = reinterpret_cast<double2 *>(y);
#pragma unroll
    for (int k = 0; k < EW_UNROLL; k++) {
      hipx_int p = base + k * kEwThreads;
      if (p < n2) {
        double2 acc = (mode == 2) ? make_double2(0.0, 0.0) : y2[p];
        if (mode == 1) {
          acc.x *= beta;
          acc.y *= beta;
        }
        // association of dvec2.c:658-693 / petscaxpy.h:197-235: groups of (nv & 3) then fours, each group summed
        // left to right before being added to y.
        constexpr int REM = NV & 3;
        if (REM) {
          double2 g = make_double2(0.0, 0.0);
#pragma unroll
          for (int j = 0; j < REM; j++) {
            double2 xv = reinterpret_cast<const double2 *>(args.x[j])[p];
            if (j == 0) {
              g.x = args.a[j] * xv.x;
              g.y = args.a[j] * xv.y;
            } else {
              g.x = g.x + args.a[j] * xv.x;
              g.y = g.y + args.a[j] * xv.y;
            }
          }
          acc.x += g.x;
          acc.y += g.y;
        }
#pragma unroll
        for (int j0 = REM; j0 < NV; j0 += 4) {
          double2 g = make_double2(0.0, 0.0);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            double2 xv = reinterpret_cast<const double2 *>(args.x[j0 + j])[p];
            if (j == 0) {
              g.x = args.a[j0] * xv.x;
              g.y = args.a[j0] * xv.y;
            } else {
              g.x = g.x + args.a[j0 + j] * xv.x;
              g.y = g.y + args.a[j0 + j] * xv.y;
            }
          }
          acc.x += g.x;
          acc.y += g.y;
        }
        y2[p] = acc;
      }
    }
  }
  // scalar path (unaligned operands) and odd tail
  hipx_int i0, step;
  if (vec) {
    if (!((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)) return;
    i0   = n - 1;
    step = n;
  } else {
    i0   = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x;
    step = (hipx_int)gridDim.x * kEwThreads;
  }
  for (hipx_int i = i0; i < n; i += step) {
    double acc = (mode == 2) ? 0.0 : y[i];
    if (mode == 1) acc *= beta;
    constexpr int REM = NV & 3;
    if (REM) {
      double g = 0.0;
#pragma unroll
      for (int j = 0; j < REM; j++) g = (j == 0) ? args.a[j] * args.x[j][i] : g + args.a[j] * args.x[j][i];
      acc += g;
    }
#pragma unroll
    for (int j0 = REM; j0 < NV; j0 += 4) {
      double g = 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++) g = (j == 0) ? args.a[j0] * args.x[j0][i] : g + args.a[j0 + j] * args.x[j0 + j][i];
      acc += g;
    }
    y[i] = acc;
  }
}

// x . y_j for j < NV (NV = 1: dot).  Each thread walks pairs p = tid, tid + T, ... in steps of 2 double2.
template <int NV>
struct MDotArgs {
  const double *y[NV];
};
template <int NV, bool COMP>
__global__ __launch_bounds__(kRedThreads) void mdot_kernel(const double *x, MDotArgs<NV> ys, hipx_int n, bool vec, RedOut out)
{
  Acc<COMP> acc[NV];
  const hipx_int T   = (hipx_int)gridDim.x * kRedThreads;
  const hipx_int tid = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    // each workgroup owns one contiguous chunk (spreads the concurrently active lines over all HBM channels; a
    // grid-wide stride of a power-of-two number of MiB makes every wave hit the same channels at the same time)
    constexpr int  U     = (NV <= 2) ? 4 : 2;  // pairs in flight per thread and operand
    const hipx_int chunk = (n2 + (hipx_int)gridDim.x - 1) / (hipx_int)gridDim.x;
    const hipx_int c0    = (hipx_int)blockIdx.x * chunk;
    const hipx_int c1    = (c0 + chunk < n2) ? c0 + chunk : n2;
    for (hipx_int p = c0 + (hipx_int)threadIdx.x; p < c1; p += U * kRedThreads) {
      double2 xa[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int q = p + u * kRedThreads;
        xa[u]            = (q < c1) ? x2[q] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const double2 *y2 = reinterpret_cast<const double2 *>(ys.y[v]);
        double2        ya[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const hipx_int q = p + u * kRedThreads;
          ya[u]            = (q < c1) ? y2[q] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          acc[v].prod(xa[u].x, ya[u].x);
          acc[v].prod(xa[u].y, ya[u].y);
        }
      }
    }
    if ((n & 1) && tid == 0) {
#pragma unroll
      for (int v = 0; v < NV; v++) acc[v].prod(x[n - 1], ys.y[v][n - 1]);
    }
  } else {
    for (hipx_int i = tid; i < n; i += T) {
#pragma unroll
      for (int v = 0; v < NV; v++) acc[v].prod(x[i], ys.y[v][i]);
    }
  }
  finish_sums<NV, COMP>(acc, out);
}


// GMRES orthogonalisation (VecMDot_Seq dvec2.c:83, the GEMV "T" of dvec2.c:515-590): x . y_j for up to NVMAX vectors in ONE
// pass over x (31 instead of 34 vector reads for 30 vectors, one launch instead of four).  The additions happen in the same
// order as in mdot_kernel<NV> for every NV (per-thread pairs p, p + T, ... in increasing order), so the sums do not depend on
// how a call is batched.
template <int NVMAX, bool COMP>
__global__ __launch_bounds__(kRedThreads) void mdot_wide_kernel(const double *x, MDotArgs<NVMAX> ys, int nv, hipx_int n, bool vec, RedOut out)
{
  Acc<COMP> acc[NVMAX];
  const hipx_int T   = (hipx_int)gridDim.x * kRedThreads;
  const hipx_int tid = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    constexpr int  U     = 2;
    const hipx_int chunk = (n2 + (hipx_int)gridDim.x - 1) / (hipx_int)gridDim.x;
    const hipx_int c0    = (hipx_int)blockIdx.x * chunk;
    const hipx_int c1    = (c0 + chunk < n2) ? c0 + chunk : n2;
    for (hipx_int p = c0 + (hipx_int)threadIdx.x; p < c1; p += U * kRedThreads) {
      double2 xa[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int q = p + u * kRedThreads;
        xa[u]            = (q < c1) ? x2[q] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int v = 0; v < NVMAX; v++) {
        if (v < nv) {
          const double2 *y2 = reinterpret_cast<const double2 *>(ys.y[v]);
          double2        ya[U];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const hipx_int q = p + u * kRedThreads;
            ya[u]            = (q < c1) ? y2[q] : make_double2(0.0, 0.0);
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            acc[v].prod(xa[u].x, ya[u].x);
            acc[v].prod(xa[u].y, ya[u].y);
          }
        }
      }
    }
    if ((n & 1) && tid == 0) {
#pragma unroll
      for (int v = 0; v < NVMAX; v++)
        if (v < nv) acc[v].prod(x[n - 1], ys.y[v][n - 1]);
    }
  } else {
    for (hipx_int i = tid; i < n; i += T) {
#pragma unroll
      for (int v = 0; v < NVMAX; v++)
        if (v < nv) acc[v].prod(x[i], ys.y[v][i]);
    }
  }
  finish_sums<NVMAX, COMP>(acc, out);
}

// VecMAXPY_Seq / VecMAXPBY for up to 36 vectors in ONE pass over y (dvec2.c:658-693: (nv & 3) vectors first, then groups of
// four, each group summed left to right before it is added to y -- the association every batch of maxpy_kernel keeps too)
constexpr int MAXPY_WIDE = 36;
struct MaxpyWideArgs {
  const double *x[MAXPY_WIDE];
  double        a[MAXPY_WIDE];
};
template <int REM>
__global__ __launch_bounds__(kEwThreads) void maxpy_wide_kernel(double *y, MaxpyWideArgs args, int nv, double beta, int mode, hipx_int n)
{
  const hipx_int n2 = n >> 1;
  double2       *y2 = reinterpret_cast<double2 *>(y);
  for (hipx_int p = (hipx_int)blockIdx.x * kEwThreads + threadIdx.x; p < n2; p += (hipx_int)gridDim.x * kEwThreads) {
    double2 acc = (mode == 2) ? make_double2(0.0, 0.0) : y2[p];
    if (mode == 1) {
      acc.x *= beta;
      acc.y *= beta;
    }
    if (REM) {
      double2 g = make_double2(0.0, 0.0);
#pragma unroll
      for (int j = 0; j < REM; j++) {
        const double2 xv = reinterpret_cast<const double2 *>(args.x[j])[p];
        if (j == 0) {
          g.x = args.a[j] * xv.x;
          g.y = args.a[j] * xv.y;
        } else {
          g.x = g.x + args.a[j] * xv.x;
          g.y = g.y + args.a[j] * xv.y;
        }
      }
      acc.x += g.x;
      acc.y += g.y;
    }
    for (int j0 = REM; j0 < nv; j0 += 4) {
      double2 xv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) xv[j] = reinterpret_cast<const double2 *>(args.x[j0 + j])[p];
      double2 g;
      g.x = args.a[j0] * xv[0].x;
      g.y = args.a[j0] * xv[0].y;
#pragma unroll
      for (int j = 1; j < 4; j++) {
        g.x = g.x + args.a[j0 + j] * xv[j].x;
        g.y = g.y + args.a[j0 + j] * xv[j].y;
      }
      acc.x += g.x;
      acc.y += g.y;
    }
    y2[p] = acc;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd tail
    const hipx_int i = n - 1;
    double acc = (mode == 2) ? 0.0 : y[i];
    if (mode == 1) acc *= beta;
    if (REM) {
      double g = 0.0;
#pragma unroll
      for (int j = 0; j < REM; j++) g = (j == 0) ? args.a[j] * args.x[j][i] : g + args.a[j] * args.x[j][i];
      acc += g;
    }
    for (int j0 = REM; j0 < nv; j0 += 4) {
      double g = args.a[j0] * args.x[j0][i];
      for (int j = 1; j < 4; j++) g = g + args.a[j0 + j] * args.x[j0 + j][i];
      acc += g;
    }
    y[i] = acc;
  }
}

// sums of |x| (NORM_1), x*x (NORM_2) in one pass: acc[0] = sum |x|, acc[1] = sum x^2
template <bool COMP>
__global__ __launch_bounds__(kRedThreads) void norm12_kernel(const double *x, hipx_int n, bool vec, RedOut out)
{
  Acc<COMP>      acc[2];
  const hipx_int T      = (hipx_int)gridDim.x * kRedThreads;
  const hipx_int tid    = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x;
  if (vec) {
    const hipx_int n2 = n >> 1;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    constexpr int U = 4;  // pairs in flight per thread (round 6: one load per round left the kernel at 3.9 TB/s, latency-bound); the accumulation order -- p, p + T, ... -- is unchanged
    for (hipx_int p0 = tid; p0 < n2; p0 += U * T) {
      double2 a[U];
#pragma unroll
      for (int u = 0; u < U; u++) a[u] = x2[p0 + u * T < n2 ? p0 + u * T : p0];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (p0 + u * T < n2) {
          acc[0].add(fabs(a[u].x));
          acc[0].add(fabs(a[u].y));
          acc[1].prod(a[u].x, a[u].x);
          acc[1].prod(a[u].y, a[u].y);
        }
    }
    if ((n & 1) && tid == 0) {
      acc[0].add(fabs(x[n - 1]));
      acc[1].prod(x[n - 1], x[n - 1]);
    }
  } else {
    for (hipx_int i = tid; i < n; i += T) {
      acc[0].add(fabs(x[i]));
      acc[1].prod(x[i], x[i]);
    }
  }
  finish_sums<2, COMP>(acc, out);
}

template <bool COMP>
__global__ __launch_bounds__(kRedThreads) void sum_kernel(const double *x, hipx_int n, RedOut out)
{
  Acc<COMP>      acc[1];
  const hipx_int T      = (hipx_int)gridDim.x * kRedThreads;
  for (hipx_int i = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += T) acc[0].add(x[i]);
  finish_sums<1, COMP>(acc, out);
}

// NORM_INFINITY with the reference's NaN propagation (bvec2.c:207-216)
__global__ __launch_bounds__(kRedThreads) void norminf_kernel(const double *x, hipx_int n, RedOut out)
{
  double         acc[1] = {0.0};
  const hipx_int T      = (hipx_int)gridDim.x * kRedThreads;
  for (hipx_int i = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += T) {
    double t = fabs(x[i]);
    acc[0]   = (t > acc[0] || t != t) ? t : acc[0];
  }
  block_finish<1, RED_MAXNAN>(acc, out);
}

// x.y and y.y in one pass (VecDotNorm2)
template <bool COMP>
__global__ __launch_bounds__(kRedThreads) void dotnorm2_kernel(const double *x, const double *y, hipx_int n, RedOut out)
{
  Acc<COMP>      acc[2];
  const hipx_int T      = (hipx_int)gridDim.x * kRedThreads;
  for (hipx_int i = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += T) {
    double a = x[i], b = y[i];
    acc[0].prod(a, b);
    acc[1].prod(b, b);
  }
  finish_sums<2, COMP>(acc, out);
}

// w = x .* y (VecPointwiseMult_Seq bvec2.c:72-97: one product per element) with the sums w.w and w.x of the vector just written, in the same pass.
// What KSPSolve_CG asks for right after PCApply_Jacobi (= VecPointwiseMult(z, r, diag), jacobi.c:354-362): VecNorm(Z) (cg.c:309) and
// VecXDot(Z, R) (cg.c:344) -- the drop-in's vector type keeps the two sums keyed on the vectors (plugin/vechipx.c: reduction cache) and
// answers those calls without another pass over z and r.  Each workgroup owns one contiguous chunk (as mdot_kernel).
template <bool COMP>
__global__ __launch_bounds__(kRedThreads) void pwmult_dots_kernel(double *w, const double *x, const double *y, hipx_int n, bool vec, RedOut out)
{
  Acc<COMP> acc[2];
  if (vec) {
    const hipx_int n2 = n >> 1;
    const double2 *x2 = reinterpret_cast<const double2 *>(x), *y2 = reinterpret_cast<const double2 *>(y);
    double2       *w2 = reinterpret_cast<double2 *>(w);
    constexpr int  U     = 2;
    const hipx_int chunk = (n2 + (hipx_int)gridDim.x - 1) / (hipx_int)gridDim.x;
    const hipx_int c0    = (hipx_int)blockIdx.x * chunk;
    const hipx_int c1    = (c0 + chunk < n2) ? c0 + chunk : n2;
    for (hipx_int p = c0 + (hipx_int)threadIdx.x; p < c1; p += U * kRedThreads) {
      double2 xa[U], ya[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int q = p + u * kRedThreads;
        xa[u]            = (q < c1) ? x2[q] : make_double2(0.0, 0.0);
        ya[u]            = (q < c1) ? y2[q] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int q = p + u * kRedThreads;
        double2        wa;
        wa.x = xa[u].x * ya[u].x;
        wa.y = xa[u].y * ya[u].y;
        if (q < c1) w2[q] = wa;
        acc[0].prod(wa.x, wa.x);
        acc[0].prod(wa.y, wa.y);
        acc[1].prod(wa.x, xa[u].x);
        acc[1].prod(wa.y, xa[u].y);
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const double wv = x[n - 1] * y[n - 1];
      w[n - 1]        = wv;
      acc[0].prod(wv, wv);
      acc[1].prod(wv, x[n - 1]);
    }
  } else {
    const hipx_int T = (hipx_int)gridDim.x * kRedThreads;
    for (hipx_int i = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += T) {
      const double xv = x[i], wv = xv * y[i];
      w[i] = wv;
      acc[0].prod(wv, wv);
      acc[1].prod(wv, xv);
    }
  }
  finish_sums<2, COMP>(acc, out);
}

// fused CG update (cg.c:305-309,344 with PCJACOBI): x += a p; r -= a w; z = r*d; sums z.z, z.r
// UPX = false: the x update is left to cg_aypx_axpy_kernel of the next iteration (p is then read once per iteration)
// DEVS = true: a = *dev_beta / *dev_dpi is formed on the device from the results of kernels queued before this one (the host
// forms the same IEEE quotient for its own bookkeeping), so the launch does not have to wait for the host to see them.
// CONSTD = true: the Jacobi diagonal is one constant (constant-coefficient operators): z = r * dconst without reading d[] --
// the same product, one vector pass less.
template <bool UPX, bool DEVS, bool CONSTD = false, bool COMP = false, bool WIDE = true>
__global__ __launch_bounds__(kRedThreads) void cg_fused_kernel(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a_arg,
                                                                const double *dev_beta, const double *dev_dpi, hipx_int n, bool vec, RedOut out, double dconst = 0.0, int sweep = 0)
{
  const double a = DEVS ? (*dev_beta / *dev_dpi) : a_arg;
  Acc<COMP>      acc[2];
  const hipx_int T      = (hipx_int)gridDim.x * kRedThreads;
  const hipx_int tid    = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x;
  const double   ma     = -a;
  if (vec) {
    const hipx_int n2 = n >> 1;
    double2       *x2 = reinterpret_cast<double2 *>(x), *r2 = reinterpret_cast<double2 *>(r), *z2 = reinterpret_cast<double2 *>(z);
    const double2 *p2 = reinterpret_cast<const double2 *>(p), *w2 = reinterpret_cast<const double2 *>(w), *d2 = reinterpret_cast<const double2 *>(d);
    const hipx_int chunk = (n2 + (hipx_int)gridDim.x - 1) / (hipx_int)gridDim.x;
    const hipx_int c0    = (hipx_int)blockIdx.x * chunk;
    const hipx_int c1    = (c0 + chunk < n2) ? c0 + chunk : n2;
    auto step = [&](hipx_int q, double2 xv, double2 rv, const double2 pv, const double2 wv, const double2 dv) {
      double2 zv;
      if (UPX) {
        xv.x  = xv.x + a * pv.x;
        xv.y  = xv.y + a * pv.y;
        x2[q] = xv;
      }
      rv.x  = rv.x + ma * wv.x;
      rv.y  = rv.y + ma * wv.y;
      zv.x  = rv.x * dv.x;
      zv.y  = rv.y * dv.y;
      r2[q] = rv;
      if (!CONSTD || z) z2[q] = zv;  // constant diagonal + z == NULL: z is never stored (cg_aypx_axpy_kernel<.., ZR> re-forms it)
      acc[0].prod(zv.x, zv.x);
      acc[0].prod(zv.y, zv.y);
      acc[1].prod(zv.x, rv.x);
      acc[1].prod(zv.y, rv.y);
    };
    // U elements per stream in flight per thread (all loads of a round issued before the first use): 4 when only r and w are streamed
    // (no x update, constant diagonal: the launch-ahead CG's configuration -- 8 x 16-byte loads per thread, 128 KiB in flight per CU;
    // with 2 the kernel ran at 0.58 of the HBM peak where the five-stream AYPX kernel beside it reaches 0.75), 2 otherwise (register
    // budget of the 1024-thread workgroup).  The per-thread accumulation order is unchanged: q, q + kRedThreads, ...
    constexpr int  U  = (!UPX && CONSTD && WIDE) ? 4 : 2;  // (WIDE = false: the round-3 loop, kept behind HIPX_CG_FUSED_U2 for same-box A/B timing)
    const double2  z0 = {0.0, 0.0};
    const double2  dc = {dconst, dconst};
    if (sweep) {  // all workgroups walk the vector together, round by round: workgroup b takes elements [(it G + b) U T, ... + U T) in round it
      const hipx_int G = (hipx_int)gridDim.x, span = (hipx_int)U * kRedThreads;
      for (hipx_int base = (hipx_int)blockIdx.x * span; base < n2; base += G * span) {
        double2 ra[U], wa[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const hipx_int qu = base + u * (hipx_int)kRedThreads + threadIdx.x;
          const hipx_int qc = qu < n2 ? qu : 0;
          if (sweep >= 2) {  // developer variants: non-temporal loads of w (2) / of w and r (3)
            const double wx = __builtin_nontemporal_load(&w[2 * qc]), wy = __builtin_nontemporal_load(&w[2 * qc + 1]);
            wa[u] = {wx, wy};
            if (sweep >= 3) {
              const double rx = __builtin_nontemporal_load(&r[2 * qc]), ry = __builtin_nontemporal_load(&r[2 * qc + 1]);
              ra[u] = {rx, ry};
            } else ra[u] = r2[qc];
          } else {
            ra[u] = r2[qc];
            wa[u] = w2[qc];
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const hipx_int qu = base + u * (hipx_int)kRedThreads + threadIdx.x;
          if (qu < n2) step(qu, UPX ? x2[qu] : z0, ra[u], UPX ? p2[qu] : z0, wa[u], CONSTD ? dc : d2[qu]);
        }
      }
    } else {
    hipx_int q = c0 + (hipx_int)threadIdx.x;
    for (; q + (U - 1) * (hipx_int)kRedThreads < c1; q += U * kRedThreads) {
      double2 xa[U], ra[U], pa[U], wa[U], da[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const hipx_int qu = q + u * (hipx_int)kRedThreads;
        xa[u] = UPX ? x2[qu] : z0;
        ra[u] = r2[qu];
        pa[u] = UPX ? p2[qu] : z0;
        wa[u] = w2[qu];
        da[u] = CONSTD ? dc : d2[qu];
      }
#pragma unroll
      for (int u = 0; u < U; u++) step(q + u * (hipx_int)kRedThreads, xa[u], ra[u], pa[u], wa[u], da[u]);
    }
    for (; q < c1; q += kRedThreads) step(q, UPX ? x2[q] : z0, r2[q], UPX ? p2[q] : z0, w2[q], CONSTD ? dc : d2[q]);
    }
    if ((n & 1) && tid == 0) {
      hipx_int i  = n - 1;
      double   rv = r[i] + ma * w[i], zv = rv * (CONSTD ? dconst : d[i]);
      if (UPX) x[i] = x[i] + a * p[i];
      r[i] = rv;
      if (!CONSTD || z) z[i] = zv;
      acc[0].prod(zv, zv);
      acc[1].prod(zv, rv);
    }
  } else {
    for (hipx_int i = tid; i < n; i += T) {
      double rv = r[i] + ma * w[i], zv = rv * (CONSTD ? dconst : d[i]);
      if (UPX) x[i] = x[i] + a * p[i];
      r[i] = rv;
      if (!CONSTD || z) z[i] = zv;
      acc[0].prod(zv, zv);
      acc[1].prod(zv, rv);
    }
  }
  finish_sums<2, COMP>(acc, out);
}

// max / min with index: two small kernels (setup-time operations, not on the solver loop)
__global__ __launch_bounds__(kRedThreads) void minmax_stage1(const double *x, hipx_int n, int want_max, double *pv, hipx_int *pi)
{
  __shared__ double   sv[kRedThreads];
  __shared__ hipx_int si[kRedThreads];
  double              best = want_max ? -HUGE_VAL : HUGE_VAL;
  hipx_int            bi   = -1;
  const hipx_int      T    = (hipx_int)gridDim.x * kRedThreads;
  for (hipx_int i = (hipx_int)blockIdx.x * kRedThreads + threadIdx.x; i < n; i += T) {
    double v = x[i];
    if (bi < 0 || (want_max ? v > best : v < best)) {
      best = v;
      bi   = i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = kRedThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      double   ov = sv[threadIdx.x + s];
      hipx_int oi = si[threadIdx.x + s];
      bool     take = oi >= 0 && (si[threadIdx.x] < 0 || (want_max ? ov > sv[threadIdx.x] : ov < sv[threadIdx.x]) || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]));
      if (take) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pv[blockIdx.x] = sv[0];
    pi[blockIdx.x] = si[0];
  }
}

__global__ void replace_zeros_kernel(double *x, hipx_int n, double value, unsigned int *count)
{
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (hipx_int)gridDim.x * blockDim.x) {
    if (x[i] == 0.0) {
      x[i] = value;
      atomicAdd(count, 1u);
    }
  }
}

inline unsigned red_grid(hipx_int n)
{
  // enough 1024-thread workgroups to cover n at 8 elements/thread, capped at kRedBlocks (one per CU); a function of n only
  static const hipx_int cap = [] {
    const char *e = getenv("HIPX_RED_BLOCKS");
    const int   v = e ? atoi(e) : 256;
    return (hipx_int)((v >= 1 && v <= kRedBlocks) ? v : 256);
  }();
  hipx_int g = (n + kRedThreads * 8 - 1) / (kRedThreads * 8);
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

static bool    g_signal  = true;     // launch_*_nosignal() clear it around one dispatch ...
static double *g_results = nullptr;  // ... and redirect the kernel's result words to device memory
static double *g_dres    = nullptr;  // launch_dot(): device copy of the result beside the host slot
static inline RedOut red_out_g(int slot)
{
  RedOut o = red_out(slot, g_signal, g_dres);
  if (g_results) {
    o.results = g_results;
    o.pairs   = rt().red_exact;  // compensated mode: the fold over the ranks that follows takes unrounded (hi, lo) pairs
  }
  return o;
}

// every launch of the fused CG update kernel: x update or not, plain or compensated sums
template <bool DEVS, bool CONSTD>
static inline void cg_fused_go(unsigned g_in, hipStream_t st, double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, const double *dev_beta,
                               const double *dev_dpi, hipx_int n, bool vec, RedOut o, double dconst)
{
  unsigned          g  = g_in;
  (void)prof_section(HIPX_PROF_CG_UPDATE, true, st);
  static const bool u2 = getenv("HIPX_CG_FUSED_U2") != nullptr;
  // the launch-ahead configuration (no x update, constant diagonal) walks the vector with all workgroups together, round by round (measured
  // stand-alone on 256^3: 66.4 us = 6.06 TB/s against 72-74 us = 5.5 TB/s for one contiguous chunk per workgroup); HIPX_CG_FUSED_CHUNK keeps the chunks
  // (2: w -- written by the product kernel just before, read here once -- is loaded non-temporally: 88 -> 75 us inside the CG loop on 256^3, +5 % iterations/s;
  //  HIPX_CG_FUSED_NT = 1 | 2 | 3 selects plain loads / w / w and r)
  static const int  sweep = getenv("HIPX_CG_FUSED_CHUNK") ? 0 : (getenv("HIPX_CG_FUSED_NT") ? atoi(getenv("HIPX_CG_FUSED_NT")) : 2);
  static const int  gover = getenv("HIPX_CG_FUSED_BLOCKS") ? atoi(getenv("HIPX_CG_FUSED_BLOCKS")) : 0;  // developer switches (timing experiments)
  if (gover >= 1 && gover <= kRedBlocks) g = (unsigned)gover;
  if (rt().red_exact) {
    if (x) cg_fused_kernel<true, DEVS, CONSTD, true><<<g, kRedThreads, 0, st>>>(x, r, z, p, w, d, a, dev_beta, dev_dpi, n, vec, o, dconst);
    else cg_fused_kernel<false, DEVS, CONSTD, true><<<g, kRedThreads, 0, st>>>(x, r, z, p, w, d, a, dev_beta, dev_dpi, n, vec, o, dconst);
  } else {
    if (x) cg_fused_kernel<true, DEVS, CONSTD, false><<<g, kRedThreads, 0, st>>>(x, r, z, p, w, d, a, dev_beta, dev_dpi, n, vec, o, dconst);
    else if (u2) cg_fused_kernel<false, DEVS, CONSTD, false, false><<<g, kRedThreads, 0, st>>>(x, r, z, p, w, d, a, dev_beta, dev_dpi, n, vec, o, dconst);
    else cg_fused_kernel<false, DEVS, CONSTD, false><<<g, kRedThreads, 0, st>>>(x, r, z, p, w, d, a, dev_beta, dev_dpi, n, vec, o, dconst, sweep);
  }
  (void)prof_section(HIPX_PROF_CG_UPDATE, false, st);
}

template <int NV>
int launch_mdot(const double *x, const double *const *y, hipx_int n, int slot)
{
  MDotArgs<NV> a;
  bool         vec = aligned16(x) && n >= 2;
  for (int v = 0; v < NV; v++) {
    a.y[v] = y[v];
    vec    = vec && aligned16(y[v]);
  }
  if (rt().red_exact) mdot_kernel<NV, true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, a, n, vec, red_out_g(slot));
  else mdot_kernel<NV, false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, a, n, vec, red_out_g(slot));
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

template <int NVMAX>
int launch_mdot_wide(const double *x, int nv, const double *const *y, hipx_int n, int slot)
{
  MDotArgs<NVMAX> a;
  bool            vec = aligned16(x) && n >= 2;
  for (int v = 0; v < NVMAX; v++) {
    a.y[v] = y[v < nv ? v : 0];
    vec    = vec && aligned16(a.y[v]);
  }
  if constexpr (2 * NVMAX <= kMaxRedVals) {  // compensated sums take two partial rows each: batches of <= 16 (hipxVecMDot splits)
    if (rt().red_exact) {
      mdot_wide_kernel<NVMAX, true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, a, nv, n, vec, red_out_g(slot));
      HIPX_LAUNCH_CHECK();
      return HIPX_SUCCESS;
    }
  } else if (rt().red_exact) return fail(HIPX_ERR_ARG, "compensated mdot: at most 16 vectors per launch", __FILE__, __LINE__);
  mdot_wide_kernel<NVMAX, false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, a, nv, n, vec, red_out_g(slot));
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int mdot_dispatch(const double *x, int nv, const double *const *y, hipx_int n, int slot)
{
  if (nv > 8 && nv <= 16) return launch_mdot_wide<16>(x, nv, y, n, slot);
  if (nv > 16 && nv <= 32) return launch_mdot_wide<32>(x, nv, y, n, slot);
  switch (nv) {
  case 1: return launch_mdot<1>(x, y, n, slot);
  case 2: return launch_mdot<2>(x, y, n, slot);
  case 3: return launch_mdot<3>(x, y, n, slot);
  case 4: return launch_mdot<4>(x, y, n, slot);
  case 5: return launch_mdot<5>(x, y, n, slot);
  case 6: return launch_mdot<6>(x, y, n, slot);
  case 7: return launch_mdot<7>(x, y, n, slot);
  case 8: return launch_mdot<8>(x, y, n, slot);
  }
  return fail(HIPX_ERR_ARG, "mdot batch size", __FILE__, __LINE__);
}

template <int NV>
int launch_maxpy(double *y, const double *alpha, const double *const *x, double beta, int mode, hipx_int n)
{
  MaxpyArgs a;
  bool      vec = aligned16(y) && n >= 2;
  for (int j = 0; j < NV; j++) {
    a.x[j] = x[j];
    a.a[j] = alpha[j];
    vec    = vec && aligned16(x[j]);
  }
  maxpy_kernel<NV><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(y, a, beta, mode, n, vec);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int maxpy_dispatch(double *y, int nv, const double *alpha, const double *const *x, double beta, int mode, hipx_int n)
{
  switch (nv) {
  case 1: return launch_maxpy<1>(y, alpha, x, beta, mode, n);
  case 2: return launch_maxpy<2>(y, alpha, x, beta, mode, n);
  case 3: return launch_maxpy<3>(y, alpha, x, beta, mode, n);
  case 4: return launch_maxpy<4>(y, alpha, x, beta, mode, n);
  case 5: return launch_maxpy<5>(y, alpha, x, beta, mode, n);
  case 6: return launch_maxpy<6>(y, alpha, x, beta, mode, n);
  case 7: return launch_maxpy<7>(y, alpha, x, beta, mode, n);
  case 8: return launch_maxpy<8>(y, alpha, x, beta, mode, n);
  }
  return fail(HIPX_ERR_ARG, "maxpy batch size", __FILE__, __LINE__);
}

__global__ void red_signal_kernel(unsigned long long *flag, unsigned long long seq, double *results, const double *src, int nvals, double *dres)
{
  for (int v = 0; v < nvals; v++) {
    results[v] = src[v];
    if (dres) dres[v] = src[v];  // device copy for the kernels queued behind (launch-ahead CG on several ranks)
  }
  __threadfence_system();
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

int hipx::launch_mdot_nosignal(const double *x, int nv, const double *const *y, hipx_int n, int slot, double *dev_results)
{
  g_signal  = false;
  g_results = dev_results;
  int ierr  = mdot_dispatch(x, nv, y, n, slot);
  g_signal  = true;
  g_results = nullptr;
  return ierr;
}

int hipx::red_signal(int slot, const double *dev_results, int nvals, double *dres)
{
  Runtime &r = rt();
  red_signal_kernel<<<1, 1, 0, r.compute>>>(r.d_flags + slot, ++r.seq[slot], slot_results_dev(slot), dev_results, nvals, dres);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

static int launch_cg_fused(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, int slot)
{
  bool vec = aligned16(x) && aligned16(r) && aligned16(z) && aligned16(p) && aligned16(w) && aligned16(d) && n >= 2;
  cg_fused_go<false, false>(red_grid(n), rt().compute, x, r, z, p, w, d, a, nullptr, nullptr, n, vec, red_out_g(slot), 0.0);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// the launch-ahead update kernel (scalars read from device memory) with its two sums left in device memory, no host signal:
// an all-reduce follows on the stream (hipx_comm.hip)
int hipx::launch_cg_fused_dev_nosignal(double *x, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi,
                                       hipx_int n, int slot, double *dev_results)
{
  bool vec = aligned16(x) && aligned16(r) && aligned16(z) && aligned16(p) && aligned16(w) && aligned16(d) && n >= 2;
  const unsigned g = red_grid(n);
  RedOut         o = red_out(slot, false, nullptr);
  o.results        = dev_results;
  o.pairs          = rt().red_exact;  // (the all-reduce that follows folds the ranks' pairs)
  hipStream_t st   = rt().compute;
  if (d) cg_fused_go<true, false>(g, st, x, r, z, p, w, d, 0.0, dev_beta, dev_dpi, n, vec, o, 0.0);
  else cg_fused_go<true, true>(g, st, x, r, z, p, w, d, 0.0, dev_beta, dev_dpi, n, vec, o, dconst);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipx::launch_dot(const double *x, const double *y, hipx_int n, int slot, double *dres)
{
  const double *ys[1] = {y};
  g_dres              = dres;
  int ierr            = mdot_dispatch(x, 1, ys, n, slot);
  g_dres              = nullptr;
  return ierr;
}

int hipx::launch_sum(const double *x, hipx_int n, int slot, double *dres)
{
  (void)prof_section(HIPX_PROF_FOLD, true, rt().compute);
  if (rt().red_exact) sum_kernel<true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, red_out(slot, true, dres));
  else sum_kernel<false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, red_out(slot, true, dres));
  (void)prof_section(HIPX_PROF_FOLD, false, rt().compute);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipx::launch_cg_fused_nosignal(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, int slot, double *dev_results)
{
  g_signal  = false;
  g_results = dev_results;
  int ierr  = launch_cg_fused(x, r, z, p, w, d, a, n, slot);
  g_signal  = true;
  g_results = nullptr;
  return ierr;
}

// One Chebyshev iteration's vector work in one pass (cheby.c:475-511 with PCJACOBI / PCNONE and no norm requested):
//   r = b - A p_k            VecAYPX(r, -1, b), dvec2.c:767           (Ap = A p_k comes from the SpMV before)
//   z = r * dinv  |  z = r   PCApply_Jacobi = VecPointwiseMult, jacobi.c:301  |  PCNONE
//   p_next = alpha p_prev + beta p_k + gamma z   VecAXPBYPCZ_Seq, bvec1.c:120-147: its four association orders (BR)
// Every element sees the same operations in the same order as the three reference loops: p_next is bit-identical; 5 vector reads
// and 1 write instead of 7 and 3, one launch instead of three.
namespace {
template <int BR, bool JAC, bool ROUT>
__global__ __launch_bounds__(256) void cheby_step_kernel(double *pn, double a, double b, double c, const double *__restrict__ pp, const double *__restrict__ pc,
                                                         const double *__restrict__ dinv, const double *__restrict__ rhs, const double *Ap, double *__restrict__ rout, hipx_int n)  // (Ap may be pn: in place)
{
  for (hipx_int i = (hipx_int)blockIdx.x * 256 + threadIdx.x; i < n; i += (hipx_int)gridDim.x * 256) {
    const double r = rhs[i] - Ap[i];
    const double z = JAC ? r * dinv[i] : r;
    const double x = pp[i], y = pc[i];
    double       o;
    if (BR == 0) o = x + b * y + c * z;
    else if (BR == 1) o = a * x + b * y + z;
    else if (BR == 2) o = a * x + b * y;
    else o = a * x + b * y + c * z;
    pn[i] = o;
    if (ROUT) rout[i] = r;
  }
}
}  // namespace

extern "C" {

int hipxVecSet(double *x, hipx_int n, double alpha)
{
  HIPX_CHECK_INIT();
  if (n <= 0) return HIPX_SUCCESS;
  if (alpha == 0.0) {  // dvec2.c:649 PetscArrayzero
    HIPX_HIP(hipMemsetAsync(x, 0, (size_t)n * sizeof(double), rt().compute));
    return HIPX_SUCCESS;
  }
  return launch_ew1(x, n, FSet{alpha});
}

int hipxVecCopy(const double *x, double *y, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (n <= 0 || x == y) return HIPX_SUCCESS;
  HIPX_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, rt().compute));
  return HIPX_SUCCESS;
}

int hipxVecScale(double *x, hipx_int n, double alpha)
{
  HIPX_CHECK_INIT();
  if (alpha == 0.0) return hipxVecSet(x, n, 0.0);  // bvec2.c:172
  if (alpha == 1.0) return HIPX_SUCCESS;           // bvec2.c:174
  return launch_ew1(x, n, FScale{alpha});
}

int hipxVecShift(double *x, hipx_int n, double shift)
{
  HIPX_CHECK_INIT();
  if (shift == 0.0) return HIPX_SUCCESS;
  return launch_ew1(x, n, FShift{shift});
}

int hipxVecReciprocal(double *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  return launch_ew1(x, n, FRecip{});
}

int hipxVecAbs(double *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  return launch_ew1(x, n, FAbs{});
}

int hipxVecSwap(double *x, double *y, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (n <= 0 || x == y) return HIPX_SUCCESS;
  hipx_int g = (n + kEwThreads - 1) / kEwThreads;
  if (g > kEwMaxBlocks) g = kEwMaxBlocks;
  swap_kernel<<<(unsigned)g, kEwThreads, 0, rt().compute>>>(x, y, n);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxVecAXPY(double *y, double alpha, const double *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (alpha == 0.0) return HIPX_SUCCESS;  // bvec1.c:75
  return launch_ew2(y, x, n, FAxpy{alpha});
}

int hipxVecAYPX(double *y, double beta, const double *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (beta == 0.0) return hipxVecCopy(x, y, n);               // dvec2.c:756
  if (beta == 1.0) return launch_ew2(y, x, n, FAxpy{beta});   // dvec2.c:758 -> VecAXPY_Seq(y, 1, x)
  if (beta == -1.0) return launch_ew2(y, x, n, FXmy{});       // dvec2.c:767
  return launch_ew2(y, x, n, FAypx{beta});                    // dvec2.c:774
}

int hipxVecAXPBY(double *y, double a, double b, const double *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (a == 0.0) return hipxVecScale(y, n, b);    // bvec1.c:94
  if (b == 1.0) return hipxVecAXPY(y, a, x, n);  // bvec1.c:96
  if (a == 1.0) return hipxVecAYPX(y, b, x, n);  // bvec1.c:98
  if (b == 0.0) return launch_ew2(y, x, n, FAx{a});
  return launch_ew2(y, x, n, FAxpby{a, b});
}

int hipxVecWAXPY(double *w, double alpha, const double *x, const double *y, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (alpha == 1.0) return launch_ew3(w, x, y, n, FWadd{});   // dvec2.c:805
  if (alpha == -1.0) return launch_ew3(w, x, y, n, FWsub{});  // dvec2.c:808
  if (alpha == 0.0) return hipxVecCopy(y, w, n);              // dvec2.c:810
  return launch_ew3(w, x, y, n, FWaxpy{alpha});
}

int hipxVecAXPBYPCZ(double *z, double alpha, double beta, double gamma, const double *x, const double *y, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (alpha == 1.0) return launch_ew3(z, x, y, n, FAbc0{beta, gamma});
  if (gamma == 1.0) return launch_ew3(z, x, y, n, FAbc1{alpha, beta});
  if (gamma == 0.0) return launch_ew3(z, x, y, n, FAbc2{alpha, beta});
  return launch_ew3(z, x, y, n, FAbc3{alpha, beta, gamma});
}

int hipxVecChebyshevStep(double *pnext, double alpha, double beta, double gamma, const double *pprev, const double *pcur, const double *dinv, const double *b, const double *Ap, double *r_out,
                         hipx_int n)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(n >= 0 && (n == 0 || (pnext && pprev && pcur && b && Ap)), "null argument");
  if (n <= 0) return HIPX_SUCCESS;
  const int      br = (alpha == 1.0) ? 0 : ((gamma == 1.0) ? 1 : ((gamma == 0.0) ? 2 : 3));
  const unsigned g  = (unsigned)std::min<hipx_int>((n + 255) / 256, 16384);
  hipStream_t    st = rt().compute;
#define HIPX_CHEB(BRV) \
  do { \
    if (dinv && r_out) cheby_step_kernel<BRV, true, true><<<g, 256, 0, st>>>(pnext, alpha, beta, gamma, pprev, pcur, dinv, b, Ap, r_out, n); \
    else if (dinv) cheby_step_kernel<BRV, true, false><<<g, 256, 0, st>>>(pnext, alpha, beta, gamma, pprev, pcur, dinv, b, Ap, r_out, n); \
    else if (r_out) cheby_step_kernel<BRV, false, true><<<g, 256, 0, st>>>(pnext, alpha, beta, gamma, pprev, pcur, dinv, b, Ap, r_out, n); \
    else cheby_step_kernel<BRV, false, false><<<g, 256, 0, st>>>(pnext, alpha, beta, gamma, pprev, pcur, dinv, b, Ap, r_out, n); \
  } while (0)
  switch (br) {
  case 0: HIPX_CHEB(0); break;
  case 1: HIPX_CHEB(1); break;
  case 2: HIPX_CHEB(2); break;
  default: HIPX_CHEB(3); break;
  }
#undef HIPX_CHEB
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxVecPointwiseMult(double *w, const double *x, const double *y, hipx_int n)
{
  HIPX_CHECK_INIT();
  return launch_ew3(w, x, y, n, FPmult{});
}

int hipxVecPointwiseMultDotsBegin(double *w, const double *x, const double *y, hipx_int n, int slot)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "reduction slot out of range (the last two are reserved)");
  HIPX_ARG(w != x && w != y, "PointwiseMultDots: w must not alias an operand (the sums are those of the NEW w with the OLD x)");
  if (n <= 0) {
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    slot_results_host(slot)[0] = slot_results_host(slot)[1] = 0.0;
    rt().h_flags[slot]                                     = ++rt().seq[slot];
    return HIPX_SUCCESS;
  }
  const bool vec = aligned16(w) && aligned16(x) && aligned16(y) && n >= 2;
  if (rt().red_exact) pwmult_dots_kernel<true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(w, x, y, n, vec, red_out(slot));
  else pwmult_dots_kernel<false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(w, x, y, n, vec, red_out(slot));
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxVecPointwiseDivide(double *w, const double *x, const double *y, hipx_int n)
{
  HIPX_CHECK_INIT();
  return launch_ew3(w, x, y, n, FPdiv{});
}

int hipxVecReplaceZeros(double *x, hipx_int n, double value, hipx_int *nreplaced_host)
{
  HIPX_CHECK_INIT();
  unsigned int *cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1);  // last slot is reserved for this counter
  HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), rt().compute));
  if (n > 0) {
    hipx_int g = (n + 255) / 256;
    if (g > kEwMaxBlocks) g = kEwMaxBlocks;
    replace_zeros_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(x, n, value, cnt);
    HIPX_LAUNCH_CHECK();
  }
  unsigned int h = 0;
  HIPX_HIP(hipMemcpyAsync(&h, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), rt().compute));
  if (nreplaced_host) *nreplaced_host = (hipx_int)h;
  return HIPX_SUCCESS;
}

// y (= beta y | = 0) += sum_j alpha[j] x[j] in one pass when everything is 16-byte aligned and nv fits; mode as in maxpy_kernel
static int maxpy_wide(double *y, hipx_int nv, const double *alpha, const double *const *x, double beta, int mode, hipx_int n, bool *done_out)
{
  *done_out = false;
  static const bool off = getenv("HIPX_MAXPY_BATCH") && atoi(getenv("HIPX_MAXPY_BATCH")) == 8;  // the round-1 batching (identical results)
  if (off || nv <= MAXPY_B || nv > MAXPY_WIDE || n < 2 || !aligned16(y)) return HIPX_SUCCESS;
  MaxpyWideArgs a;
  for (int j = 0; j < MAXPY_WIDE; j++) {
    a.x[j] = x[j < nv ? j : 0];
    a.a[j] = j < nv ? alpha[j] : 0.0;
    if (!aligned16(a.x[j])) return HIPX_SUCCESS;
  }
  const unsigned g = (unsigned)std::min<hipx_int>(((n >> 1) + kEwThreads - 1) / kEwThreads, 8192);
  switch (nv & 3) {
  case 0: maxpy_wide_kernel<0><<<g, kEwThreads, 0, rt().compute>>>(y, a, (int)nv, beta, mode, n); break;
  case 1: maxpy_wide_kernel<1><<<g, kEwThreads, 0, rt().compute>>>(y, a, (int)nv, beta, mode, n); break;
  case 2: maxpy_wide_kernel<2><<<g, kEwThreads, 0, rt().compute>>>(y, a, (int)nv, beta, mode, n); break;
  default: maxpy_wide_kernel<3><<<g, kEwThreads, 0, rt().compute>>>(y, a, (int)nv, beta, mode, n); break;
  }
  HIPX_LAUNCH_CHECK();
  *done_out = true;
  return HIPX_SUCCESS;
}

int hipxVecMAXPY(double *y, hipx_int nv, const double *alpha, const double *const *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(nv >= 0, "nv < 0");
  if (n <= 0 || nv == 0) return HIPX_SUCCESS;
  {
    bool done1 = false;
    int  ierr  = maxpy_wide(y, nv, alpha, x, 0.0, 0, n, &done1);
    if (ierr || done1) return ierr;
  }
  // the reference processes (nv & 3) vectors first, then groups of four (dvec2.c:672-690); batches here keep
  // that grouping: first batch takes (nv & 3) + 4 (or fewer), later batches multiples of 4.
  hipx_int done = 0, rem = nv & 3;
  while (done < nv) {
    hipx_int left = nv - done, take;
    if (left <= MAXPY_B) take = left;
    else take = (done == 0 && rem) ? rem + 4 : MAXPY_B;
    int ierr = maxpy_dispatch(y, take, alpha + done, x + done, 0.0, 0, n);
    if (ierr) return ierr;
    done += take;
  }
  return HIPX_SUCCESS;
}

int hipxVecMAXPBY(double *y, hipx_int nv, const double *alpha, double beta, const double *const *x, hipx_int n)
{
  HIPX_CHECK_INIT();
  // rvector.c:1394-1440 with no ops->maxpby: beta == 0 -> VecSet(y, 0) else VecScale(y, beta); then VecMAXPY.
  int ierr;
  if (n > 0 && nv > 0) {  // one pass: y = beta y (or 0) and the sums, same operations in the same order as the two-step form
    bool done1 = false;
    ierr       = maxpy_wide(y, nv, alpha, x, beta, beta == 0.0 ? 2 : 1, n, &done1);
    if (ierr || done1) return ierr;
  }
  if (beta == 0.0) ierr = hipxVecSet(y, n, 0.0);
  else ierr = hipxVecScale(y, n, beta);
  if (ierr) return ierr;
  return hipxVecMAXPY(y, nv, alpha, x, n);
}

int hipxVecDotBegin(const double *x, const double *y, hipx_int n, int slot)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "reduction slot out of range (the last two are reserved)");
  if (n <= 0) {
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    slot_results_host(slot)[0] = 0.0;
    rt().h_flags[slot]         = ++rt().seq[slot];
    return HIPX_SUCCESS;
  }
  const double *ys[1] = {y};
  return mdot_dispatch(x, 1, ys, n, slot);
}

int hipxRedEnd(int slot, int nvals, double *results)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && nvals >= 0 && nvals <= kMaxRedVals, "reduction slot / count out of range");
  return red_wait(slot, nvals, results);
}

int hipxVecDot(const double *x, const double *y, hipx_int n, double *result)
{
  HIPX_CHECK_INIT();
  if (n <= 0) {
    *result = 0.0;
    return HIPX_SUCCESS;
  }
  int ierr = hipxVecDotBegin(x, y, n, 0);
  if (ierr) return ierr;
  return red_wait(0, 1, result);
}

int hipxVecMDot(const double *x, hipx_int nv, const double *const *y, hipx_int n, double *results)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(nv >= 0, "nv < 0");
  if (n <= 0) {
    for (hipx_int j = 0; j < nv; j++) results[j] = 0.0;
    return HIPX_SUCCESS;
  }
  // batches of <= 32 vectors (one pass over x each; GMRES(30) is ONE launch), each in its own slot so a single wait at the end
  // suffices.  HIPX_MDOT_BATCH=8 restores the round-1 batching (the sums are identical either way: same addition order).
  static const hipx_int BATCH0 = (getenv("HIPX_MDOT_BATCH") && atoi(getenv("HIPX_MDOT_BATCH")) == 8) ? 8 : 32;
  const hipx_int        BATCH  = (rt().red_exact && BATCH0 > 16) ? 16 : BATCH0;  // compensated sums: two partial rows each
  hipx_int done = 0;
  int      slot = 1;
  while (done < nv) {
    hipx_int take = nv - done > BATCH ? BATCH : nv - done;
    int      ierr = mdot_dispatch(x, take, y + done, n, slot);
    if (ierr) return ierr;
    done += take;
    slot++;
  }
  done = 0;
  slot = 1;
  while (done < nv) {
    hipx_int take = nv - done > BATCH ? BATCH : nv - done;
    int      ierr = red_wait(slot, take, results + done);
    if (ierr) return ierr;
    done += take;
    slot++;
  }
  return HIPX_SUCCESS;
}

int hipxVecNorm(const double *x, hipx_int n, int type, double *results)
{
  HIPX_CHECK_INIT();
  results[0] = 0.0;
  if (type == 4) results[1] = 0.0;
  if (n <= 0) return HIPX_SUCCESS;
  const int slot = 0;
  if (type == 1 || type == 2) {  // NORM_2 / FROBENIUS: sqrt(x.x), bvec2.c:204
    const double *ys[1] = {x};
    int           ierr  = mdot_dispatch(x, 1, ys, n, slot);
    if (ierr) return ierr;
    double s;
    ierr = red_wait(slot, 1, &s);
    if (ierr) return ierr;
    results[0] = sqrt(s);
  } else if (type == 0 || type == 4) {
    if (rt().red_exact) norm12_kernel<true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, aligned16(x) && n >= 2, red_out(slot));
    else norm12_kernel<false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, aligned16(x) && n >= 2, red_out(slot));
    HIPX_LAUNCH_CHECK();
    double s[2];
    int    ierr = red_wait(slot, 2, s);
    if (ierr) return ierr;
    results[0] = s[0];
    if (type == 4) results[1] = sqrt(s[1]);
  } else if (type == 3) {
    norminf_kernel<<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, red_out(slot));
    HIPX_LAUNCH_CHECK();
    return red_wait(slot, 1, results);
  } else return fail(HIPX_ERR_ARG, "unknown NormType", __FILE__, __LINE__);
  return HIPX_SUCCESS;
}

int hipxVecDotNorm2(const double *x, const double *y, hipx_int n, double *dp, double *nm)
{
  HIPX_CHECK_INIT();
  *dp = *nm = 0.0;
  if (n <= 0) return HIPX_SUCCESS;
  if (rt().red_exact) dotnorm2_kernel<true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, y, n, red_out(0));
  else dotnorm2_kernel<false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, y, n, red_out(0));
  HIPX_LAUNCH_CHECK();
  double s[2];
  int    ierr = red_wait(0, 2, s);
  if (ierr) return ierr;
  *dp = s[0];
  *nm = s[1];
  return HIPX_SUCCESS;
}

int hipxVecSum(const double *x, hipx_int n, double *result)
{
  HIPX_CHECK_INIT();
  *result = 0.0;
  if (n <= 0) return HIPX_SUCCESS;
  if (rt().red_exact) sum_kernel<true><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, red_out(0));
  else sum_kernel<false><<<red_grid(n), kRedThreads, 0, rt().compute>>>(x, n, red_out(0));
  HIPX_LAUNCH_CHECK();
  return red_wait(0, 1, result);
}

static int minmax(const double *x, hipx_int n, int want_max, hipx_int *idx, double *result)
{
  HIPX_CHECK_INIT();
  if (n <= 0) {  // dvec2.c:592-640: empty vector -> idx -1, +-PETSC_MAX_REAL-like sentinel
    if (idx) *idx = -1;
    *result = want_max ? -1.7976931348623157e308 : 1.7976931348623157e308;
    return HIPX_SUCCESS;
  }
  const unsigned g  = red_grid(n);
  double        *pv = slot_partials(0);
  hipx_int      *pi = reinterpret_cast<hipx_int *>(slot_partials(1));
  minmax_stage1<<<g, kRedThreads, 0, rt().compute>>>(x, n, want_max, pv, pi);
  HIPX_LAUNCH_CHECK();
  static double   hv[kRedBlocks];
  static hipx_int hi[kRedBlocks];
  HIPX_HIP(hipMemcpyAsync(hv, pv, g * sizeof(double), hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipMemcpyAsync(hi, pi, g * sizeof(hipx_int), hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  double   best = hv[0];
  hipx_int bi   = hi[0];
  for (unsigned b = 1; b < g; b++) {
    if (hi[b] < 0) continue;
    if ((want_max ? hv[b] > best : hv[b] < best) || (hv[b] == best && hi[b] < bi)) {
      best = hv[b];
      bi   = hi[b];
    }
  }
  *result = best;
  if (idx) *idx = bi;
  return HIPX_SUCCESS;
}
int hipxVecMax(const double *x, hipx_int n, hipx_int *idx, double *result) { return minmax(x, n, 1, idx, result); }
int hipxVecMin(const double *x, hipx_int n, hipx_int *idx, double *result) { return minmax(x, n, 0, idx, result); }

int hipxCGFusedUpdate(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, double *sums2)
{
  HIPX_CHECK_INIT();
  sums2[0] = sums2[1] = 0.0;
  if (n <= 0) return HIPX_SUCCESS;
  int ierr = launch_cg_fused(x, r, z, p, w, d, a, n, 0);
  if (ierr) return ierr;
  return red_wait(0, 2, sums2);
}

int hipxVecAXPYPointwiseMultDotsBegin(double *y, double alpha, const double *x, double *w, const double *d, double dconst, hipx_int n, int slot)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "reduction slot out of range");
  HIPX_ARG(n <= 0 || (y && x && w && y != x && w != y && w != d && y != d), "null or aliased argument");  // (w == x is fine: an element is read before it is written, by the same thread)
  if (n <= 0) return HIPX_SUCCESS;
  if (d) return launch_cg_fused(nullptr, y, w, nullptr, x, d, -alpha, n, slot);  // (the kernel forms r + (-a) w with a = -alpha: the multiplier is alpha itself)
  const bool vec = aligned16(y) && aligned16(w) && aligned16(x) && n >= 2;
  cg_fused_go<false, true>(red_grid(n), rt().compute, nullptr, y, w, nullptr, x, nullptr, -alpha, nullptr, nullptr, n, vec, red_out_g(slot), dconst);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxCGAypxAxpy(double *p, double b, const double *z, double *x, double a, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (n <= 0) return HIPX_SUCCESS;
  if (b == 0.0 || b == 1.0 || b == -1.0 || a == 0.0) {  // the reference's special-cased loops: keep them literally
    int ierr = hipxVecAXPY(x, a, p, n);
    if (ierr) return ierr;
    return hipxVecAYPX(p, b, z, n);
  }
  bool vec = aligned16(p) && aligned16(x) && aligned16(z) && n >= 2;
  (void)prof_section(HIPX_PROF_CG_DIR, true, rt().compute);
  cg_aypx_axpy_kernel<false, false, true><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(p, x, z, 0.0, b, a, nullptr, nullptr, nullptr, n, vec);
  (void)prof_section(HIPX_PROF_CG_DIR, false, rt().compute);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxCGAypxAxpyR(double *p, double b, const double *r, double dconst, double *x, double a, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (n <= 0) return HIPX_SUCCESS;
  bool vec = aligned16(p) && aligned16(x) && aligned16(r) && n >= 2;
  (void)prof_section(HIPX_PROF_CG_DIR, true, rt().compute);
  if (x) cg_aypx_axpy_kernel<false, true, true><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(p, x, r, dconst, b, a, nullptr, nullptr, nullptr, n, vec);
  else cg_aypx_axpy_kernel<false, true, false><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(p, x, r, dconst, b, a, nullptr, nullptr, nullptr, n, vec);
  (void)prof_section(HIPX_PROF_CG_DIR, false, rt().compute);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxCGAypxAxpyDev(double *p, const double *z, const double *r, double dconst, double *x, const double *dev_beta_new, const double *dev_beta_old, const double *dev_dpi, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (n <= 0) return HIPX_SUCCESS;
  HIPX_ARG((z || r) && x && dev_beta_new && dev_beta_old && dev_dpi, "null argument");
  const double *src = z ? z : r;
  bool          vec = aligned16(p) && aligned16(x) && aligned16(src) && n >= 2;
  (void)prof_section(HIPX_PROF_CG_DIR, true, rt().compute);
  if (z) cg_aypx_axpy_kernel<true, false, true><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(p, x, src, 0.0, 0.0, 0.0, dev_beta_new, dev_beta_old, dev_dpi, n, vec);
  else cg_aypx_axpy_kernel<true, true, true><<<ew_grid(n, vec), kEwThreads, 0, rt().compute>>>(p, x, src, dconst, 0.0, 0.0, dev_beta_new, dev_beta_old, dev_dpi, n, vec);
  (void)prof_section(HIPX_PROF_CG_DIR, false, rt().compute);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxCGSingleReductionUpdate(double *p, double *w, double *x, double *r, double *z, const double *s, const double *d, double b, double a, hipx_int n)
{
  HIPX_CHECK_INIT();
  if (n <= 0) return HIPX_SUCCESS;
  HIPX_ARG(p && w && x && r && z && s, "null argument");
  hipx_int g = (n + kEwThreads * 2 - 1) / (kEwThreads * 2);
  if (g > 8192) g = 8192;
  cg_sr_update_kernel<<<(unsigned)(g < 1 ? 1 : g), kEwThreads, 0, rt().compute>>>(p, w, x, r, z, s, d, b, a, n);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxCGSingleReductionUpdateDev(double *p, double *w, double *x, double *r, double *z, const double *s, const double *d, const double *dev_sums3, const double *dev_state_old,
                                   double *dev_state_new, hipx_int n)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(dev_sums3 && dev_state_old && dev_state_new && dev_state_old != dev_state_new, "null / aliased scalar blocks");
  HIPX_ARG(n <= 0 || (p && w && x && r && z && s), "null argument");
  hipx_int g = (n + kEwThreads * 2 - 1) / (kEwThreads * 2);
  if (g > 8192) g = 8192;
  cg_sr_update_dev_kernel<<<(unsigned)(g < 1 ? 1 : g), kEwThreads, 0, rt().compute>>>(p, w, x, r, z, s, d, dev_sums3, dev_state_old, dev_state_new, n);  // (n == 0: the scalar block still advances)
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// x . y_j, j < nv <= 16, enqueued: hipxRedEnd(slot, nv, ...) collects the sums, dev_results receives a device copy for kernels queued behind
int hipxVecMDotBegin(const double *x, hipx_int nv, const double *const *y, hipx_int n, int slot, double *dev_results)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(nv >= 1 && nv <= 16 && slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "1 <= nv <= 16, slot in range");
  HIPX_ARG(n > 0, "hipxVecMDotBegin needs a non-empty vector");
  g_dres   = dev_results;
  int ierr = mdot_dispatch(x, nv, y, n, slot);
  g_dres   = nullptr;
  return ierr;
}

int hipxCGFusedUpdateBegin(double *x, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi,
                           hipx_int n, int slot, double *dev_sums2)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2 && n > 0 && dev_beta && dev_dpi, "bad slot / empty vector / null scalars");
  HIPX_ARG(z || !d, "z may only be omitted with a constant diagonal (d == NULL)");
  bool vec = aligned16(x) && aligned16(r) && aligned16(z) && aligned16(p) && aligned16(w) && aligned16(d) && n >= 2;
  const unsigned g = red_grid(n);
  RedOut         o = red_out(slot, true, dev_sums2);
  hipStream_t    st = rt().compute;
  if (d) cg_fused_go<true, false>(g, st, x, r, z, p, w, d, 0.0, dev_beta, dev_dpi, n, vec, o, 0.0);
  else cg_fused_go<true, true>(g, st, x, r, z, p, w, d, 0.0, dev_beta, dev_dpi, n, vec, o, dconst);  // d == NULL: constant Jacobi diagonal
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

}  // extern "C"

// ---- indexed gather / scatter (PetscSF pack / unpack on the device: sfpack.c Pack / UnpackAndInsert / UnpackAndAdd for unit = one
// scalar).  dst[didx ? didx[k] : k] (= | +=) src[sidx ? sidx[k] : k].  mode 1 (add) requires didx without duplicates inside one
// call (the caller checks at set-up): then the result does not depend on the thread schedule -- bit-identical to the sequential loop.
namespace {
__global__ __launch_bounds__(256) void scatter_indexed_kernel(const double *__restrict__ src, const hipx_int *__restrict__ sidx, double *dst, const hipx_int *__restrict__ didx, hipx_int n,
                                                              int mode)
{
  for (hipx_int k = (hipx_int)blockIdx.x * 256 + threadIdx.x; k < n; k += (hipx_int)gridDim.x * 256) {
    const double   v = src[sidx ? sidx[k] : k];
    const hipx_int d = didx ? didx[k] : k;
    if (mode == 0) dst[d] = v;
    else dst[d] = dst[d] + v;
  }
}
}  // namespace

extern "C" int hipxVecScatterIndexed(const double *src, const hipx_int *sidx, double *dst, const hipx_int *didx, hipx_int n, int mode)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(n >= 0 && (mode == 0 || mode == 1), "bad arguments");
  if (!n) return HIPX_SUCCESS;
  HIPX_ARG(src && dst, "null vector");
  hipx_int g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  scatter_indexed_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(src, sidx, dst, didx, n, mode);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// ---- PCApply_PBJacobi / PCApplyTranspose_PBJacobi (src/ksp/pc/impls/pbjacobi/pbjacobi.c:4-124,126-241): y_i = D_i^{-1} x_i with the
// inverted bs x bs diagonal blocks stored column-major (MatInvertBlockDiagonal).  One thread per row of a block; the products are
// added left to right as the reference's unrolled expressions do (bs <= 7: the sum starts with the first product; larger blocks:
// from 0), each product and sum rounded separately -> bit-identical.
namespace {
template <bool TR>
__global__ __launch_bounds__(256) void pbjacobi_kernel(const double *__restrict__ diag, hipx_int bs, hipx_int n, const double *__restrict__ x, double *__restrict__ y)
{
  for (hipx_int r = (hipx_int)blockIdx.x * 256 + threadIdx.x; r < n; r += (hipx_int)gridDim.x * 256) {
    const hipx_int i = r / bs, ib = r - i * bs;
    const double  *d = diag + (size_t)i * bs * bs;
    const double  *xx = x + (size_t)i * bs;
    double         s;
    if (bs <= 7) {
      s = (TR ? d[ib * bs] : d[ib]) * xx[0];
      for (hipx_int jb = 1; jb < bs; jb++) s = s + (TR ? d[ib * bs + jb] : d[ib + jb * bs]) * xx[jb];
    } else {
      s = 0.0;
      for (hipx_int jb = 0; jb < bs; jb++) s += (TR ? d[ib * bs + jb] : d[ib + jb * bs]) * xx[jb];
    }
    y[r] = s;
  }
}
}  // namespace

extern "C" int hipxPCPBJacobiApply(const double *diag, hipx_int bs, hipx_int mbs, const double *x, double *y, int transpose)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(bs >= 1 && mbs >= 0, "bad block size");
  const long long n = (long long)bs * mbs;
  HIPX_ARG(n <= 0x7fffffffLL, "vector too long");
  if (!n) return HIPX_SUCCESS;
  HIPX_ARG(diag && x && y && x != y, "null / aliased argument");
  hipx_int g = (hipx_int)((n + 255) / 256);
  if (g > 8192) g = 8192;
  if (transpose) pbjacobi_kernel<true><<<(unsigned)g, 256, 0, rt().compute>>>(diag, bs, (hipx_int)n, x, y);
  else pbjacobi_kernel<false><<<(unsigned)g, 256, 0, rt().compute>>>(diag, bs, (hipx_int)n, x, y);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}
