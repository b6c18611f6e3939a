// hipx_sor.hip -- MatSOR_SeqAIJ replacement (aij.c:1797-2007).  Placeholder until the level-scheduled
// kernels land: fails loudly (no CPU fallback).
#include "hipx_internal.h"
using namespace hipx;
extern "C" int hipxMatSOR(hipxMat A, const double *b, double omega, int flag, double shift, hipx_int its, hipx_int lits, double *x)
{
  HIPX_CHECK_INIT();
  (void)A; (void)b; (void)omega; (void)flag; (void)shift; (void)its; (void)lits; (void)x;
  return fail(HIPX_ERR_SUP, "hipxMatSOR: not implemented yet", __FILE__, __LINE__);
}
