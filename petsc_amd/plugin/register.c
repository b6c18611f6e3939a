/*
 * register.c -- entry symbol of libpetschipx.so.  PetscDLLibraryOpen (src/sys/dll/dl.c:139-205) strips "lib" and ".so"
 * from the file name and calls PetscDLLibraryRegister_<base>(); it is reached with
 *     ./app -dll_prepend /path/libpetschipx.so -vec_type hipx -mat_type aijhipx [-pc_type jacobihipx]
 * (src/sys/dll/reg.c:79,150) or by calling the function directly after PetscInitialize() when the library is linked.
 */
#include "hipxplugin.h"
#include <petscsf.h>

/* the entry symbol is derived from the file name; the MPICH flavour of the plugin is built as libpetschipx_mpich.so */
#if !defined(HIPX_PLUGIN_REGISTER)
  #define HIPX_PLUGIN_REGISTER PetscDLLibraryRegister_petschipx
#endif

PETSC_EXTERN PetscErrorCode HIPX_PLUGIN_REGISTER(void)
{
  PetscFunctionBegin;
  PetscCall(VecRegister(VECSEQHIPX, VecCreate_SeqHIPX));
  PetscCall(VecRegister(VECMPIHIPX, VecCreate_MPIHIPX));
  PetscCall(VecRegister(VECHIPX, VecCreate_HIPX));
  PetscCall(MatRegister(MATSEQAIJHIPX, MatCreate_SeqAIJHIPX));
  PetscCall(MatRegister(MATMPIAIJHIPX, MatCreate_MPIAIJHIPX));
  PetscCall(MatRegisterRootName(MATAIJHIPX, MATSEQAIJHIPX, MATMPIAIJHIPX)); /* -mat_type aijhipx resolves by communicator size, matreg.c:128-138 */
  PetscCall(PCRegister(PCJACOBIHIPX, PCCreate_JacobiHIPX));
  PetscCall(PCRegister("pbjacobihipx", PCCreate_PBJacobiHIPX));
  PetscCall(KSPRegister("cghipx", KSPCreate_CGHIPX));
  PetscCall(KSPRegister("chebyshevhipx", KSPCreate_ChebyshevHIPX));
  PetscCall(KSPRegister("pipecghipx", KSPCreate_PIPECGHIPX));
  PetscCall(KSPRegister("groppcghipx", KSPCreate_GROPPCGHIPX));
  PetscCall(PetscSFInitializePackage()); /* registers "basic", whose creator the hipx type builds on */
  PetscCall(PetscSFRegister(PETSCSFHIPX, PetscSFCreate_HIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
