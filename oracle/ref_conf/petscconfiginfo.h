#if defined(HIPX_REF_MPICH)
static const char *petscconfigureoptions = "none: compiled by oracle/build_ref.py with the hand-written oracle/ref_conf/petscconf.h (gcc -O2, MPICH of the image, libmkl_rt)";
#else
static const char *petscconfigureoptions = "none: compiled by oracle/build_ref.py with the hand-written oracle/ref_conf/petscconf.h (gcc -O2, MPIUNI, libmkl_rt)";
#endif
