static const char *petscconfigureoptions = "none: compiled by oracle/build_ref.py with the hand-written oracle/ref_conf/petscconf.h (gcc -O2, MPIUNI, libmkl_rt)";
