static const char *petscmachineinfo = "\n-----------------------------------------\nLibraries compiled by oracle/build_ref.py\nUsing PETSc directory: /root/reference\nUsing PETSc arch: oracle-ref\n-----------------------------------------\n";
static const char *petsccompilerinfo = "\nUsing C compiler: gcc -fPIC -fstack-protector -fvisibility=hidden -O2\n-----------------------------------------\n";
static const char *petsccompilerflagsinfo = "\nUsing include paths: -I/root/reference/include -Ioracle/ref_conf\n-----------------------------------------\n";
static const char *petsclinkerinfo = "\nUsing C linker: gcc\nUsing libraries: -lpetsc -lmkl_rt -lm\n-----------------------------------------\n";
