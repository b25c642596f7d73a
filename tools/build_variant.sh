#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh NAME "-DSOME_MACRO ..."   ->  groth_sahai_rs_amd/lib/var/NAME.so
# and run anything against it with GS_AMD_LIB=groth_sahai_rs_amd/lib/var/NAME.so (capi.py honours it).
set -e
cd "$(dirname "$0")/../groth_sahai_rs_amd/csrc"
NAME=$1; shift
mkdir -p ../lib/var ../lib/obj
FLAGS=$(make -s print-flags ${GS_MAKE_ARGS})   # GS_MAKE_ARGS="INLINE=" builds the out-of-line fallback
/opt/rocm/bin/hipcc $FLAGS "$@" -c gs_amd.hip -o ../lib/obj/var_$NAME.o
[ -f ../lib/obj/gs_multi.o ] || make ../lib/obj/gs_multi.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../lib/obj/var_$NAME.o ../lib/obj/gs_multi.o -o ../lib/var/$NAME.so -ldl -lpthread
echo built ../lib/var/$NAME.so
