#!/bin/bash
# A/B builds of the library: profiles/build_variant.sh NAME "-DFLAG=1 ..." [f64|f32|both]
# -> emdee.jl_amd/variants/libemdee_hip_NAME.so (selected at run time with EMDEE_HIP_LIB=...), only the kernel
# translation unit(s) recompiled with the extra flags.  Run in the container (hipcc cross-compiles); *.so travels with gpurun.
set -e
NAME=$1; FLAGS=$2; WHICH=${3:-f64}
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/emdee.jl_amd/csrc; V=$R/emdee.jl_amd/variants; mkdir -p $V/obj
make -s -C $S -j4 >/dev/null
CXX="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast"
O64=$S/impl_f64.o; O32=$S/impl_f32.o
if [ $WHICH = f64 ] || [ $WHICH = both ]; then O64=$V/obj/impl_f64_$NAME.o; /opt/rocm/bin/hipcc $CXX $FLAGS -c $S/impl_f64.hip -o $O64; fi
if [ $WHICH = f32 ] || [ $WHICH = both ]; then O32=$V/obj/impl_f32_$NAME.o; /opt/rocm/bin/hipcc $CXX $FLAGS -c $S/impl_f32.hip -o $O32; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libemdee_hip_$NAME.so $S/capi.o $O32 $O64
echo "built $V/libemdee_hip_$NAME.so"
