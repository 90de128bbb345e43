#!/bin/bash
# Diagnostic library variants of the row-matrix EPiC path (timing only): the shipped objects (build/obj; run __graft_entry__.build() first) with
# ew_kernels.hip recompiled under $PFM_DEFS (e.g. -DPFM_EW_AB_NOCHAIN) -> tests/diag/libew_ab.so; use with PFM_DIAG=1 PFM_LIB_PATH=...  CPU container.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/build/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $PFM_DEFS -I$R/include -I$R/particle_fm_amd/csrc \
    -c $R/particle_fm_amd/csrc/ew_kernels.hip -o $O/ew_kernels_ab.o
objs=$(ls $O/*.hip.o | grep -v ew_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $O/ew_kernels_ab.o -o $R/tests/diag/libew_ab.so
echo built $R/tests/diag/libew_ab.so
