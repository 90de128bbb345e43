#!/bin/bash
# Diagnostic library for tests/diag/bwd_stamps.py: the shipped objects (build/obj, python -c "import __graft_entry__ as g; g.build()" first) with
# epic_train.hip recompiled under -DPFM_BDIAG (s_memtime stamps in the backward chain kernel) -> tests/diag/libtr_stamps.so.  CPU container.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/build/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DPFM_BDIAG $PFM_DEFS -I$R/include -I$R/particle_fm_amd/csrc \
    -c $R/particle_fm_amd/csrc/epic_train.hip -o $O/epic_train_bdiag.o
objs=$(ls $O/*.hip.o | grep -v epic_train)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $O/epic_train_bdiag.o -o $R/tests/diag/libtr_stamps.so
echo built $R/tests/diag/libtr_stamps.so
