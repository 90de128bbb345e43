#!/bin/bash
# Timing-only ablation builds of the lean sampler (csrc/epic_fast.h, PFM_AB_* switches): each library is epic_kernels.hip alone with one
# piece of an evaluation removed.  Run HERE (CPU container, ~1 minute on 8 cores); then on the GPU box:
#     python tests/diag/ab_time.py tests/diag/lib_ab_base.so tests/diag/lib_ab_*.so > gpurun_out/ab_fixed.log
#     python tests/diag/fixed_cost_table.py gpurun_out/ab_fixed.log
set -e
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Iinclude -Iparticle_fm_amd/csrc"
build() { /opt/rocm/bin/hipcc $FL $2 particle_fm_amd/csrc/epic_kernels.hip -o tests/diag/lib_ab_$1.so > /tmp/ab_$1.log 2>&1 && echo built $1; }
build base "" &
build nochain "-DPFM_AB_NOCHAIN" &
build nohead "-DPFM_AB_NOHEAD" &
build nol1 "-DPFM_AB_NOL1" &
build nobar "-DPFM_AB_NOBAR" &
build noepi "-DPFM_AB_NOEPI" &
build nopoolfin "-DPFM_AB_NOPOOLFIN" &
build nopf "-DPFM_AB_NOPF" &
wait
build mfmaonly "-DPFM_AB_NOCHAIN -DPFM_AB_NOHEAD -DPFM_AB_NOL1 -DPFM_AB_NOBAR -DPFM_AB_NOEPI -DPFM_AB_NOPOOLFIN -DPFM_AB_NOPF" &
build nochain_nobar "-DPFM_AB_NOCHAIN -DPFM_AB_NOBAR" &
wait
ls -la tests/diag/lib_ab_*.so
