"""pfm_optim_step is a pure function of its inputs: the global gradient norm is reduced in a fixed order (no atomics), so two
ranks holding the same all-reduced gradient apply bit-identical updates (the invariant DDP relies on)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [561330, 8504698 + 3, 1023])
def test_optim_step_bitwise_repeatable_and_matches_torch_norm(n):
    from particle_fm_amd import _lib, hip_ops
    P = hip_ops._ptr
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g).cuda()
    grad = (torch.randn(n, generator=g) * 3.0).cuda()  # norm >> 0.5: the clip factor matters
    outs = []
    for rep in range(6):
        p, m, v, ema = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), p0.clone()
        scratch = torch.full((1024,), float(rep), device="cuda")  # stale contents must not matter
        rc = _lib.load().pfm_optim_step(P(p), P(grad), P(m), P(v), P(ema), P(scratch), ctypes.c_int64(n), 0.5, 0.5, 1e-3, 0.9, 0.999,
                                        1e-8, 5e-5, 0.999, 1, hip_ops._stream_ptr(p.device))
        _lib.check(rc, "pfm_optim_step")
        torch.cuda.synchronize()
        outs.append((p, m, v, ema, scratch[0].clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    want = (grad.double() * 0.5).norm() ** 2
    assert abs(outs[0][4].double().item() - want.item()) < 1e-5 * want.item()
