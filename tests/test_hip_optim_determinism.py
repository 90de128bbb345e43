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


def _ragged_loss_inputs(B, N=150, seed=7):
    from tests.conftest import load_golden
    g = load_golden("jetnet150")
    gen = torch.Generator().manual_seed(seed)
    n = torch.randint(1, N + 1, (B,), generator=gen)
    n[0], n[1] = N, 1  # a full jet and a one-particle jet
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, 3, generator=gen) * mask
    t = torch.rand(B, generator=gen)
    z = torch.randn(B, N, 3, generator=gen)
    return g, x, mask, t, z


def test_epic_backward_is_bitwise_repeatable_and_matches_the_oracle_on_a_ragged_batch():
    """The weight gradient is formed without atomics (gradient rows -> one dW GEMM over all jets' rows, split by row ranges,
    partials summed in a fixed order; rank-1 sums over jets in jet order): the same inputs give the same bits, run after run, and
    with many row splits in play (B = 40 jets, N = 150, ragged) every parameter gradient matches the oracle's autograd."""
    from oracle.fm_ref import EpicVectorField, fm_ot_loss
    from particle_fm_amd.fm_loss import epic_fm_loss
    from particle_fm_amd.layout import EpicLayout
    from tests.test_layout_cpu import cfg_of
    B = 40
    g, x, mask, t, z = _ragged_loss_inputs(B)
    hp = dict(g.hp)
    lay = EpicLayout(cfg_of(hp), flags=1)
    grads = []
    for rep in range(3):
        state = {k: v.clone().cuda().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items()}
        src = lay.source_vector(state, "flows.0.net.", freqs=g.freqs)
        loss = epic_fm_loss(lay, src, x.cuda(), t.cuda(), z.cuda(), None, mask.cuda(), sigma=1e-4)
        loss.backward()
        grads.append({k: v.grad.detach().clone() for k, v in state.items() if v.grad is not None})
        if rep == 1:  # something else in between must not matter (stale scratch)
            lay.__dict__.get("_bwd_scratch", {}).clear()
    for other in grads[1:]:
        for k, v in grads[0].items():
            assert torch.equal(v, other[k]), f"{k}: gradient differs between two runs (max {float((v - other[k]).abs().max()):.3e})"
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items()}
    vf = EpicVectorField(st, "flows.0.net", hp, freqs=g.freqs)
    lref, *_ = fm_ot_loss(vf, x, mask, None, t, z, sigma=1e-4)
    lref.backward()
    torch.testing.assert_close(loss.detach().cpu(), lref.detach(), atol=2e-6, rtol=2e-5)
    assert len(grads[0]) == 87
    for k, v in st.items():
        if v.grad is None:
            continue
        scale = max(v.grad.abs().max().item(), 1e-8)
        err = (grads[0][k].cpu() - v.grad).abs().max().item() / scale
        assert err <= 2e-4, (k, err)


def test_backward_writes_every_gradient_slot_and_chunked_batches_agree():
    """(i) the atomics-free backward WRITES the gradient blob -- FusedFMTrainer allocates it once and never zeroes it again,
    which is only right if every slot the unpack reads (layout.src_gpos) is written on every call: fill it with NaN first and look;
    (ii) batches beyond hip_ops.BWD_CHUNK_JETS run in chunks whose gradients are added in chunk order: same numbers as one call up to
    fp32 re-association, and bitwise repeatable."""
    import ctypes
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.test_layout_cpu import cfg_of
    B = 20
    g, x, mask, t, z = _ragged_loss_inputs(B, seed=11)
    lay = EpicLayout(cfg_of(dict(g.hp)), flags=1)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    maskf = mask.reshape(B, -1).float().cuda().contiguous()
    parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x.cuda(), t.cuda(), z.cuda(), None, maskf, 1e-4, "FM-OT", None)
    inv_total = (1.0 / count.sum()).reshape(1).contiguous()
    one = torch.ones(1, device="cuda")
    gpos = torch.from_numpy(lay.src_gpos.astype("int64")).cuda()

    def run():
        gblob = torch.full_like(blob, float("nan"))
        hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, inv_total, one, gblob)
        return gblob[gpos]

    whole = run()
    assert torch.isfinite(whole).all(), f"{int((~torch.isfinite(whole)).sum())} gradient slots were not written"
    assert float(whole.abs().max()) > 0
    keep = hip_ops.BWD_CHUNK_JETS
    try:
        hip_ops.BWD_CHUNK_JETS = 8  # 8 + 8 + 4 jets
        a, b = run(), run()
    finally:
        hip_ops.BWD_CHUNK_JETS = keep
    assert torch.isfinite(a).all() and torch.equal(a, b)
    scale = float(whole.abs().max())
    assert float((a - whole).abs().max()) <= 2e-6 * scale + 1e-9, float((a - whole).abs().max()) / scale
    # the scratch cache stays bounded
    assert len(lay.__dict__["_bwd_scratch"]) <= hip_ops._BWD_SCRATCH_SLOTS


def test_longest_first_launch_order_of_the_training_kernels_changes_nothing():
    """From hip_ops.JET_ORDER_MIN_JETS jets on the loss forward / backward take their jets longest first (pfm_epic_jet_order: scheduling
    only, every jet writes its own records): loss parts, saved activations and every gradient slot keep their bits."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    from tests.test_layout_cpu import cfg_of
    g = load_golden("jetnet30")
    N, F, B = g.hp["num_particles"], g.hp["features"], 448
    lay = EpicLayout(cfg_of(dict(g.hp)), flags=1)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    gen = torch.Generator().manual_seed(3)
    n = torch.randint(1, N + 1, (B,), generator=gen)
    maskf = (torch.arange(N)[None] < n[:, None]).float().cuda().contiguous()
    x = (torch.randn(B, N, F, generator=gen)).cuda() * maskf[..., None]
    t, z = torch.rand(B, generator=gen).cuda(), torch.randn(B, N, F, generator=gen).cuda()
    gpos = torch.from_numpy(lay.src_gpos.astype("int64")).cuda()
    one = torch.ones(1, device="cuda")

    def run():
        parts, count, saved = hip_ops.epic_fm_loss_forward(lay, blob, x, t, z, None, maskf, 1e-4, "FM-OT", None)
        gblob = torch.full_like(blob, float("nan"))
        hip_ops.epic_loss_backward(lay, blob, None, maskf, saved, (1.0 / count.sum()).reshape(1), one, gblob)
        return parts.clone(), saved.clone(), gblob[gpos].clone()

    assert hip_ops.jet_order(maskf, B, N) is not None
    order = hip_ops.jet_order(maskf, B, N).cpu().long()
    cnt = maskf.sum(1).cpu()
    assert sorted(order.tolist()) == list(range(B)) and bool((cnt[order][:-1] >= cnt[order][1:]).all())
    a = run()
    keep = hip_ops.JET_ORDER_MIN_JETS
    try:
        hip_ops.JET_ORDER_MIN_JETS = 1 << 30  # batch order
        b = run()
    finally:
        hip_ops.JET_ORDER_MIN_JETS = keep
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert torch.isfinite(a[2]).all()
