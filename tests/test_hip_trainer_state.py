"""FusedFMTrainer stands in for Lightning + AdamW + the EMA callback (callbacks/ema.py): its state must be saveable and resumable
(Adam moments, step count, EMA weights), the EMA weights usable (swap / export), the learning rate schedulable."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

HP = dict(model="epic", features=3, hidden_dim=128, num_particles=30, frequencies=16, layers=2, latent=10, t_local_cat=True,
          t_global_cat=True, add_time_to_input=False, t_emb="cosine", loss_type="FM-OT", sigma=1e-4)


def _batch(B=16, N=30):
    gen = torch.Generator().manual_seed(3)
    n = torch.randint(6, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, N, 3, generator=gen) * mask
    return x.cuda(), mask.cuda(), torch.zeros(B).cuda()


def _make(seed=11, **kw):
    from particle_fm_amd.engine import FusedFMTrainer
    from particle_fm_amd.models import SetFlowMatchingLitModule
    torch.manual_seed(seed)
    m = SetFlowMatchingLitModule(optimizer=None, **HP).cuda()
    return m, FusedFMTrainer(m, lr=1e-3, weight_decay=5e-5, max_grad_norm=0.5, ema_decay=0.999, **kw)


def test_save_load_resume_is_bit_identical():
    from particle_fm_amd.engine import cosine_warmup
    batch = _batch()
    sched = cosine_warmup(3, 50)
    m, tr = _make(lr_schedule=sched)
    torch.manual_seed(5); torch.cuda.manual_seed_all(5)
    for _ in range(3):
        tr.step(batch)
    ck_model = copy.deepcopy(m.state_dict())
    ck_tr = tr.state_dict()
    rng = (torch.get_rng_state(), torch.cuda.get_rng_state())
    for _ in range(2):
        tr.step(batch)
    want = (tr.fp.flat.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.ema.clone(), tr.step_count)

    m2, tr2 = _make(seed=99, lr_schedule=sched)  # different init: everything must come from the checkpoint
    m2.load_state_dict(ck_model)
    tr2.load_state_dict(ck_tr)
    assert tr2.step_count == 3 and abs(tr2.current_lr() - 1e-3 * sched(3)) < 1e-12
    torch.set_rng_state(rng[0]); torch.cuda.set_rng_state(rng[1])
    for _ in range(2):
        tr2.step(batch)
    got = (tr2.fp.flat, tr2.exp_avg, tr2.exp_avg_sq, tr2.ema, tr2.step_count)
    assert got[4] == want[4] == 5
    for a, b, name in zip(want[:4], got[:4], ("params", "exp_avg", "exp_avg_sq", "ema")):
        assert torch.equal(a, b), f"{name} differ after resume: max {float((a - b).abs().max()):.3e}"

    # a checkpoint of another parameter list is refused (the same names in another ORDER of the flat buffer are loaded by name:
    # tests/test_hip_trainer_split.py)
    bad = dict(ck_tr, param_names=["renamed." + ck_tr["param_names"][0]] + list(ck_tr["param_names"][1:]))
    with pytest.raises(ValueError):
        tr2.load_state_dict(bad)


def test_ema_swap_and_export():
    batch = _batch()
    m, tr = _make()
    for _ in range(3):
        tr.step(batch)
    live = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ema_sd = tr.ema_state_dict()
    assert list(ema_sd.keys()) == list(live.keys())  # the EMA callback zips by position (ema.py:149-151)
    names = dict(m.named_parameters())
    # the fused kernel's EMA == the callback's update ema -= (1 - decay) (ema - w), three times from ema0 = w0 (ema.py:73-81)
    changed = [k for k in names if not torch.equal(ema_sd[k], live[k])]
    assert len(changed) == len(names) == 39  # 13 Linears x (weight_g, weight_v, bias), every one has moved
    with torch.no_grad():
        torch.manual_seed(1)
        s_live = m.sample(4, cond=None, mask=batch[1][:4], ode_steps=4)
        with tr.swap_ema():
            for k, p in m.named_parameters():
                assert torch.equal(p.detach(), ema_sd[k]), k
            torch.manual_seed(1)
            s_ema = m.sample(4, cond=None, mask=batch[1][:4], ode_steps=4)
        for k, v in m.state_dict().items():
            assert torch.equal(v, live[k]), k  # restored
        torch.manual_seed(1)
        s_back = m.sample(4, cond=None, mask=batch[1][:4], ode_steps=4)
    assert torch.equal(s_live, s_back) and not torch.equal(s_live, s_ema)
    # buffers (frequencies) are carried over untouched
    for k, v in live.items():
        if k not in names:
            assert torch.equal(ema_sd[k], v)


def test_lr_schedule_is_applied():
    batch = _batch()
    m0, tr0 = _make(lr_schedule=lambda k: 0.0)
    p0 = tr0.fp.flat.clone()
    tr0.step(batch)
    # lr = 0: AdamW moves nothing (weight decay is lr * wd * p), the moments still update
    assert torch.equal(tr0.fp.flat, p0) and float(tr0.exp_avg.abs().max()) > 0
    m1, tr1 = _make(lr_schedule=lambda k: 0.5)
    m2, tr2 = _make()
    tr2.lr = 0.5e-3
    torch.manual_seed(2); torch.cuda.manual_seed_all(2)
    tr1.step(batch)
    torch.manual_seed(2); torch.cuda.manual_seed_all(2)
    tr2.step(batch)
    assert torch.equal(tr1.fp.flat, tr2.fp.flat)


def test_a_loss_kept_across_many_steps_keeps_its_value():
    """The loss step() returns lives in a buffer of that step's own ([loss, 1 / sum(mask)], allocated per step; round 3 handed out a view
    into a 16-slot ring): a caller that collects per-step losses for an epoch mean still reads each step's own number 40 steps later."""
    batch = _batch()
    m, tr = _make()
    kept, now = [], []
    for _ in range(40):
        l = tr.step(batch)
        kept.append(l)
        now.append(float(l))
    later = [float(l) for l in kept]
    assert later == now
    assert len(set(now)) > 30  # the loss does move from step to step (fresh t, z draws and updated weights)
    assert abs(float(torch.stack(kept).mean()) - sum(now) / len(now)) < 1e-5
