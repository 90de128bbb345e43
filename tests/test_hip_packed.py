"""The packed midpoint sampler: two short jets share a workgroup (one weight stream, one set of phases; epic_kernels.hip).
Results must be those of the one-jet-per-workgroup kernel and of the reference graph, whatever the pairing."""
import ctypes

import pytest
import torch

from oracle.fm_ref import EpicVectorField, sample_midpoint
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu


def _ragged(B, N, F, C, seed, lo=1):
    gen = torch.Generator().manual_seed(seed)
    n = torch.randint(lo, N + 1, (B,), generator=gen)
    n[0], n[1], n[2] = N, 1, 16  # a full jet, a one-particle jet, exactly one tile
    mask = (torch.arange(N)[None] < n[:, None]).float()
    # holes inside a jet (mask 0 before the last valid particle) must behave like in the one-jet kernel
    mask[3, : int(n[3]) // 2] = 0.0
    if int(n[3]) > 0:
        mask[3, int(n[3]) - 1] = 1.0
    z = torch.randn(B, N, F, generator=gen)
    cond = torch.randn(B, C, generator=gen) if C else None
    return n, mask.unsqueeze(-1), z, cond


def _pack_list(lay, B, steps):
    from particle_fm_amd import _lib
    lib = _lib.load()
    scr = [v for k, v in lay.__dict__["_sample_scratch"].items() if k[0] == steps and k[3] == B][0]
    total = lib.pfm_epic_sample_scratch_floats(ctypes.byref(lay.desc), steps - 1, B)
    ints = scr.view(torch.int32)[total - ((2 * B + 1 + 63) // 64) * 64:].cpu()
    nwg = int(ints[0])
    return nwg, ints[1:1 + 2 * nwg].reshape(nwg, 2)


@pytest.mark.parametrize("name,B", [("jetnet150", 48), ("cond_gl", 24), ("jetnet30", 40)])
def test_packed_sampler_matches_unpacked_and_oracle(name, B):
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden(name)
    N, F, C = g.hp["num_particles"], g.hp["features"], g.hp["global_cond_dim"]
    lay = EpicLayout(cfg_of(g.hp), flags=1 | 16)  # PFM_F_SKIP_MASKED_TAIL | PFM_F_PACK_JETS
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    n, mask, z, cond = _ragged(B, N, F, C, seed=31 + B, lo=1)
    dev = lambda a: None if a is None else a.cuda()
    steps = 6
    packed = hip_ops.epic_sample_midpoint(lay, blob, dev(z), dev(cond), dev(mask), ode_steps=steps).cpu()
    tile = hip_ops.packed_tile_rows(lay, N)  # short sets run on a larger LDS tile (zero-padded inputs, same weights)
    nwg, wl = _pack_list(lay.padded(tile) if tile else lay, B, steps)
    N_tile = tile or N
    assert sorted(int(v) for v in wl.reshape(-1) if v >= 0) == list(range(B))  # every jet exactly once
    pairs = int((wl[:, 1] >= 0).sum())
    assert nwg == B - pairs
    if N_tile >= 64:
        assert pairs >= B // 6, f"only {pairs} pairs formed out of {B} ragged jets"
    for a, b in wl.tolist():  # capacity rule
        if b >= 0:
            ra = int(mask[a].nonzero()[:, 0].max()) + 1
            rb = int(mask[b].nonzero()[:, 0].max()) + 1
            assert (ra + 15) // 16 * 16 + rb <= hip_ops._seg2_rows(N_tile)
    single = hip_ops.epic_sample_midpoint(lay, blob, dev(z), dev(cond), dev(mask), ode_steps=steps, time_table=False).cpu()
    # same arithmetic per row and per jet; the only difference is the tabulated time term (fp32 re-association, ~1e-7)
    torch.testing.assert_close(packed, single, atol=2e-6, rtol=1e-5)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref = sample_midpoint(vf, z, cond, mask, ode_steps=steps)
    torch.testing.assert_close(packed, ref, atol=2e-5, rtol=1e-4)
    assert torch.all(packed[mask.squeeze(-1) == 0] == 0)


def test_packed_sampler_is_deterministic_and_order_independent():
    """The same jets in another batch order pair up differently (the k-th longest with the shortest that fits): every jet's result
    must not depend on its partner beyond fp32 re-association of nothing at all -- rows and per-jet vectors never mix."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden("jetnet150")
    N, F = 150, 3
    lay = EpicLayout(cfg_of(g.hp), flags=1 | 16)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    B = 64
    n, mask, z, _ = _ragged(B, N, F, 0, seed=5, lo=8)
    a = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=5).cpu()
    a2 = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=5).cpu()
    assert torch.equal(a, a2)
    lay1 = EpicLayout(cfg_of(g.hp), flags=1)  # one jet per workgroup (same kernel, the lean evaluation of epic_fast.h): the same bits
    blob1 = lay1.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    assert torch.equal(a, hip_ops.epic_sample_midpoint(lay1, blob1, z.cuda(), None, mask.cuda(), ode_steps=5).cpu())
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    keep = perm[: B // 2]  # drop half of the jets: partners change
    b = hip_ops.epic_sample_midpoint(lay, blob, z[keep].cuda(), None, mask[keep].cuda(), ode_steps=5).cpu()
    assert torch.equal(a[keep], b)


def test_short_sets_pack_on_a_larger_tile():
    """A 30-particle model's own LDS tile never takes two jets; with packing on, the call runs on the smallest tile that does (80 rows,
    hip_ops.packed_tile_rows) with zero-padded inputs and the same weights: every jet keeps its bits."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden("jetnet30")
    N, F = g.hp["num_particles"], g.hp["features"]
    assert N == 30
    lay = EpicLayout(cfg_of(g.hp), flags=1 | 16)
    assert hip_ops.packed_tile_rows(lay, N) == 80
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    B = 96
    n, mask, z, _ = _ragged(B, N, F, 0, seed=77, lo=1)
    packed = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=7).cpu()
    lay1 = EpicLayout(cfg_of(g.hp), flags=1)
    blob1 = lay1.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    single = hip_ops.epic_sample_midpoint(lay1, blob1, z.cuda(), None, mask.cuda(), ode_steps=7).cpu()
    assert torch.equal(packed, single)
    # and the workgroup list of the padded call really holds pairs
    big = lay.padded(80)
    scr = [v for k, v in big.__dict__["_sample_scratch"].items() if k[0] == 7 and k[3] == B][0]
    from particle_fm_amd import _lib
    total = _lib.load().pfm_epic_sample_scratch_floats(ctypes.byref(big.desc), 6, B)
    ints = scr.view(torch.int32)[total - ((2 * B + 1 + 63) // 64) * 64:].cpu()
    nwg = int(ints[0])
    assert nwg == B // 2, f"{nwg} workgroups for {B} jets of <= 30 particles"
    # no mask at all: every jet is full length
    full = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, None, ode_steps=4).cpu()
    full1 = hip_ops.epic_sample_midpoint(lay1, blob1, z.cuda(), None, None, ode_steps=4).cpu()
    assert torch.equal(full, full1)


# ---- the instantiations behind the published cfg-2 numbers (bench_secondary.py --workload jetnet30 [--precision bf16]) -----------------
# epic_sample_midpoint_fast_kernel<1, true, false> (bf16 operands, two jets per workgroup on the 80-row tile), <1, false, true>
# (bf16, conditioned jets), and the full-size cfg-2 call (1024 jets of <= 30 particles, 100 steps).
def _autocast_bar(vf, z, cond, mask, steps):
    ref = sample_midpoint(vf, z, cond, mask, ode_steps=steps)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        rac = sample_midpoint(vf, z, cond, mask, ode_steps=steps).float()
    return ref, (rac - ref).abs()


def test_bf16_packed_sampler_equals_unpacked_bf16_bitwise_and_meets_the_autocast_bar():
    """flags = SKIP_MASKED_TAIL | BF16_MFMA | PACK_JETS on the 30-particle model (the kernel behind the cfg-2 bf16 line; round 3: FOUR
    jets per workgroup in fixed 32-row slots of a 128-row tile, epic_sample_midpoint_quad_kernel<1>): the same bits as one jet per
    workgroup with bf16 operands (rows and per-jet vectors never mix; the operand rounding is per element), and no further from the
    fp32 reference than the oracle under torch.autocast(bfloat16) (the bar of tests/test_hip_bf16.py)."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout, PFM_F_QUAD_JETS
    from tests.conftest import load_golden
    g = load_golden("jetnet30")
    N, F = g.hp["num_particles"], g.hp["features"]
    lay_p = EpicLayout(cfg_of(g.hp), flags=1 | 2 | 16)
    lay_1 = EpicLayout(cfg_of(g.hp), flags=1 | 2)
    big = hip_ops.packed_layout(lay_p, N)
    assert big.cfg.num_particles == 128 and int(big.desc.flags) & PFM_F_QUAD_JETS  # the call really runs four jets per workgroup
    assert hip_ops.packed_layout(lay_1, N) is None
    blob_p = lay_p.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    blob_1 = lay_1.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    steps = 12
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    for B in (96, 7, 1):  # whole quads; a last workgroup with 3 jets; a single jet
        n, mask, z, _ = _ragged(max(B, 4), N, F, 0, seed=123 + B, lo=1)
        mask, z = mask[:B], z[:B]
        packed = hip_ops.epic_sample_midpoint(lay_p, blob_p, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu()
        single = hip_ops.epic_sample_midpoint(lay_1, blob_1, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu()
        assert torch.equal(packed, single), B
        assert torch.all(packed[mask.squeeze(-1) == 0] == 0)
        if B == 96:
            ref, eac = _autocast_bar(vf, z, None, mask, steps)
            e16 = (packed - ref).abs()
            assert 1e-5 < e16.max() <= 1.5 * eac.max() + 2e-3, (e16.max(), eac.max())
            assert e16.mean() <= 1.5 * eac.mean() + 1e-4, (e16.mean(), eac.mean())
    # no mask at all: every jet is full length
    full = hip_ops.epic_sample_midpoint(lay_p, blob_p, z.cuda(), None, None, ode_steps=4).cpu()
    full1 = hip_ops.epic_sample_midpoint(lay_1, blob_1, z.cuda(), None, None, ode_steps=4).cpu()
    assert torch.equal(full, full1)
    # the fixture's own 100-step vector (reference vector field + restated integrator), through the quad bf16 kernel
    tag = "midpoint_100/"
    zz, mm, want = g.get(tag + "z"), g.get(tag + "mask"), g.get(tag + "x_end")
    out = hip_ops.epic_sample_midpoint(lay_p, blob_p, zz.cuda(), None, None if mm is None else mm.float().cuda(), ode_steps=100).cpu()
    out1 = hip_ops.epic_sample_midpoint(lay_1, blob_1, zz.cuda(), None, None if mm is None else mm.float().cuda(), ode_steps=100).cpu()
    assert torch.equal(out, out1)
    assert (out - want).abs().max() < 5e-2


def test_bf16_pair_packing_on_long_sets_equals_unpacked_bf16_bitwise():
    """epic_sample_midpoint_fast_kernel<1, true, false> (bf16 operands, two jets of arbitrary length per workgroup): what a bf16 model
    with sets beyond 32 particles runs under PFM_F_PACK_JETS.  Same bits as one jet per workgroup; pairs really form."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden("jetnet150")
    N, F = g.hp["num_particles"], g.hp["features"]
    lay_p = EpicLayout(cfg_of(g.hp), flags=1 | 2 | 16)
    lay_1 = EpicLayout(cfg_of(g.hp), flags=1 | 2)
    assert hip_ops.packed_layout(lay_p, N) is None  # the 150-row tile takes the pairs itself
    blob_p = lay_p.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    blob_1 = lay_1.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    B, steps = 48, 6
    n, mask, z, _ = _ragged(B, N, F, 0, seed=77, lo=1)
    packed = hip_ops.epic_sample_midpoint(lay_p, blob_p, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu()
    single = hip_ops.epic_sample_midpoint(lay_1, blob_1, z.cuda(), None, mask.cuda(), ode_steps=steps).cpu()
    assert torch.equal(packed, single)
    nwg, wl = _pack_list(lay_p, B, steps)
    assert int((wl[:, 1] >= 0).sum()) >= B // 6 and nwg == B - int((wl[:, 1] >= 0).sum())
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref, eac = _autocast_bar(vf, z, None, mask, steps)
    assert (packed - ref).abs().max() <= 1.5 * eac.max() + 2e-3


@pytest.mark.parametrize("B,steps", [(24, 8), (6, 40)])
def test_bf16_conditioned_lean_sampler_tracks_the_generic_bf16_kernel(B, steps):
    """epic_sample_midpoint_fast_kernel<1, false, true> (bf16 operands, conditioned jets: fm_tops*_cond.yaml, 2 + 2 values) against the
    generic bf16 kernel (PFM_F_GENERIC_SAMPLER), the autocast bar, and the reference's cond_gl midpoint vectors."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden("cond_gl")
    N, F, C = g.hp["num_particles"], g.hp["features"], g.hp["global_cond_dim"]
    fast = EpicLayout(cfg_of(g.hp), flags=1 | 2)
    gen = EpicLayout(cfg_of(g.hp), flags=1 | 2 | 32)
    from particle_fm_amd import _lib
    assert _lib.load().pfm_epic_sample_is_fast(ctypes.byref(fast.desc)) == 1
    assert _lib.load().pfm_epic_sample_is_fast(ctypes.byref(gen.desc)) == 0
    blob_f = fast.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    blob_g = gen.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    n, mask, z, cond = _ragged(B, N, F, C, seed=41 + B, lo=1)
    a = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=steps).cpu()
    b = hip_ops.epic_sample_midpoint(gen, blob_g, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=steps).cpu()
    # both round the same operands to bf16; inputs that differ by 1e-7 (tabulated terms) may round to neighbouring bf16 values
    torch.testing.assert_close(a, b, atol=3e-3, rtol=3e-3)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref, eac = _autocast_bar(vf, z, cond, mask, steps)
    e16 = (a - ref).abs()
    assert 1e-5 < e16.max() <= 1.5 * eac.max() + 2e-3, (e16.max(), eac.max())
    assert e16.max() <= 1.5 * (b - ref).abs().max() + 1e-3
    assert torch.all(a[mask.squeeze(-1) == 0] == 0)
    assert torch.equal(a, hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), cond.cuda(), mask.cuda(), ode_steps=steps).cpu())
    # a jet's result depends on its own conditioning only
    cond2 = cond.clone(); cond2[3:] += 1.0
    a2 = hip_ops.epic_sample_midpoint(fast, blob_f, z.cuda(), cond2.cuda(), mask.cuda(), ode_steps=steps).cpu()
    assert torch.equal(a2[:3], a[:3]) and not torch.equal(a2[3:], a[3:])
    for st in (3, 10):
        tag = f"midpoint_{st}/"
        out = hip_ops.epic_sample_midpoint(fast, blob_f, g.get(tag + "z").cuda(), g.get(tag + "cond").cuda(),
                                           None if g.get(tag + "mask") is None else g.get(tag + "mask").float().cuda(), ode_steps=st).cpu()
        assert (out - g.get(tag + "x_end")).abs().max() < 5e-2


@pytest.mark.parametrize("flags", [1 | 16, 1 | 2 | 16], ids=["fp32-packed", "bf16-packed"])
def test_cfg2_full_size_properties(flags):
    """BASELINE cfg 2 at its own size -- 1024 jets of the 30-particle model, multiplicities U{10..30}, ode_steps = 100, two jets per
    workgroup on the 80-row tile (fp32: 512 paired workgroups = two rounds on 256 CUs) or four per workgroup in 32-row slots (bf16: 256
    workgroups = one round; what bench_secondary.py --workload jetnet30 [--precision bf16] times) --
    through size-independent properties: a jet's result does not depend on its batch or partner (bitwise, 40 picked jets), masked rows
    are exactly 0, everything is finite, and the fp32 call agrees with the oracle on a handful of jets."""
    from particle_fm_amd import hip_ops
    from particle_fm_amd.layout import EpicLayout
    from tests.conftest import load_golden
    g = load_golden("jetnet30")
    N, F, B = g.hp["num_particles"], g.hp["features"], 1024
    lay = EpicLayout(cfg_of(g.hp), flags=flags)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    gen = torch.Generator().manual_seed(30 + flags)
    n = torch.randint(10, N + 1, (B,), generator=gen)
    mask = (torch.arange(N)[None] < n[:, None]).float().unsqueeze(-1)
    z = torch.randn(B, N, F, generator=gen)
    out = hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=100).cpu()
    if flags & 2:  # bf16: four jets per workgroup (256 workgroups = one round on 256 CUs), no workgroup list
        from particle_fm_amd.layout import PFM_F_QUAD_JETS
        big = hip_ops.packed_layout(lay, N)
        assert big.cfg.num_particles == 128 and int(big.desc.flags) & PFM_F_QUAD_JETS
    else:
        nwg, wl = _pack_list(lay.padded(80), B, 100)
        assert nwg == B // 2 and sorted(int(v) for v in wl.reshape(-1)) == list(range(B))
    assert torch.isfinite(out).all()
    assert torch.all(out[mask.squeeze(-1) == 0] == 0)
    pick = torch.randperm(B, generator=gen)[:40]
    sub = hip_ops.epic_sample_midpoint(lay, blob, z[pick].cuda(), None, mask[pick].cuda(), ode_steps=100).cpu()
    assert torch.equal(sub, out[pick])
    assert torch.equal(out, hip_ops.epic_sample_midpoint(lay, blob, z.cuda(), None, mask.cuda(), ode_steps=100).cpu())
    # one jet per workgroup, same operands: the same bits
    lay1 = EpicLayout(cfg_of(g.hp), flags=flags & ~16)
    blob1 = lay1.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    few = pick[:8]
    one = hip_ops.epic_sample_midpoint(lay1, blob1, z[few].cuda(), None, mask[few].cuda(), ode_steps=100).cpu()
    assert torch.equal(one, out[few])
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    ref = sample_midpoint(vf, z[few], None, mask[few], ode_steps=100)
    if flags & 2:
        assert (out[few] - ref).abs().max() < 5e-2
    else:
        torch.testing.assert_close(out[few], ref, atol=5e-5, rtol=1e-4)
