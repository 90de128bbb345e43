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
