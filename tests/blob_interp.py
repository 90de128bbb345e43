"""CPU interpreter of the kernel-side weight blob (TEST helper, not product code).

Evaluates the EPiC network *from the packed blob + descriptor* with plain torch ops, following the
same decomposition the HIP kernels use (per-particle block + per-jet bias GEMV).  Comparing it with
the oracle validates particle_fm_amd/layout.py on a machine without a GPU."""
import math

import torch
import torch.nn.functional as F

H = 128


def _plain(blob, off, K, OUT):
    return blob[off: off + K * OUT].view(K, OUT)


def _kmajor(blob, off, K, OUT):
    """decode a KM16 (OUT=128) or KP16 (OUT<=16) block back to [K][OUT]"""
    K16 = (K + 15) // 16 * 16
    if OUT == H:
        k = torch.arange(K16).view(-1, 1)
        o = torch.arange(OUT).view(1, -1)
        pos = ((k >> 4) * 32 + (o >> 2)) * 64 + (k & 15) * 4 + (o & 3)
        return blob[off: off + K16 * OUT][pos][:K]
    return blob[off: off + K16 * 16].view(K16, 16)[:K, :OUT]


def _mfma_a(blob, off, transposed=False):
    """undo the MFMA_A / MFMA_AT packing -> W_x (128 x 128), [out][in]"""
    a = blob[off: off + H * H].view(8, 8, 64, 4)
    W = torch.zeros(H, H, dtype=blob.dtype)
    w = torch.arange(8).view(8, 1, 1, 1)
    kt = torch.arange(8).view(1, 8, 1, 1)
    lane = torch.arange(64).view(1, 1, 64, 1)
    r = torch.arange(4).view(1, 1, 1, 4)
    i = (16 * w + (lane & 15)).expand(8, 8, 64, 4)
    k = (16 * kt + 4 * (lane >> 4) + r).expand(8, 8, 64, 4)
    if not transposed:
        W[i.reshape(-1), k.reshape(-1)] = a.reshape(-1)
    else:
        W[k.reshape(-1), i.reshape(-1)] = a.reshape(-1)
    return W


def interp_forward(layout, blob, t, x, cond, mask, trace=None):
    d, cfg = layout.desc, layout.cfg
    N, Fe, T, C, Cl, L = cfg.num_particles, cfg.features, cfg.t_dim, cfg.global_cond_dim, cfg.local_cond_dim, cfg.latent
    B = x.shape[0]
    slope, ss = cfg.neg_slope, cfg.sum_scale
    act = lambda z: F.leaky_relu(z, slope)
    maskf = torch.ones(B, N) if mask is None else mask.reshape(B, N).float()
    freqs = blob[d.freqs: d.freqs + T]
    temb = torch.cos((t[:, None] + 0.0) * freqs[None, :] * math.pi / 1.0)  # (B,T)
    condv = cond if C > 0 else torch.zeros(B, 0)
    e_loc = torch.cat([temb, condv[:, :Cl]], -1)
    Ke = T + Cl

    def bvec(off, n):
        return blob[off: off + n]

    bj1 = bvec(d.l1_b, H) + e_loc @ _kmajor(blob, d.l1_We, Ke, H)
    bj2 = bvec(d.l2.b, H) + e_loc @ _kmajor(blob, d.l2.We, Ke, H)
    x1 = act(x @ _plain(blob, d.l1x.W, Fe, H) + bj1[:, None, :])
    h = act(x1 @ _mfma_a(blob, d.l2.A).T + bj2[:, None, :] + x1)
    if trace is not None:
        trace.update(temb=temb, x1=x1, x2=h, bj1_stem=bj1, bj2_stem=bj2)
    nvalid = maskf.sum(1, keepdim=True)

    def pool(hh):
        s = (hh * maskf[..., None]).sum(1)
        return s / nvalid, s * ss

    mean, ssum = pool(h)
    vin = torch.cat([temb, condv, mean, ssum], -1)
    g1 = act(vin @ _kmajor(blob, d.g1.W, T + C + 2 * H, H) + bvec(d.g1.b, H))
    g = act(torch.cat([temb, condv, g1], -1) @ _kmajor(blob, d.g2.W, T + C + H, L) + bvec(d.g2.b, L))
    if trace is not None:
        trace.update(gstem1=g1, gstem=g, pool0=(h * maskf[..., None]).sum(1))
    for k in range(cfg.layers):
        ly = d.layer[k]
        mean, ssum = pool(h)
        vin = torch.cat([temb, condv, mean, ssum, g], -1)
        g1 = act(vin @ _kmajor(blob, ly.gl1.W, T + C + 2 * H + L, H) + bvec(ly.gl1.b, H))
        g = act(torch.cat([temb, condv, g1], -1) @ _kmajor(blob, ly.gl2.W, T + C + H, L) + bvec(ly.gl2.b, L) + g)
        e1 = torch.cat([e_loc, g], -1)
        bj1 = bvec(ly.lc1.b, H) + e1 @ _kmajor(blob, ly.lc1.We, Ke + L, H)
        bj2 = bvec(ly.lc2.b, H) + e_loc @ _kmajor(blob, ly.lc2.We, Ke, H)
        W1 = _mfma_a(blob, ly.lc1.A)
        W2 = _mfma_a(blob, ly.lc2.A)
        if ly.lc1.AT >= 0:
            assert torch.equal(_mfma_a(blob, ly.lc1.AT, True), W1)
            assert torch.equal(_mfma_a(blob, ly.lc2.AT, True), W2)
        l1 = act(h @ W1.T + bj1[:, None, :])
        h = act(l1 @ W2.T + bj2[:, None, :] + h)
        if trace is not None:
            trace[f'l1_{k}'] = l1; trace[f'xo_{k}'] = h; trace[f'g1_{k}'] = g1; trace[f'g_{k}'] = g
    W3 = blob[d.l3_W: d.l3_W + Fe * H].view(Fe, H)
    bj3 = bvec(d.l3_b, Fe) + e_loc @ _plain(blob, d.l3_We, Ke, Fe)
    out = act(h @ W3.T + bj3[:, None, :])
    return out * maskf[..., None]
