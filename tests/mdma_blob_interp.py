"""CPU interpreter of the MDMA weight blob (formats of include/pfm_mdma.h), as tf_blob_interp.py: proves on the CPU that
layout_mdma.py puts each parameter where the kernels will look for it -- and that the kernels' split of the network
(particle-row GEMMs, per-jet token steps, the token columns of fc1 as a jet bias) is the reference's arithmetic."""
import math

import torch
import torch.nn.functional as F

from .tf_blob_interp import kmajor, mfma_ak, vec


def forward(desc, blob, t, x, mask, cond=None):
    """t (B,), x (B,N,F), mask (B,N) float, cond (B,1) (conditional variant, desc.c_cat) -> v (B,N) (one output per particle); also
    checks the transposed weight copies."""
    d = desc
    B, N, Fe = x.shape
    H, L, T = d.hidden, d.latent, d.t_dim
    sl = d.neg_slope

    def W(lin, NO, K):
        w = mfma_ak(blob, lin.W, NO, K)
        assert torch.equal(w.t(), mfma_ak(blob, lin.WT, K, NO))
        return w

    jbt = vec(blob, d.emb_b, H).expand(B, H)
    Tl, Tg = (T if d.t_cat & 1 else 0), (T if d.t_cat & 2 else 0)
    if d.time_in_input or d.t_cat:
        freqs = vec(blob, d.freqs, T)
        if d.flags & 2:
            a = freqs * t[:, None]
            temb = torch.cat([a[:, :T // 2].cos(), a[:, T // 2:].sin()], -1)
        else:
            temb = torch.cos(t[:, None] * freqs * math.pi)
        tact = F.leaky_relu(temb, sl)
    if d.time_in_input:
        jbt = jbt + temb @ kmajor(blob, d.emb_Wt, T, H).t()
    if Tl:
        jbt = jbt + temb @ kmajor(blob, d.emb_Wt2, T, H).t()
    gcd, gcc, lcc = d.c_cat & 1, (d.c_cat >> 1) & 1, (d.c_cat >> 2) & 1
    cnd = cond.reshape(B, 1).float() if d.c_cat else None
    if lcc:
        jbt = jbt + cnd @ kmajor(blob, d.emb_Wc, 1, H).t()
    X = F.leaky_relu(x @ kmajor(blob, d.emb_Wx, Fe, H).t() + jbt[:, None, :], sl) * (mask != 0)[..., None]
    nv = mask.sum(1, keepdim=True)
    cl = cnd if (gcd or gcc) else nv  # cond[..., -1:] of the reference
    pooled = torch.cat([X.sum(1) / d.avg_n, nv] + ([cnd] if gcd else []), -1)
    ea = pooled @ kmajor(blob, d.ecls_W, H + 1 + gcd, L).t() + vec(blob, d.ecls_b, L)
    eg = torch.cat([nv] + ([cnd] if gcd else []), -1) @ kmajor(blob, d.cond_W, 1 + gcd, L).t() + vec(blob, d.cond_b, L)
    xc = ea * torch.sigmoid(eg)
    pad = torch.zeros(B, 1, N).masked_fill(mask[:, None, :] == 0, -float("inf"))
    for l in range(d.layers):
        k = d.block[l]
        Hh = F.leaky_relu(X, sl) @ W(k.fc0, H, H).t() + vec(blob, k.fc0.b, H)
        if Tl:
            Hh = Hh + (tact @ kmajor(blob, k.fc0.Wt, T, H).t())[:, None, :]
        if lcc:
            Hh = Hh + (F.leaky_relu(cl, sl) @ kmajor(blob, k.fc0.Wc, 1, H).t())[:, None, :]
        al = torch.cat([F.leaky_relu(xc, sl)] + ([tact] if Tg else []) + ([F.leaky_relu(cl, sl)] if gcc else []), -1)
        pre = al @ kmajor(blob, k.fc0c_W, L + Tg + gcc, H).t() + vec(blob, k.fc0c_b, H)
        c = F.layer_norm(pre, (H,), vec(blob, k.ln_g, H), vec(blob, k.ln_b, H), d.ln_eps)
        q = c @ kmajor(blob, k.q_W, H, H).t() + vec(blob, k.q_b, H)
        kk, vv = (Hh @ W(k.kv, 2 * H, H).t() + vec(blob, k.kv.b, 2 * H)).chunk(2, -1)
        qh = q.view(B, d.heads, 1, d.head_dim)
        kh, vh = (a.view(B, N, d.heads, d.head_dim).transpose(1, 2) for a in (kk, vv))
        s = qh @ kh.transpose(-2, -1) / math.sqrt(d.head_dim) + pad[:, None]
        att = (torch.softmax(s, -1) @ vh).reshape(B, H)
        o = att @ kmajor(blob, k.o_W, H, H).t() + vec(blob, k.o_b, H)
        tg = [temb] if Tg else []
        c2 = torch.cat([o, nv] + ([cnd] if gcd else []) + tg, -1) @ kmajor(blob, k.fc1c_W, H + 1 + gcd + Tg, L).t() + vec(blob, k.fc1c_b, L)
        xc = torch.cat([c2] + tg + ([cl] if gcc else []), -1) @ kmajor(blob, k.fc2c_W, L + Tg + gcc, L).t() + vec(blob, k.fc2c_b, L)
        jb = xc @ kmajor(blob, k.fc1.Wc, L, H).t() + vec(blob, k.fc1.b, H)
        if lcc:
            jb = jb + cl @ kmajor(blob, k.fc1.Wt, 1, H).t()
        X = Hh @ W(k.fc1, H, H).t() + jb[:, None, :] + X
    hb = vec(blob, d.out_b, 1)
    if lcc:
        hb = hb + F.leaky_relu(cnd, sl) * vec(blob, d.out_Wc, 1)
    return (F.leaky_relu(X, sl) @ vec(blob, d.out_W, H) + hb) * mask
