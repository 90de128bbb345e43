"""CPU interpreter of the transformer weight blob (formats of include/pfm_tf.h): decodes every region through the
descriptor's offsets and evaluates the network with plain torch ops.  Test infrastructure: proves on the CPU that
layout_tf.py puts each parameter where the kernels will look for it."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def mfma_ak(blob, off, NO, K):
    """inverse of the MFMA_AK packing -> W [NO][K]"""
    nst = K // 64  # 64-wide k steps; two consecutive steps are one 128-wide chunk of the header's description
    n = NO * K
    a = blob[off:off + n].reshape(NO // 16, nst, 4, 64, 4)
    W = torch.empty(NO, K)
    ob, st, kt, lane, r = np.meshgrid(np.arange(NO // 16), np.arange(nst), np.arange(4), np.arange(64), np.arange(4),
                                      indexing="ij")
    W[16 * ob + (lane & 15), 64 * st + 16 * kt + 4 * (lane >> 4) + r] = a
    return W


def kmajor(blob, off, K, NO):
    return blob[off:off + K * NO].reshape(K, NO).t()  # -> [NO][K]


def vec(blob, off, n):
    return blob[off:off + n]


def ln(blob, nrm, n, x, eps):
    return F.layer_norm(x, (n,), vec(blob, nrm.gamma, n), vec(blob, nrm.beta, n), eps)


def forward(desc, blob, t, x, cond, mask):
    """t (B,), x (B,N,F), cond (B,C)|None, mask (B,N) float -> v (B,N,F); also checks MFMA_AKT copies."""
    d = desc
    B, N, Fe = x.shape
    D, Hd, T, C, CO, CH = d.model_dim, d.hidden, d.t_dim, d.cond_dim, d.ctxt_dim, d.ctxt_hidden
    freqs = vec(blob, d.freqs, T)
    if d.flags & 2:  # PFM_TF_F_TEMB_SINCOS: table = [f ; f]
        a = freqs * t[:, None]
        temb = torch.cat([a[:, :T // 2].cos(), a[:, T // 2:].sin()], -1)
    else:
        temb = torch.cos((t[:, None] + 0.0) * freqs * math.pi / 1.0)
    cin = temb if C == 0 else torch.cat([temb, cond], -1)
    h = F.leaky_relu(cin @ kmajor(blob, d.c1.W, T + C, CH).t() + vec(blob, d.c1.b, CH), d.neg_slope)
    h = ln(blob, d.c_norm, CH, h, d.ln_eps)
    ctxt = h @ kmajor(blob, d.c2.W, CH, CO).t() + vec(blob, d.c2.b, CO)

    def jet_bias(lin, with_t):
        jb = vec(blob, lin.b, Hd) + ctxt @ kmajor(blob, lin.Wc, CO, Hd).t()
        if with_t:
            jb = jb + temb @ kmajor(blob, lin.Wt, T, Hd).t()
        return jb[:, None, :]

    def check_T(lin, NO, K):
        W = mfma_ak(blob, lin.W, NO, K)
        WT = mfma_ak(blob, lin.WT, K, NO)
        assert torch.equal(W.t(), WT)
        return W

    h = F.leaky_relu(x @ kmajor(blob, d.n1.W, Fe, Hd).t() + jet_bias(d.n1, bool(d.time_in_input)), d.neg_slope)
    h = ln(blob, d.n_norm, Hd, h, d.ln_eps)
    xs = h @ check_T(d.n2, D, Hd).t() + vec(blob, d.n2.b, D)
    kvbias = torch.zeros(B, 1, 1, N).masked_fill(mask[:, None, None, :] == 0, -float("inf"))
    for l in range(d.layers):
        L = d.layer[l]
        qkv = ln(blob, L.norm1, D, xs, d.ln_eps) @ check_T(L.qkv, 3 * D, D).t() + vec(blob, L.qkv.b, 3 * D)
        q, k, v = (a.view(B, N, d.heads, d.head_dim).transpose(1, 2) for a in qkv.chunk(3, -1))
        s = q @ k.transpose(-2, -1) / math.sqrt(d.head_dim) + kvbias
        a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, N, D)
        a = ln(blob, L.attn_norm, D, a, d.ln_eps)
        xs = xs + a @ check_T(L.out, D, D).t() + vec(blob, L.out.b, D)
        h = ln(blob, L.norm2, D, xs, d.ln_eps) @ check_T(L.d1, Hd, D).t() + jet_bias(L.d1, False)
        h = ln(blob, L.d_norm, Hd, F.leaky_relu(h, d.neg_slope), d.ln_eps)
        xs = xs + h @ check_T(L.d2, D, Hd).t() + vec(blob, L.d2.b, D)
    h = ln(blob, d.final_norm, D, xs, d.ln_eps) @ check_T(d.o1, Hd, D).t() + jet_bias(d.o1, False)
    h = ln(blob, d.o_norm, Hd, F.leaky_relu(h, d.neg_slope), d.ln_eps)
    W3 = blob[d.o2.W:d.o2.W + Fe * Hd].reshape(Fe, Hd)
    return h @ W3.t() + vec(blob, d.o2.b, Fe)
