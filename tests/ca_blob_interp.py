"""CPU interpreter of the cross-attention weight blob (formats of include/pfm_ca.h), as tf_blob_interp.py: proves on the
CPU that layout_ca.py puts each parameter where the kernels will look for it."""
import math

import torch
import torch.nn.functional as F

from .tf_blob_interp import kmajor, ln, mfma_ak, vec


def forward(desc, blob, t, x, cond, mask):
    """t (B,), x (B,N,F), cond (B,C)|None, mask (B,N) float -> v (B,N,F); also checks the transposed weight copies."""
    d = desc
    B, N, Fe = x.shape
    D, Hd, T, C, CO, CH, Tk = d.model_dim, d.hidden, d.t_dim, d.cond_dim, d.ctxt_dim, d.ctxt_hidden, d.tokens
    freqs = vec(blob, d.freqs, T)
    if d.flags & 2:
        a = freqs * t[:, None]
        temb = torch.cat([a[:, :T // 2].cos(), a[:, T // 2:].sin()], -1)
    else:
        temb = torch.cos(t[:, None] * freqs * math.pi)
    cin = temb if C == 0 else torch.cat([temb, cond], -1)
    h = F.leaky_relu(cin @ kmajor(blob, d.c1.W, T + C, CH).t() + vec(blob, d.c1.b, CH), d.neg_slope)
    h = ln(blob, d.c_norm, CH, h, d.ln_eps)
    ctxt = h @ kmajor(blob, d.c2.W, CH, CO).t() + vec(blob, d.c2.b, CO)

    def jet_bias(lin, with_t):
        jb = vec(blob, lin.b, Hd) + ctxt @ kmajor(blob, lin.Wc, CO, Hd).t()
        if with_t:
            jb = jb + temb @ kmajor(blob, lin.Wt, T, Hd).t()
        return jb[:, None, :]

    def W(lin, NO, K):
        w = mfma_ak(blob, lin.W, NO, K)
        assert torch.equal(w.t(), mfma_ak(blob, lin.WT, K, NO))
        return w

    def layer(L, q_seq, kv_seq, bias):
        qn, kvn = ln(blob, L.norm1, D, q_seq, d.ln_eps), ln(blob, L.norm0, D, kv_seq, d.ln_eps)
        q = qn @ W(L.q, D, D).t() + vec(blob, L.q.b, D)
        k, v = (kvn @ W(L.kv, 2 * D, D).t() + vec(blob, L.kv.b, 2 * D)).chunk(2, -1)
        q, k, v = (a.view(B, -1, d.heads, d.head_dim).transpose(1, 2) for a in (q, k, v))
        s = q @ k.transpose(-2, -1) / math.sqrt(d.head_dim)
        if bias is not None:
            s = s + bias
        a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, -1, D)
        a = ln(blob, L.attn_norm, D, a, d.ln_eps)
        q_seq = q_seq + a @ W(L.out, D, D).t() + vec(blob, L.out.b, D)
        h = ln(blob, L.norm2, D, q_seq, d.ln_eps) @ W(L.d1, Hd, D).t() + jet_bias(L.d1, False)
        h = ln(blob, L.d_norm, Hd, F.leaky_relu(h, d.neg_slope), d.ln_eps)
        return q_seq + h @ W(L.d2, D, Hd).t() + vec(blob, L.d2.b, D)

    h = F.leaky_relu(x @ kmajor(blob, d.n1.W, Fe, Hd).t() + jet_bias(d.n1, bool(d.time_in_input)), d.neg_slope)
    h = ln(blob, d.n_norm, Hd, h, d.ln_eps)
    seq = h @ W(d.n2, D, Hd).t() + vec(blob, d.n2.b, D)
    tok = vec(blob, d.global_tokens, Tk * D).reshape(1, Tk, D).expand(B, -1, -1)
    kvbias = torch.zeros(B, 1, 1, N).masked_fill(mask[:, None, None, :] == 0, -float("inf"))
    for l in range(d.layers):
        tok = layer(d.from_layer[l], tok, seq, kvbias)
        seq = layer(d.to_layer[l], seq, tok, None)
    h = seq @ W(d.o1, Hd, D).t() + jet_bias(d.o1, False)
    h = ln(blob, d.o_norm, Hd, F.leaky_relu(h, d.neg_slope), d.ln_eps)
    W3 = blob[d.o2.W:d.o2.W + Fe * Hd].reshape(Fe, Hd)
    return h @ W3.t() + vec(blob, d.o2.b, Fe)
