"""Dropout on the HIP path (reference transformer.py:28 SDPA dropout_p and transformer.py:40 nn.Dropout).
The RNG stream cannot match torch's, so the checks are: (1) masks are regenerated identically in
backward (probe inputs make the mask directly visible in forward outputs and in gradients), (2) the
drop rate and 1/(1-p) scaling are right, (3) p = 0 reproduces the no-dropout path exactly,
(4) the module semantics (SDPA dropout even in eval, MLP dropout only in training)."""
import pytest
import torch

import vit_oracle as O

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def test_fc2_dropout_mask_is_shared_by_forward_and_backward(hip):
    from vitamd import ops
    M, N, K, p, seed = 1000, 768, 256, 0.25, 12345
    a = torch.zeros(M, K, device="cuda", dtype=BF16)
    w = torch.zeros(N, K, device="cuda", dtype=BF16)
    bias = torch.full((N,), 2.0, device="cuda")
    resid = torch.zeros(M, N, device="cuda")
    y = ops.linear_dropout_resid(a, w, bias, resid, (p, seed)).cpu()        # = dropout(2.0) elementwise
    kept = y != 0
    assert torch.allclose(y[kept], torch.full_like(y[kept], float(torch.tensor(2.0 / (1 - p)).bfloat16())))
    rate = 1.0 - kept.float().mean().item()
    assert abs(rate - p) < 0.01
    assert abs(kept.float().mean(0).min().item() - (1 - p)) < 0.08 and abs(kept.float().mean(1).min().item() - (1 - p)) < 0.08
    # backward side 1: the cast kernel (top layer) with the same (p, seed)
    g = torch.ones(M, N, device="cuda")
    dy = ops.cast_bf16_dropout(g, (p, seed)).float().cpu()
    assert torch.equal(dy != 0, kept)
    # backward side 2: the LayerNorm-backward emission with the same (p, seed)
    x = torch.randn(M, N, device="cuda")
    _, _, mean, rstd = ops.layernorm_fwd(x)
    dyl = torch.zeros(M, N, device="cuda", dtype=BF16)
    gres = torch.ones(M, N, device="cuda")
    cs = torch.zeros(N, device="cuda")
    _, gb = ops.layernorm_bwd(dyl, x, mean, rstd, g_res=gres, want_bf16=True, colsum=cs, dropout=(p, seed))
    assert torch.equal(gb.float().cpu() != 0, kept)
    assert O.rel_l2(cs.cpu(), gb.float().cpu().sum(0)) < 1e-5
    # a different seed gives a different mask; p = 0 keeps everything
    y2 = ops.linear_dropout_resid(a, w, bias, resid, (p, seed + 1)).cpu()
    assert (y2 != 0).ne(kept).float().mean().item() > 0.2
    assert torch.all(ops.linear_dropout_resid(a, w, bias, resid, (0.0, 0)).cpu() == 2.0)


@pytest.mark.parametrize("N", [64, 197, 288, 600])
def test_attention_dropout_mask_is_shared_by_forward_and_backward(hip, N):
    """q = k = 0 makes P uniform (1/N); one-hot V rows expose (P o mask)[q, k] for k < 64 in the
    forward output, and dO = 1 exposes the same mask's column sums in dV."""
    from vitamd import ops
    B, H, p, seed = 2, 2, 0.3, 777
    D = H * 64
    qkv = torch.zeros(B * N, 3 * D)
    v = qkv.view(B, N, 3, H, 64)[:, :, 2]                      # [B, N, H, 64]
    for k in range(min(N, 64)):
        v[:, k, :, k] = 1.0
    qd = qkv.to("cuda", BF16)
    o, lse = ops.attention_fwd(qd, B, N, H, False, dropout=(p, seed))
    o = o.float().cpu().view(B, N, H, 64)                       # o[b, q, h, k] = mask(b,h,q,k) / ((1-p) N)
    kept = o != 0
    nk = min(N, 64)
    assert abs(1.0 - kept[..., :nk].float().mean().item() - p) < 0.02
    scale = 1.0 / ((1 - p) * N)
    assert torch.allclose(o[kept], torch.full_like(o[kept], scale), rtol=2e-2)
    # without dropout every probed probability is present
    o0, _ = ops.attention_fwd(qd, B, N, H, False)
    assert torch.all(o0.float().cpu().view(B, N, H, 64)[..., :nk] != 0)
    # backward: dV[b, k, h, :] = sum_q (P o mask)[q, k] * dO[q, :]  with dO = 1
    d_o = torch.ones(B * N, D, device="cuda", dtype=BF16)
    dqkv = ops.attention_bwd(qd, o.view(B * N, D).to("cuda", BF16), lse, d_o, B, N, H, False, dropout=(p, seed)).float().cpu()
    dv = dqkv.view(B, N, 3, H, 64)[:, :, 2]                    # [B, N(key), H, 64]
    want = kept[..., :nk].float().sum(1) * scale                # [B, H, nk]: sum over queries
    got = dv[:, :nk, :, 0].permute(0, 2, 1)                     # [B, H, nk]
    assert O.rel_l2(got, want) < 1e-2
    # a different seed in backward would NOT reproduce the forward's mask
    dqkv_bad = ops.attention_bwd(qd, o.view(B * N, D).to("cuda", BF16), lse, d_o, B, N, H, False, dropout=(p, seed + 5)).float().cpu()
    assert O.rel_l2(dqkv_bad.view(B, N, 3, H, 64)[:, :nk, 2, :, 0].permute(0, 2, 1), want) > 2e-2


def test_module_dropout_semantics(hip):
    import transformer as T
    import weights as W
    cfg0 = T.TransformerConfig(n_layers=2, n_heads=2, n_embd=128, block_size=40, dropout=0.0)
    cfgp = T.TransformerConfig(n_layers=2, n_heads=2, n_embd=128, block_size=40, dropout=0.2)
    sd = W.transformer_state(3, "", 2, 128)
    m0, mp = T.Transformer(cfg0), T.Transformer(cfgp)
    m0.load_state_dict(sd); mp.load_state_dict(sd)
    m0, mp = m0.cuda(), mp.cuda()
    x = W.normal(3, "x", (4, 40, 128)).cuda()
    y0 = m0(x)
    torch.manual_seed(1); ya = mp(x)
    torch.manual_seed(1); yb = mp(x)
    torch.manual_seed(2); yc = mp(x)
    assert torch.equal(ya, yb)                                   # torch.manual_seed makes dropout repeatable
    assert not torch.equal(ya, yc) and not torch.equal(ya, y0)
    assert 0.02 < O.rel_l2(ya.cpu(), y0.cpu()) < 1.0             # a perturbation, not garbage
    mp.eval()
    torch.manual_seed(1); ye = mp(x)
    assert not torch.equal(ye, y0)                               # SDPA dropout stays on in eval (reference quirk) ...
    assert O.rel_l2(ye.cpu(), y0.cpu()) < O.rel_l2(ya.cpu(), y0.cpu())   # ... but the MLP dropout is off
    # training step with dropout: gradients are finite and the backward is linear in dy (same masks both times)
    mp.train()
    xg = x.clone().requires_grad_(True)
    torch.manual_seed(5); y = mp(xg); y.sum().backward()
    g1 = [p.grad.clone() for p in mp.parameters()]; gx1 = xg.grad.clone()
    mp.zero_grad(); xg.grad = None
    torch.manual_seed(5); y = mp(xg); (3.0 * y).sum().backward()
    for a, b in zip(g1, [p.grad for p in mp.parameters()]):
        assert torch.isfinite(b).all() and O.rel_l2(b.cpu(), 3.0 * a.cpu()) < 2e-2
    assert O.rel_l2(xg.grad.cpu(), 3.0 * gx1.cpu()) < 2e-2
