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


# ------------------------------------------------------------------------------------------------------------------------------
# Round 2: training-mode drop rates on the blocks.py surface (reference blocks.py:118 proj_drop, :168-170 Mlp.drop, :124-152 DropPath)
def _mask(shape, p, seed, group=1):
    """the kernel's keep-scale tensor (1/(1-p) or 0) for a seed, read off a tensor of ones"""
    from vitamd import ops
    return ops.dropout(torch.ones(shape, device="cuda"), p, seed, group=group).cpu()


def test_dropout_kernel_masks(hip):
    from vitamd import ops
    x = torch.randn(64, 50, 128, device="cuda")
    y = ops.dropout(x, 0.3, 1234)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    nz = y != 0
    assert torch.allclose(y[nz], x[nz] / 0.7, rtol=1e-6)
    assert torch.equal(ops.dropout(x, 0.3, 1234), y) and not torch.equal(ops.dropout(x, 0.3, 1235), y)     # stateless: (seed, index) decides
    yb = ops.dropout(x.to(torch.bfloat16), 0.3, 1234)
    assert torch.equal((yb != 0), (y != 0))
    z = ops.dropout(x, 0.5, 99, group=50 * 128)                 # DropPath: one decision per sample
    per_sample = (z != 0).float().mean(dim=(1, 2))
    assert set(per_sample.tolist()) <= {0.0, 1.0} and 0 < per_sample.sum().item() < 64
    kept = per_sample.bool()
    assert torch.allclose(z[kept], 2.0 * x[kept], rtol=1e-6)
    assert torch.equal(ops.dropout(x, 0.0, 5), x)


def test_blocks_mlp_training_dropout_matches_masked_reference(hip):
    """blocks.Mlp in training mode with drop > 0 and a DropPath around it: forward and every gradient against an fp32 torch
    computation that uses the SAME three masks (re-generated from the seeds the module draws from torch's generator)."""
    import blocks as BK
    from vitamd.functions import new_seed
    torch.manual_seed(0)
    m = BK.Mlp(128, 256, drop=0.25).cuda().train()
    x = torch.randn(6, 9, 128, device="cuda", requires_grad=True)
    dy = torch.randn(6, 9, 128, device="cuda")
    p, pp = 0.25, 0.5
    torch.manual_seed(77)
    s_in, s_out, s_path = new_seed(), new_seed(), new_seed()
    torch.manual_seed(77)
    y = m(x, _drop_path=pp)
    (y * dy).sum().backward()
    k_in, k_out = _mask((54, 256), p, s_in).cuda(), _mask((54, 128), p, s_out).cuda()
    k_path = _mask((6, 9 * 128), pp, s_path, group=9 * 128).view(6, 9, 128).cuda()
    xr = x.detach().clone().requires_grad_(True)
    w1, b1, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias))
    h = torch.nn.functional.gelu(xr.view(54, 128) @ w1.t() + b1) * k_in
    yr = ((h @ w2.t() + b2) * k_out).view(6, 9, 128) * k_path
    (yr * dy).sum().backward()
    assert O.rel_l2(y.detach().cpu(), yr.detach().cpu()) < 1e-2
    assert O.rel_l2(x.grad.cpu(), xr.grad.cpu()) < 2e-2
    for got, want in ((m.fc1.weight.grad, w1.grad), (m.fc1.bias.grad, b1.grad), (m.fc2.weight.grad, w2.grad), (m.fc2.bias.grad, b2.grad)):
        assert O.rel_l2(got.cpu(), want.cpu()) < 2e-2
    m.eval()
    assert torch.equal(m(x.detach()), m(x.detach()))                                   # eval: deterministic, no mask


def test_blocks_attention_and_uvit_training_drops(hip):
    import blocks as BK
    from vitamd.functions import new_seed
    torch.manual_seed(1)
    att = BK.Attention(128, 2, qkv_bias=True, proj_drop=0.2).cuda()
    x = torch.randn(8, 21, 128, device="cuda")
    att.eval()
    y0 = att(x)
    att.train()
    torch.manual_seed(5)
    s_proj, s_path = new_seed(), new_seed()
    torch.manual_seed(5)
    y1 = att(x, _drop_path=0.5)
    k = _mask((8 * 21, 128), 0.2, s_proj).view(8, 21, 128).cuda() * _mask((8, 21 * 128), 0.5, s_path, group=21 * 128).view(8, 21, 128).cuda()
    assert O.rel_l2(y1.cpu(), (y0 * k).cpu()) < 1e-2                     # = the eval output under the two masks (bf16 rounding of the scaled values)
    # whole block: training with every rate on runs, differs from eval, gives finite gradients, and checkpointing reproduces it bit for bit
    blk = BK.UViTBlock(128, 2, qkv_bias=True, drop=0.1, drop_path=0.2).cuda().train()
    xb = torch.randn(8, 21, 128, device="cuda", requires_grad=True)
    torch.manual_seed(9)
    yb = blk(xb)
    yb.sum().backward()
    g_plain = {k_: p.grad.clone() for k_, p in blk.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g_plain.values()) and torch.isfinite(xb.grad).all()
    blk.eval()
    assert O.rel_l2(yb.detach().cpu(), blk(xb.detach()).cpu()) > 1e-2
    blk.train()
    blk.use_checkpoint = True
    blk.zero_grad(set_to_none=True)
    torch.manual_seed(9)
    yc = blk(xb)
    yc.sum().backward()
    assert torch.equal(yc, yb)
    for k_, p in blk.named_parameters():
        assert O.rel_l2(p.grad.cpu(), g_plain[k_].cpu()) < 1e-5, k_          # (LayerNorm gamma / beta gradients are atomic sums: order differs)
    # DropPath module alone
    dp = BK.DropPath(0.5).cuda().train()
    z = dp(x)
    frac = (z.flatten(1).abs().sum(1) != 0).float().mean().item()
    assert 0.0 < frac < 1.0
    dp.eval()
    assert torch.equal(dp(x), x)
