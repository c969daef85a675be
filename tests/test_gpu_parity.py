"""Parity of the drop-in modules (HIP path, through the C ABI) against
 (a) the golden vectors the real reference produced (tests/golden, fp32 CPU), and
 (b) the CPU oracle on the same seeded inputs (fp32 and bf16-flow emulation).

Tolerance protocol (north_star: 1e-3 bf16 relative): per layer the bf16 path is compared with the
oracle's bf16-flow emulation (same rounding points, fp32 accumulation) at 4e-3 rel-L2 on outputs and
1e-2 on gradients; against the fp32 goldens the bar is the reference's OWN bf16-autocast deviation
from fp32 recorded in each fixture (x2 margin) — deeper stacks cannot beat bf16 itself."""
import pytest
import torch

import vit_oracle as O
import weights as W
from conftest import load_golden

pytestmark = pytest.mark.gpu
STRIDE = 997


def _sample(t):
    return t.detach().flatten()[::STRIDE].float().cpu()


def _err(got, ref):
    """rel-L2 against a summarised golden tensor: the full tensor when the fixture holds it,
    else its every-997th-element sample."""
    if "full" in ref:
        return O.rel_l2(got.cpu(), ref["full"])
    return O.rel_l2(_sample(got), ref["sample"])


def _build_transformer(c, sd):
    import transformer as T
    cfg = T.TransformerConfig(n_layers=c["n_layers"], n_heads=c["n_heads"], n_embd=c["n_embd"], block_size=c["seq"], causal=c["causal"])
    m = T.Transformer(cfg)
    m.load_state_dict(sd, strict=True)   # reference key names
    return m.cuda()


def _run_transformer(g):
    c = g["cfg"]
    sd = W.transformer_state(c["seed"], "", c["n_layers"], c["n_embd"], causal_block=c["seq"] if c["causal"] else None)
    m = _build_transformer(c, sd)
    x = W.normal(c["seed"], "x", (c["batch"], c["seq"], c["n_embd"])).cuda().requires_grad_(True)
    dy = W.normal(c["seed"], "dy", (c["batch"], c["seq"], c["n_embd"])).cuda()
    y = m(x)
    (y * dy).sum().backward()
    torch.cuda.synchronize()
    return y.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()}, sd, m


@pytest.mark.parametrize("name", ["transformer_tiny.pt", "transformer_tiny_causal.pt"])
def test_transformer_tiny_vs_reference_golden(hip, name):
    g = load_golden(name)
    y, dx, grads, sd, m = _run_transformer(g)
    assert sorted(m.state_dict().keys()) == g["state_keys"]
    floor = g["ref_bf16_floor"]
    assert O.rel_l2(y, g["y"]) < 2 * floor["y"] + 1e-3
    assert O.rel_l2(dx, g["dx"]) < 2 * floor["dx"] + 2e-3
    for k, ref in g["grads"].items():
        assert O.rel_l2(grads[k], ref) < 2 * floor["grads"][k] + 3e-3, k
    # tight check against the oracle's bf16-flow emulation
    c = g["cfg"]
    x = W.normal(c["seed"], "x", (c["batch"], c["seq"], c["n_embd"])).requires_grad_(True)
    dyv = W.normal(c["seed"], "dy", (c["batch"], c["seq"], c["n_embd"]))
    leaves = {k: v.clone().requires_grad_("mask" not in k) for k, v in sd.items()}
    yo = O.transformer(x, leaves, "", c["n_layers"], c["n_heads"], c["causal"], lowp=True)
    names = [k for k in leaves if "mask" not in k]
    go = torch.autograd.grad((yo * dyv).sum(), [x] + [leaves[k] for k in names])
    assert O.rel_l2(y, yo) < 3.8e-4
    assert O.rel_l2(dx, go[0]) < 2.0e-3
    for k, gk in zip(names, go[1:]):
        assert O.rel_l2(grads[k], gk) < 5.7e-3, k


def test_deferred_residual_is_bit_identical_to_the_fused_epilogue(hip):
    """functions.DEFER_RESID (round 3): inside a stack fc2 writes bf16 y and the NEXT layer's first LayerNorm forms x2 = x1 + y - claimed to be the
    very fp32 + bf16 sum the fused fc2 epilogue formed.  Pinned here: output, input gradient and every parameter gradient of a 2-layer stack
    are torch.equal with the switch on and off (weight gradients come from the bitwise-reproducible split-K reduce pass, bias gradients from atomics
    whose order may differ: those to 1e-6)."""
    from vitamd import functions as F
    g = load_golden("transformer_tiny.pt")
    keep = F.DEFER_RESID
    try:
        runs = {}
        for flag in (True, False):
            F.DEFER_RESID = flag
            y, dx, grads, _, _ = _run_transformer(g)
            runs[flag] = (y, dx, grads)
    finally:
        F.DEFER_RESID = keep
    assert torch.equal(runs[True][0], runs[False][0]) and torch.equal(runs[True][1], runs[False][1])
    for k in runs[True][2]:
        a, b = runs[True][2][k], runs[False][2][k]
        assert torch.equal(a, b) if a.dim() == 2 else O.rel_l2(a, b) < 1e-6, k


def test_transformer_layer_b_vs_reference_golden(hip):
    g = load_golden("transformer_layer_b.pt")
    y, dx, grads, _, _ = _run_transformer(g)
    floor = g["ref_bf16_floor"]
    assert O.rel_l2(_sample(y), g["y"]["sample"]) < 2 * floor["y"] + 1e-3
    assert O.rel_l2(y[0, 0], g["y_row0"]) < 2 * floor["y"] + 2e-3
    assert O.rel_l2(_sample(dx), g["dx"]["sample"]) < 2 * floor["dx"] + 2e-3
    for k, ref in g["grads"].items():
        assert _err(grads[k], ref) < 2 * floor["grads"][k] + 4e-3, k
        assert abs(float(grads[k].double().norm()) - ref["norm"]) / ref["norm"] < 1e-2, k


def _run_classifier(g):
    import train_vit as TV
    c = g["cfg"]
    cfg = TV.ViTConfig(c["image_size"], 3, c["patch"], c["preset"], c["extra_tokens"], 0.0)
    m = TV.ViTClassifier(cfg, num_classes=c["num_classes"])
    sd = W.classifier_state(c["seed"], 3, c["patch"], c["n_patches"], c["extra_tokens"], c["n_layers"], c["n_embd"], c["num_classes"])
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    images = W.normal(c["seed"], "images", (c["batch"], 3, c["image_size"], c["image_size"])).cuda()
    labels = W.randint(c["seed"], "labels", (c["batch"],), c["num_classes"]).cuda()
    logits = m(images)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().cpu(), float(loss.detach()), {k: p.grad.cpu() for k, p in m.named_parameters()}, m


@pytest.mark.parametrize("name", ["vit_s32.pt", "vit_b224.pt"])  # BASELINE configs[0] and the configs[1] shape
def test_classifier_vs_reference_golden(hip, name):
    g = load_golden(name)
    logits, loss, grads, m = _run_classifier(g)
    assert sorted(m.state_dict().keys()) == g["state_keys"]
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]
    floor = g["ref_bf16_floor"]
    assert O.rel_l2(logits, g["logits"]) < 2 * floor["logits"] + 2e-3
    assert abs(loss - g["loss"]) < 2 * floor["loss_abs"] + 2e-3
    worst = 0.0
    for k, ref in g["grads"].items():
        e = _err(grads[k], ref)
        worst = max(worst, e / (2 * floor["grads"][k] + 5e-3))
        assert e < 2 * floor["grads"][k] + 5e-3, (k, e, floor["grads"][k])
    for k, ref in g.get("full_grads", {}).items():
        assert O.rel_l2(grads[k], ref) < 2 * floor["grads"][k] + 5e-3, k


def test_standalone_attention_and_layer_modules(hip):
    """Attention / TransformerLayer used on their own (reference transformer.py:16-45 surface)."""
    import transformer as T
    cfg = T.TransformerConfig(n_layers=1, n_heads=2, n_embd=128, block_size=50)
    sd = W.transformer_state(5, "", 1, 128)
    layer = T.TransformerLayer(cfg)
    layer.load_state_dict({k[len("layers.0."):]: v for k, v in sd.items()})
    layer = layer.cuda()
    x = W.normal(5, "x", (2, 50, 128))
    xg = x.cuda().requires_grad_(True)
    y = layer(xg)
    y.sum().backward()
    xo = x.clone().requires_grad_(True)
    yo = O.transformer_layer(xo, sd, "layers.0.", 2, False, lowp=True)
    yo.sum().backward()
    assert O.rel_l2(y.detach().cpu(), yo.detach()) < 1.8e-5
    assert O.rel_l2(xg.grad.cpu(), xo.grad) < 1.7e-3
    attn = layer.multi_attn
    xa = O.layer_norm(x)
    ya = attn(xa.cuda())
    yao = O.attention(xa, sd, "layers.0.multi_attn.", 2, False, lowp=True)
    assert ya.shape == (2, 50, 128) and O.rel_l2(ya.detach().cpu(), yao) < 1.0e-6


def test_full_size_properties(hip):
    """BASELINE configs[1] full size (batch 256): size-independent checks — linearity of the
    backward in dy, and batch independence (sample i's output does not depend on the others)."""
    import train_vit as TV
    torch.manual_seed(0)
    cfg = TV.ViTConfig(224, 3, 16, "B", 1, 0.0)
    m = TV.ViTClassifier(cfg).cuda()
    images = torch.randn(256, 3, 224, 224, device="cuda")
    with torch.no_grad():
        full = m(images)
        part = m(images[64:96].contiguous())
    assert torch.isfinite(full).all()
    assert O.rel_l2(full[64:96].cpu(), part.cpu()) < 1e-6   # same kernels, same per-sample arithmetic
    labels = torch.randint(0, 1000, (256,), device="cuda")
    loss = torch.nn.functional.cross_entropy(m(images), labels)
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    (2.0 * torch.nn.functional.cross_entropy(m(images), labels)).backward()
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert O.rel_l2(p.grad.cpu(), 2.0 * g1[k].cpu()) < 1.4e-6, k


def test_adamw_kernel_matches_torch(hip):
    from vitamd.optim import AdamW
    torch.manual_seed(3)
    shapes = [(768, 768), (3072,), (5, 7, 3), (1,)]
    ps = [torch.randn(s) for s in shapes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    o_ref = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    o_mine = AdamW(mine, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for step in range(5):
        for r, m in zip(ref, mine):
            g = torch.randn(r.shape)
            r.grad, m.grad = g.clone(), g.clone().cuda()
        o_ref.step(); o_mine.step()
    for r, m in zip(ref, mine):
        assert O.rel_l2(m.detach().cpu(), r.detach()) < 1e-6


def test_training_steps_match_reference_loop(hip):
    """BASELINE configs[0] plumbing: 4 full optimiser steps of the reference loop body
    (train_vit.py:99-107) on a fixed batch — reference fp32 CPU losses vs HIP path + fused AdamW + our
    LR scheduler.  The loss trajectory pins forward, backward, optimiser and schedule together."""
    import train_vit as TV
    import utils as U
    from vitamd.optim import AdamW
    g = load_golden("train_steps_s32.pt")
    c = g["cfg"]
    cfg = TV.ViTConfig(32, 3, 16, "S", 1, 0.0)
    tc = cfg.trans_config
    m = TV.ViTClassifier(cfg, num_classes=c["num_classes"])
    m.load_state_dict(W.classifier_state(c["seed"], 3, 16, cfg.n_patches, 1, tc.n_layers, tc.n_embd, c["num_classes"]))
    m = m.cuda()
    images = W.normal(c["seed"], "images", (c["batch"], 3, 32, 32)).cuda()
    labels = W.randint(c["seed"], "labels", (c["batch"],), c["num_classes"]).cuda()
    optim = AdamW(m.parameters(), lr=c["lr"], weight_decay=c["weight_decay"])
    sched = U.get_lr_scheduler(optim, c["warmup"], c["train_steps"], c["min_lr"])
    losses = [float(TV.train_step(m, images, labels, optim, sched)) for _ in range(c["steps"])]
    ref = g["losses"].tolist()
    assert abs(losses[0] - ref[0]) < 5e-3
    for a, b in zip(losses, ref):
        assert abs(a - b) < 3e-2 * max(1.0, abs(b)), (losses, ref)
    assert O.rel_l2(m.head.bias.detach().cpu(), g["final_head_bias"]) < 7.3e-4


# ------------------------------------------------------------------ tokenizers (BASELINE configs[3], [4])
def _tokenizer_model(name):
    import weights as W2
    from test_oracle import TOKENIZERS
    g = load_golden(name)
    c, t = g["cfg"], TOKENIZERS[name]
    n_img_patches = (c["image_size"] // c["patch"]) ** 2
    sd = W2.tokenizer_state(c["seed"], t["enc"], t["quant"], t["dec"], n_img_patches, c["latent_tokens"], t["enc_extra"], t["dec_extra"],
                            c["patch"], c["n_layers"], c["n_embd"], c["codebook_size"], c["latent_dim"])
    if name.startswith("titok"):
        import train_titok as TT
        m = TT.TiTok(TT.TiTokConfig(c["image_size"], c["patch"], c["latent_tokens"], c["codebook_size"], c["latent_dim"], c["preset"]))
        enc = m.enc
    else:
        import train_vit_vqgan as TQ
        m = TQ.ViTVQGAN(TQ.ViTVQGANConfig(c["image_size"], c["patch"], c["codebook_size"], c["latent_dim"], c["preset"]))
        enc = m.encoder
    assert sorted(m.state_dict().keys()) == g["state_keys"]
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == g["state_shapes"]
    m.load_state_dict(sd, strict=True)
    images = W2.uniform(c["seed"], "images", (c["batch"], 3, c["image_size"], c["image_size"]), 0.5) + 0.5
    return g, m.cuda(), enc, images.cuda()


@pytest.mark.parametrize("name", ["titok_s256.pt", "vitvqgan_b256.pt"])
def test_tokenizer_vs_reference_golden(hip, name):
    g, m, enc, images = _tokenizer_model(name)
    floor = g["ref_bf16_floor"]
    with torch.no_grad():
        latents = enc(images)
        recon_fixed = m.decode_indices(g["indices"].cuda())     # decoder alone, on the reference's own code sequence
    assert O.rel_l2(latents.cpu(), g["latents"]) < 5.7e-3          # latents are 12-dim projections of a 6/12-layer bf16 stack
    assert _err(recon_fixed, g["recon_from_indices"]) < 2 * floor["recon"] + 3e-3
    recon, idx, qloss = m(images)
    agree = float((idx.cpu() == g["indices"]).float().mean())
    assert agree >= min(floor["index_agreement"], 1.0) - 0.03, agree   # nearest-code decisions may flip on near-ties
    assert abs(float(qloss) - g["quantize_loss"]) < 2e-3
    loss = torch.nn.functional.mse_loss(recon, images) + qloss
    assert abs(float(loss.detach()) - g["loss"]) < 5e-3
    loss.backward()
    torch.cuda.synchronize()
    bad = []
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        if p.grad is None or ref["norm"] == 0.0 or p.numel() == 0:
            continue
        assert torch.isfinite(p.grad).all(), k
        e = _err(p.grad, ref)
        if e > 2 * floor["grads"][k] + 1e-2:          # (round 4: tightened from 3 x floor + 2e-2; worst measured 5.2e-2 where the reference's own autocast floor is 3.6e-2)
            bad.append((k, round(e, 4), round(floor["grads"][k], 4)))
    assert not bad, bad[:8]


@pytest.mark.parametrize("preset,seq,batch", [("L", 65, 2), ("S", 288, 2), ("B", 256, 1), ("B", 577, 1)])   # 577 = ViT-B/16 at 384 px: long-sequence attention
def test_single_layer_presets_vs_oracle(hip, preset, seq, batch):
    """One layer of each reference preset (transformer.py:56-58) at the sequence lengths the tokenizers
    use (288 = TiTok, 256 = ViT-VQGAN) against the oracle's bf16-flow emulation."""
    import transformer as T
    L_, H, D = O.PRESETS[preset]
    sd = W.transformer_state(40 + seq, "", 1, D)
    cfg = T.TransformerConfig(n_layers=1, n_heads=H, n_embd=D, block_size=seq)
    m = T.Transformer(cfg)
    m.load_state_dict(sd)
    m = m.cuda()
    x = W.normal(40 + seq, "x", (batch, seq, D))
    dy = W.normal(40 + seq, "dy", (batch, seq, D))
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    (y * dy.cuda()).sum().backward()
    xo = x.clone().requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = O.transformer(xo, leaves, "", 1, H, False, lowp=True)
    go = torch.autograd.grad((yo * dy).sum(), [xo] + list(leaves.values()))
    assert O.rel_l2(y.detach().cpu(), yo.detach()) < 7.0e-4
    assert O.rel_l2(xg.grad.cpu(), go[0]) < 6.4e-4
    for (k, p), gk in zip(m.named_parameters(), go[1:]):
        assert O.rel_l2(p.grad.cpu(), gk) < 5.6e-3, k


# ------------------------------------------------------------------ blocks.py surface
BLOCK_CTORS = {
    "rab": ("ResidualAttentionBlock", dict(d_model=128, n_head=2)),
    "rab_nomlp": ("ResidualAttentionBlock", dict(d_model=128, n_head=2, mlp_ratio=0)),
    "uvit_skip": ("UViTBlock", dict(dim=128, num_heads=2, skip=True)),
    "uvit_bias": ("UViTBlock", dict(dim=128, num_heads=2, qkv_bias=True)),
    "attn": ("Attention", dict(dim=128, num_heads=2, qkv_bias=True)),
    "mlp": ("Mlp", dict(in_features=128, hidden_features=512)),
}


@pytest.mark.parametrize("name", list(BLOCK_CTORS))
def test_blocks_surface_vs_reference_golden(hip, name):
    import blocks as BK
    from test_oracle import blocks_oracle_run
    case = load_golden("blocks_tiny.pt")[name]
    cls, kw = BLOCK_CTORS[name]
    m = getattr(BK, cls)(**kw)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == case["shapes"]       # checkpoint-key contract
    m.load_state_dict(W.module_state(case["seed"], case["shapes"]), strict=True)
    m = m.cuda()
    L, N, D = 37, 3, 128
    lnd = name.startswith("rab")
    x = W.normal(case["seed"], "x", (L, N, D) if lnd else (N, L, D)).cuda().requires_grad_(True)
    dy = W.normal(case["seed"], "dy", tuple(x.shape)).cuda()
    args = [x]
    if name == "uvit_skip":
        args.append(W.normal(case["seed"], "skip", (N, L, D)).cuda().requires_grad_(True))
    y = m(*args)
    (y * dy).sum().backward()
    torch.cuda.synchronize()
    floor = case["ref_bf16_floor"]
    assert O.rel_l2(y.detach().cpu(), case["y"]) < 2 * floor["y"] + 2e-3
    assert O.rel_l2(x.grad.cpu(), case["dx"]) < 2 * floor["dx"] + 4e-3
    if "dskip" in case:
        assert O.rel_l2(args[1].grad.cpu(), case["dskip"]) < 5.5e-3
    for k, p in m.named_parameters():
        assert O.rel_l2(p.grad.cpu(), case["grads"][k]) < 2 * floor["grads"][k] + 6e-3, k
    # tight: against the oracle's bf16-flow emulation of the same block
    lo = blocks_oracle_run(name, case, lowp=True)
    assert O.rel_l2(y.detach().cpu(), lo["y"]) < 5.3e-4
    assert O.rel_l2(x.grad.cpu(), lo["dx"]) < 5.8e-3


def test_blocks_attention_bf16_sdpa_at_depth_vs_reference(hip, record_property):
    """VERDICT r1 'missing' item 4.  The reference's blocks.Attention upcasts q, k, v to fp32 before SDPA (blocks.py:100), so under
    autocast its attention runs in fp32; the build's attention kernels take bf16 operands.  A stack of 12 UViTBlocks (N = 197 tokens)
    against the reference's fp32 output, with the reference's OWN bf16-autocast deviation (fp32 SDPA inside) as the yardstick:
    measured 1.00x (output, 4.3e-3), 1.06x (input gradient, 5.1e-3), <= 1.32x (worst parameter gradient) of that floor: at 12 blocks
    deep the bf16 SDPA operands add nothing measurable to the deviation the bf16 Linear layers cause anyway."""
    import blocks as BK
    g = load_golden("blocks_depth.pt")
    c = g["case"]
    stack = []
    for i in range(c["depth"]):
        m = BK.UViTBlock(dim=c["dim"], num_heads=c["num_heads"], qkv_bias=c["qkv_bias"])
        m.load_state_dict(W.module_state(c["seed"] + i, g["shapes"]), strict=True)
        stack.append(m.cuda())
    x = W.normal(c["seed"], "x", (c["batch"], c["tokens"], c["dim"])).cuda().requires_grad_(True)
    dy = W.normal(c["seed"], "dy", tuple(x.shape)).cuda()
    h = x
    for m in stack:
        h = m(h)
    (h * dy).sum().backward()
    torch.cuda.synchronize()
    floor = g["ref_bf16_floor"]
    ey, edx = O.rel_l2(h.detach().cpu(), g["y"]), O.rel_l2(x.grad.cpu(), g["dx"])
    worst, worst_k = 0.0, None
    for i, m in enumerate(stack):
        for k, p in m.named_parameters():
            ref = g["grads"][f"{i}.{k}"]
            got = p.grad.flatten() if p.numel() <= 1024 else p.grad.flatten()[::13]      # vectors in full, matrices sampled (as the fixture holds them)
            e = O.rel_l2(got.cpu(), ref["sample"]) / max(floor["grads"][f"{i}.{k}"], 1e-4)
            if e > worst:
                worst, worst_k = e, f"{i}.{k}"
    record_property("depth12_y_over_floor", ey / floor["y"])
    record_property("depth12_dx_over_floor", edx / floor["dx"])
    record_property("depth12_worst_grad_over_floor", worst)
    print(f"depth-12 UViT stack: y {ey:.3e} ({ey / floor['y']:.2f}x floor), dx {edx:.3e} ({edx / floor['dx']:.2f}x), worst grad {worst:.2f}x floor at {worst_k}")
    assert ey < 1.5 * floor["y"]          # 1.5 x measured
    assert edx < 1.6 * floor["dx"]
    assert worst < 2.0, worst_k


# ------------------------------------------------------------------ blocks.py tokenizer wrappers
@pytest.mark.parametrize("name,cls", [("encoder", "TiTokEncoder"), ("decoder", "TiTokDecoder"), ("tatitok_decoder", "TATiTokDecoder")])
def test_block_tokenizers_vs_reference_golden(hip, name, cls):
    import blocks as BK
    from test_host import _tok_cfg
    from test_oracle import block_tokenizer_inputs, block_tokenizer_oracle_run
    g = load_golden("blocks_tokenizers.pt")
    cfg, B, case = g["config"], g["batch"], g[name]
    m = getattr(BK, cls)(_tok_cfg(cfg))
    m.load_state_dict(W.module_state(case["seed"], case["shapes"]), strict=True)
    m = m.cuda()
    ins = [t.cuda().requires_grad_(True) for t in block_tokenizer_inputs(name, case, cfg, B)]
    y = m(*ins)
    assert list(y.shape) == case["y"]["shape"]
    dy = W.normal(case["seed"], "dy", tuple(y.shape)).cuda()
    (y * dy).sum().backward()
    torch.cuda.synchronize()
    floor = case["ref_bf16_floor"]
    # against the reference's fp32 result, judged by the reference's own bf16-autocast deviation
    assert _err(y.detach(), case["y"]) < 2 * floor["y"] + 2e-3
    for t, ref, fl in zip(ins, case["dinputs"], floor["dinputs"]):
        assert _err(t.grad, ref) < 2 * fl + 5e-3
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert _err(p.grad, case["grads"][k]) < 2 * floor["grads_max"] + 6e-3, k
    lo_y, lo_dins, lo_grads = block_tokenizer_oracle_run(name, case, cfg, B, lowp=True)
    # tighter: against the oracle's bf16-flow emulation (8 layers deep: rounding-order differences accumulate)
    assert O.rel_l2(y.detach().cpu(), lo_y) < 1e-2
    for t, ref in zip(ins, lo_dins):
        assert O.rel_l2(t.grad.cpu(), ref) < 1.8e-2


@pytest.mark.parametrize("name", ["vq_plain", "vq_l2norm", "vq_wide", "vq_cluster"])
def test_vector_quantizer_vs_reference_golden(hip, name):
    import blocks as BK
    from test_oracle import vq_case_tensors
    case = load_golden("blocks_tokenizers.pt")[name]
    code, z, dy = vq_case_tensors(case)
    vq = BK.VectorQuantizer(**case["kwargs"])
    vq.load_state_dict({"embedding.weight": code}, strict=False)
    vq = vq.cuda().train()
    z = z.cuda().requires_grad_(True)
    zq, res = vq(z)
    ((zq * dy.cuda()).sum() + res["quantizer_loss"]).backward()
    torch.cuda.synchronize()
    # fp32 arithmetic: indices bit-exact with the reference, values to fp32 rounding
    assert torch.equal(res["min_encoding_indices"].cpu(), case["indices"])
    assert abs(float(res["quantizer_loss"]) - case["quantizer_loss"]) < 1e-6
    assert abs(float(res["commitment_loss"]) - case["commitment_loss"]) < 1e-7 and abs(float(res["codebook_loss"]) - case["codebook_loss"]) < 1e-7
    assert _err(zq.detach(), case["zq"]) < 1e-6 and _err(z.grad, case["dz"]) < 1e-5
    assert _err(vq.embedding.weight.grad, case["dcodebook"]) < 1e-5
    if case["kwargs"].get("clustering_vq"):
        assert _err(vq.embedding.weight.detach(), case["codebook_after"]) < 1e-6
        assert torch.allclose(vq.embed_prob.cpu(), case["embed_prob_after"], atol=1e-7)


def test_conv3x3_kernel_vs_oracle(hip):
    """The decoders' 3x3 conv_out (reference blocks.py:333): fp32 kernel against the oracle's shifted-sum restatement,
    forward, input gradient, weight and bias gradient; odd sizes exercise the zero padding and the grid-stride tails."""
    from vitamd import ops
    for (B, H, Wd) in ((2, 5, 7), (3, 64, 48), (1, 256, 256)):
        x = W.normal(5, f"x{H}", (B, 3, H, Wd)); w = W.normal(5, "w", (3, 3, 3, 3), 0.3); b = W.normal(5, "b", (3,))
        dy = W.normal(5, f"dy{H}", (B, 3, H, Wd))
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = O.conv3x3_same(xr, wr, br)
        gx, gw, gb = torch.autograd.grad((yr * dy).sum(), [xr, wr, br])
        y = ops.conv3x3_fwd(x.cuda(), w.cuda(), b.cuda())
        dx, dw, db = ops.conv3x3_bwd(x.cuda(), w.cuda(), dy.cuda())
        torch.cuda.synchronize()
        assert O.rel_l2(y.cpu(), yr.detach()) < 1e-6
        assert O.rel_l2(dx.cpu(), gx) < 1e-6
        assert O.rel_l2(dw.cpu(), gw) < 2.4e-6 and O.rel_l2(db.cpu(), gb) < 2e-5
    from vitamd.lib import VitamdError
    with pytest.raises(VitamdError):
        ops.conv3x3_fwd(torch.zeros(1, 4, 8, 8, device="cuda"), torch.zeros(3, 4, 3, 3, device="cuda"), None)


def test_vq_nearest_wide_codes(hip):
    """d > 64 path of the nearest-code kernel against a float64 brute force (first-minimum tie-break; duplicates planted)."""
    from vitamd import ops
    for (M, K, d) in ((37, 300, 65), (513, 1024, 256), (9, 70, 1024)):
        x = W.normal(9, f"x{d}", (M, d)); e = W.normal(9, f"e{d}", (K, d))
        e[K // 2] = e[3]                                    # exact duplicate: the lower index must win
        x[0] = e[3]
        idx = ops.vq_nearest(x.cuda(), e.cuda()).cpu()
        d2 = (x.double().unsqueeze(1) - e.double().unsqueeze(0)).pow(2).sum(-1)
        ref = d2.argmin(dim=1)
        assert idx[0] == 3
        bad = (idx != ref).nonzero().flatten()
        for m in bad.tolist():                              # fp32-vs-fp64 near ties only
            assert abs(float(d2[m, idx[m]] - d2[m, ref[m]])) < 1e-4 * float(d2[m, ref[m]])
        assert len(bad) <= max(1, M // 100)


def test_vq_nearest_narrow_codes_cut_over_the_codebook(hip):
    """d <= 64 with more than 256 codes (TiTok / ViT-VQGAN: 2 048 x 12): (row blocks) x (code chunks) workgroups folded with a 64-bit atomicMin of
    (distance bits, index).  Integer-valued data make every distance exact, so the result must equal the exact first-minimum argmin - duplicates of
    the winning code are planted in EARLIER and LATER chunks - and a second call must give the same bits; random data against float64 with a
    near-tie allowance; K <= 256 still takes the single-pass kernel."""
    from vitamd import ops
    g = torch.Generator().manual_seed(12)
    for (M, K, d) in ((8192, 2048, 12), (777, 1000, 16), (300, 2048, 33), (50, 256, 12)):
        x = torch.randint(-4, 5, (M, d), generator=g).float()
        e = torch.randint(-4, 5, (K, d), generator=g).float()
        e[K - 1] = e[5]; e[K // 2] = e[5]; x[0] = e[5]; x[1] = e[K // 2]      # exact duplicates across chunks: the lowest index must win
        d2 = (x.unsqueeze(1) - e.unsqueeze(0)).pow(2).sum(-1)                # exact in fp32 (integers < 2^24)
        ref = d2.argmin(dim=1)                                               # torch returns the first minimum
        first = (d2 == d2.min(dim=1, keepdim=True).values).float().argmax(dim=1)
        assert torch.equal(ref, first)
        i1 = ops.vq_nearest(x.cuda(), e.cuda()).cpu()
        i2 = ops.vq_nearest(x.cuda(), e.cuda()).cpu()
        assert torch.equal(i1, ref) and torch.equal(i1, i2) and int(i1[0]) <= 5 and int(i1[1]) <= 5, (M, K, d)
    x = W.normal(10, "xn", (4096, 12)); e = W.normal(10, "en", (2048, 12))
    idx = ops.vq_nearest(x.cuda(), e.cuda()).cpu()
    d2 = (x.double().unsqueeze(1) - e.double().unsqueeze(0)).pow(2).sum(-1)
    ref = d2.argmin(dim=1)
    bad = (idx != ref).nonzero().flatten().tolist()
    for m in bad:
        assert abs(float(d2[m, idx[m]] - d2[m, ref[m]])) < 1e-5 * float(d2[m, ref[m]])
    assert len(bad) <= 4


def test_blocks_drop_rates_are_identity_in_eval(hip):
    """Reference semantics: nn.Dropout / DropPath do nothing in eval mode (blocks.py:124-139).  A block built WITH drop rates
    must reproduce the reference's (drop-free) golden output in eval mode; training with the rates on is covered by
    tests/test_gpu_dropout.py (masks, masked-reference parity, checkpointing)."""
    import blocks as BK
    case = load_golden("blocks_tiny.pt")["uvit_bias"]
    m = BK.UViTBlock(dim=128, num_heads=2, qkv_bias=True, drop=0.1, attn_drop=0.1, drop_path=0.2)
    m.load_state_dict(W.module_state(case["seed"], case["shapes"]), strict=True)
    m = m.cuda().eval()
    x = W.normal(case["seed"], "x", (3, 37, 128)).cuda()
    with torch.no_grad():
        y = m(x)
    assert O.rel_l2(y.cpu(), case["y"]) < 2 * case["ref_bf16_floor"]["y"] + 2e-3
    m.train()
    assert torch.isfinite(m(x)).all()


def test_graphed_step_matches_eager(hip):
    """hipGraph capture of forward + loss + backward (vitamd/graph.py) on BASELINE configs[0] (ViT-S, 32x32, batch 64):
    replays on NEW inputs must reproduce the eager step on those inputs (atomics reorder fp32 sums: 1e-5, not bitwise)."""
    import train_vit as TV
    from vitamd.graph import GraphedStep
    torch.manual_seed(0)
    m = TV.ViTClassifier(TV.ViTConfig(32, 3, 16, "S", 1, 0.0), num_classes=10).cuda()
    ce = torch.nn.functional.cross_entropy
    xs = [W.normal(90 + i, "x", (64, 3, 32, 32)).cuda() for i in range(3)]
    ys = [W.randint(90 + i, "y", (64,), 10).cuda() for i in range(3)]
    step = GraphedStep(m, ce, xs[0], ys[0])
    for i in (1, 2, 0):
        loss_g = float(step(xs[i], ys[i]))
        got = {k: p.grad.clone() for k, p in m.named_parameters()}
        m.zero_grad(set_to_none=True)
        loss_e = ce(m(xs[i]), ys[i]); loss_e.backward()
        assert abs(loss_g - float(loss_e)) < 1e-6
        for k, p in m.named_parameters():
            assert O.rel_l2(got[k].cpu(), p.grad.cpu()) < 1.0e-6, k
    from vitamd.lib import VitamdError
    with pytest.raises(VitamdError):
        step(xs[0][:32], ys[0][:32])


# ------------------------------------------------------------------------------------------------------------------------------
# Round 2: checks that do not pass through the oracle's kernel-shaped `lowp` emulation
def test_vs_reference_own_bf16_autocast_outputs(hip):
    """HIP path against the REFERENCE's own bf16-autocast tensors (tests/golden/ref_autocast_bf16.pt, written by
    oracle/gen_golden.py from the unmodified reference under torch.autocast(cpu, bf16)).  Two independent bf16 flows of the same
    fp32 function differ by about sqrt(2) x the bf16 floor of each, so the bound is 2 x the reference's recorded floor (its own
    bf16-vs-fp32 distance) + a small constant; the measured values land in parity_errors.json."""
    ref = load_golden("ref_autocast_bf16.pt")
    for name in ("transformer_tiny", "transformer_layer_b"):
        g = load_golden(name + ".pt")
        floor = g["ref_bf16_floor"]
        y, dx, grads, _, _ = _run_transformer(g)
        r = ref[name]
        if name == "transformer_tiny":
            assert O.rel_l2(y, r["y"]) < 2 * floor["y"] + 1e-3
            assert O.rel_l2(dx, r["dx"]) < 2 * floor["dx"] + 2e-3
            for k, want in r["grads"].items():
                assert O.rel_l2(grads[k], want) < 2 * floor["grads"][k] + 3e-3, k
        else:
            assert O.rel_l2(_sample(y), r["y"]["sample"]) < 2 * floor["y"] + 1e-3
            assert O.rel_l2(_sample(dx), r["dx"]["sample"]) < 2 * floor["dx"] + 2e-3
            for k, want in r["grads"].items():
                assert _err(grads[k], want) < 2 * floor["grads"][k] + 4e-3, k
    g = load_golden("vit_s32.pt")
    floor = g["ref_bf16_floor"]
    logits, loss, grads, _ = _run_classifier(g)
    r = ref["vit_s32"]
    assert O.rel_l2(logits, r["logits"]) < 2 * floor["logits"] + 2e-3
    assert abs(loss - r["loss"]) < 2 * floor["loss_abs"] + 2e-3
    for k, want in r["grads"].items():
        assert _err(grads[k], want) < 2 * floor["grads"][k] + 5e-3, k
    # the headline shape: 12-layer ViT-B/16 at 224 px, batch 2, against the reference's own autocast run
    g = load_golden("vit_b224.pt")
    floor = g["ref_bf16_floor"]
    logits, loss, grads, _ = _run_classifier(g)
    r = ref["vit_b224"]
    assert O.rel_l2(logits, r["logits"]) < 2 * floor["logits"] + 2e-3
    assert abs(loss - r["loss"]) < 2 * floor["loss_abs"] + 2e-3
    for k, want in r["grads"].items():
        assert _err(grads[k], want) < 2 * floor["grads"][k] + 5e-3, k


def test_reference_loop_autocast_and_gradscaler_are_transparent(hip):
    """The reference's loop runs the model under torch.autocast("cuda") (fp16) with a GradScaler (train_vit.py:84,100-106).
    The modules own their precision flow (custom_fwd(cast_inputs=fp32)): logits must be IDENTICAL to the plain call, and the
    unscaled gradients equal to it up to the exact power-of-two scale."""
    import train_vit as TV
    torch.manual_seed(3)
    m = TV.ViTClassifier(TV.ViTConfig(32, 3, 16, "S", 1, 0.0), num_classes=10).cuda()
    x, y = torch.randn(16, 3, 32, 32, device="cuda"), torch.randint(0, 10, (16,), device="cuda")
    logits = m(x)
    torch.nn.functional.cross_entropy(logits, y).backward()
    plain = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    scaler = torch.amp.GradScaler("cuda")
    with torch.autocast("cuda", dtype=torch.float16):
        logits_amp = m(x)
        loss_amp = torch.nn.functional.cross_entropy(logits_amp, y)
    scaler.scale(loss_amp).backward()
    scale = scaler.get_scale()
    assert logits_amp.dtype == torch.float32 and torch.equal(logits_amp, logits)
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert O.rel_l2((p.grad / scale).cpu(), plain[k].cpu()) < 1e-6, k
