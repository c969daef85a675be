"""Pins the CPU oracle (oracle/vit_oracle.py) to the golden vectors that
oracle/gen_golden.py produced from the real reference (fp32, CPU)."""
import torch

import vit_oracle as O
import weights as W
from conftest import load_golden

STRIDE = 997


def _sample(t):
    return t.detach().flatten()[::STRIDE].float()


def _err(got, ref):
    return O.rel_l2(got, ref["full"]) if "full" in ref else O.rel_l2(_sample(got), ref["sample"])


def _run_transformer(g, lowp=False):
    c = g["cfg"]
    sd = W.transformer_state(c["seed"], "", c["n_layers"], c["n_embd"], causal_block=c["seq"] if c["causal"] else None)
    x = W.normal(c["seed"], "x", (c["batch"], c["seq"], c["n_embd"])).requires_grad_(True)
    dy = W.normal(c["seed"], "dy", (c["batch"], c["seq"], c["n_embd"]))
    leaves = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "mask" not in k) for k, v in sd.items()}
    y = O.transformer(x, leaves, "", c["n_layers"], c["n_heads"], c["causal"], lowp)
    names = [k for k in leaves if "mask" not in k]
    grads = torch.autograd.grad((y * dy).sum(), [x] + [leaves[k] for k in names])
    return y.detach(), grads[0], dict(zip(names, grads[1:])), sd


def test_transformer_tiny_matches_reference():
    for name in ("transformer_tiny.pt", "transformer_tiny_causal.pt"):
        g = load_golden(name)
        y, dx, grads, sd = _run_transformer(g)
        assert sorted(sd.keys()) == g["state_keys"]  # checkpoint-key contract incl. causal mask buffer
        assert O.rel_l2(y, g["y"]) < 2e-6
        assert O.rel_l2(dx, g["dx"]) < 2e-6
        for k, ref in g["grads"].items():
            assert O.rel_l2(grads[k], ref) < 5e-6, k


def test_transformer_layer_b_matches_reference():
    g = load_golden("transformer_layer_b.pt")
    y, dx, grads, _ = _run_transformer(g)
    assert O.rel_l2(_sample(y), g["y"]["sample"]) < 2e-6
    assert abs(float(y.double().norm()) - g["y"]["norm"]) / g["y"]["norm"] < 1e-6
    assert O.rel_l2(y[0, 0], g["y_row0"]) < 2e-6
    assert O.rel_l2(_sample(dx), g["dx"]["sample"]) < 2e-6
    for k, ref in g["grads"].items():
        assert _err(grads[k], ref) < 1e-5, k
        assert abs(float(grads[k].double().norm()) - ref["norm"]) / ref["norm"] < 1e-5, k


def test_lowp_emulation_is_within_the_references_own_bf16_floor():
    """oracle(lowp) models the autocast dtype flow; its distance from fp32 must be of the
    same size as the reference's own CPU-autocast distance (recorded in the fixture)."""
    g = load_golden("transformer_tiny.pt")
    y, dx, grads, _ = _run_transformer(g, lowp=True)
    floor = g["ref_bf16_floor"]
    assert O.rel_l2(y, g["y"]) < 2.0 * floor["y"] + 1e-3
    assert O.rel_l2(dx, g["dx"]) < 2.0 * floor["dx"] + 1e-3


def _run_classifier(g, lowp=False):
    c = g["cfg"]
    cfg = O.OracleViTConfig(c["image_size"], 3, c["patch"], c["n_layers"], c["n_heads"], c["n_embd"], c["extra_tokens"])
    sd = W.classifier_state(c["seed"], 3, c["patch"], c["n_patches"], c["extra_tokens"], c["n_layers"], c["n_embd"], c["num_classes"])
    images = W.normal(c["seed"], "images", (c["batch"], 3, c["image_size"], c["image_size"]))
    labels = W.randint(c["seed"], "labels", (c["batch"],), c["num_classes"])
    return O.classifier_loss_and_grads(images, labels, sd, cfg, lowp), sd


def _check_classifier(name, tol):
    g = load_golden(name)
    (logits, loss, grads), sd = _run_classifier(g)
    assert sorted(sd.keys()) == g["state_keys"]
    assert {k: list(v.shape) for k, v in sd.items()} == g["state_shapes"]
    assert sum(v.numel() for v in sd.values()) == g["n_params"]
    assert O.rel_l2(logits, g["logits"]) < tol
    assert abs(float(loss) - g["loss"]) < tol * max(1.0, abs(g["loss"]))
    for k, ref in g["grads"].items():
        assert _err(grads[k], ref) < 20 * tol, k
        assert abs(float(grads[k].double().norm()) - ref["norm"]) / max(ref["norm"], 1e-30) < 20 * tol, k
    for k, ref in g.get("full_grads", {}).items():
        assert O.rel_l2(grads[k], ref) < 20 * tol, k


def test_vit_s32_classifier_matches_reference():   # BASELINE config 1 shape (S/16, 32x32, 10 classes, batch 64)
    _check_classifier("vit_s32.pt", 3e-6)


def test_vit_b224_classifier_matches_reference():  # BASELINE config 2 shape at batch 2
    g = load_golden("vit_b224.pt")
    assert g["n_params"] == 79_441_384 and len(g["state_keys"]) == 78
    _check_classifier("vit_b224.pt", 5e-6)


def test_lr_schedule_matches_reference_trace():
    g = load_golden("lr_schedule.pt")
    for s, lr in enumerate(g["lrs"].tolist()):
        mine = O.lr_at(s, g["base_lr"], g["warmup_steps"], g["train_steps"], g["min_lr"])
        assert abs(mine - lr) < 1e-12, (s, mine, lr)


# ------------------------------------------------------------------ tokenizers
TOKENIZERS = {
    "titok_s256.pt": dict(enc="enc.", quant="quant.", dec="dec.", enc_extra=32, dec_extra=256, keep_enc=32, keep_dec=256),
    "vitvqgan_b256.pt": dict(enc="encoder.", quant="quant.", dec="decoder.", enc_extra=0, dec_extra=0, keep_enc=None, keep_dec=None),
}


def tokenizer_setup(name):
    g = load_golden(name)
    c, t = g["cfg"], TOKENIZERS[name]
    heads = O.PRESETS[c["preset"]][1]
    n_img_patches = (c["image_size"] // c["patch"]) ** 2
    sd = W.tokenizer_state(c["seed"], t["enc"], t["quant"], t["dec"], n_img_patches, c["latent_tokens"], t["enc_extra"], t["dec_extra"],
                           c["patch"], c["n_layers"], c["n_embd"], c["codebook_size"], c["latent_dim"])
    enc_cfg = O.OracleViTConfig(c["image_size"], 3, c["patch"], c["n_layers"], heads, c["n_embd"], t["enc_extra"])
    dec_cfg = O.OracleViTConfig(c["latent_tokens"], c["n_embd"], 1, c["n_layers"], heads, c["n_embd"], t["dec_extra"], n_patches=c["latent_tokens"])
    images = W.uniform(c["seed"], "images", (c["batch"], 3, c["image_size"], c["image_size"]), 0.5) + 0.5
    return g, c, t, sd, enc_cfg, dec_cfg, images


def test_tokenizers_match_reference():
    for name in TOKENIZERS:
        g, c, t, sd, enc_cfg, dec_cfg, images = tokenizer_setup(name)
        assert sorted(sd.keys()) == g["state_keys"]
        assert {k: list(v.shape) for k, v in sd.items()} == g["state_shapes"]
        assert sum(v.numel() for v in sd.values()) == g["n_params"]
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        grid = c["image_size"] // c["patch"]
        recon, idx, qloss, latents = O.tokenizer_forward(images, leaves, t["enc"], t["quant"], t["dec"], enc_cfg, dec_cfg,
                                                         t["keep_enc"], t["keep_dec"], grid, c["patch"])
        assert O.rel_l2(latents, g["latents"]) < 1e-5
        assert torch.equal(idx, g["indices"])
        assert abs(float(qloss) - g["quantize_loss"]) < 1e-6
        assert _err(recon, g["recon"]) < 1e-5
        loss = ((recon - images) ** 2).mean() + qloss
        assert abs(float(loss) - g["loss"]) < 1e-6
        grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        for (k, _), gr in zip(leaves.items(), grads):
            ref = g["grads"][k]
            if ref["norm"] == 0.0:
                assert gr is None or gr.numel() == 0 or float(gr.abs().max()) == 0.0, k
            else:
                assert _err(gr, ref) < 2e-4, k


# ------------------------------------------------------------------ blocks.py surface
def blocks_oracle_run(name, case, lowp=False):
    sd = W.module_state(case["seed"], case["shapes"])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    L, N, D = 37, 3, 128
    lnd = name.startswith("rab")
    x = W.normal(case["seed"], "x", (L, N, D) if lnd else (N, L, D)).requires_grad_(True)
    dy = W.normal(case["seed"], "dy", tuple(x.shape))
    skip = None
    if name == "uvit_skip":
        skip = W.normal(case["seed"], "skip", (N, L, D)).requires_grad_(True)
    if lnd:
        y = O.residual_attention_block(x, leaves, 2, lowp)
    elif name.startswith("uvit"):
        y = O.uvit_block(x, leaves, 2, skip, lowp)
    elif name == "attn":
        y = O.attention_proj(x, leaves["qkv.weight"], leaves["qkv.bias"], leaves["proj.weight"], leaves["proj.bias"], 2, lowp)
    else:
        y = O.mlp2(x, leaves["fc1.weight"], leaves["fc1.bias"], leaves["fc2.weight"], leaves["fc2.bias"], lowp)
    wrt = [x] + ([skip] if skip is not None else []) + list(leaves.values())
    grads = torch.autograd.grad((y * dy).sum(), wrt)
    out = {"y": y.detach(), "dx": grads[0], "grads": dict(zip(leaves.keys(), grads[-len(leaves):]))}
    if skip is not None:
        out["dskip"] = grads[1]
    return out


def test_blocks_oracle_matches_reference():
    g = load_golden("blocks_tiny.pt")
    assert set(g) == {"rab", "rab_nomlp", "uvit_skip", "uvit_bias", "attn", "mlp"}
    for name, case in g.items():
        got = blocks_oracle_run(name, case)
        assert O.rel_l2(got["y"], case["y"]) < 3e-6, name
        assert O.rel_l2(got["dx"], case["dx"]) < 1e-5, name
        if "dskip" in case:
            assert O.rel_l2(got["dskip"], case["dskip"]) < 1e-5, name
        for k, ref in case["grads"].items():
            assert O.rel_l2(got["grads"][k], ref) < 2e-5, (name, k)


# ---------------------------------------------------------------- blocks.py tokenizer wrappers
def block_tokenizer_inputs(name, case, cfg, B):
    s, lat, ldim = case["seed"], cfg["latent_tokens"], cfg["latent_dim"]
    if name == "encoder":
        return [W.normal(s, "pixels", (B, 3, cfg["image_size"], cfg["image_size"])), W.normal(s, "latent_tokens", (lat, 512), 0.05)]
    ins = [W.normal(s, "zq", (B, ldim, 1, lat))]
    if name == "tatitok_decoder":
        ins.append(W.normal(s, "text", (B, cfg["text_context_length"], cfg["text_embed_dim"])))
    return ins


def block_tokenizer_oracle_run(name, case, cfg, B, lowp=False):
    sd = W.module_state(case["seed"], case["shapes"])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ins = [t.requires_grad_(True) for t in block_tokenizer_inputs(name, case, cfg, B)]
    patch, grid = cfg["patch_size"], cfg["image_size"] // cfg["patch_size"]
    if name == "encoder":
        y = O.titok_block_encoder(ins[0], ins[1], leaves, patch, grid, 8, 8, lowp)
    else:
        y = O.titok_block_decoder(ins[0], leaves, patch, grid, 8, 8, text_guidance=ins[1] if len(ins) > 1 else None, lowp=lowp)
    dy = W.normal(case["seed"], "dy", tuple(y.shape))
    names = list(leaves)
    grads = torch.autograd.grad((y.float() * dy).sum(), ins + [leaves[k] for k in names])
    return y.detach(), list(grads[:len(ins)]), dict(zip(names, grads[len(ins):]))


def test_block_tokenizers_oracle_matches_reference():
    g = load_golden("blocks_tokenizers.pt")
    cfg, B = g["config"], g["batch"]
    for name in ("encoder", "decoder", "tatitok_decoder"):
        case = g[name]
        y, dins, grads = block_tokenizer_oracle_run(name, case, cfg, B)
        assert list(y.shape) == case["y"]["shape"]
        assert _err(y, case["y"]) < 1e-5, name
        for got, ref in zip(dins, case["dinputs"]):
            assert _err(got, ref) < 3e-5, name
        assert set(grads) == set(case["grads"])
        for k, ref in case["grads"].items():
            assert _err(grads[k], ref) < 5e-5, (name, k)


def vq_case_tensors(case):
    sd = W.module_state(case["seed"], case["shapes"])
    z = W.normal(case["seed"], "z", tuple(case["zshape"])) * 0.05
    dy = W.normal(case["seed"], "dy", tuple(case["zshape"]))
    return sd["embedding.weight"], z, dy


def test_vector_quantizer_oracle_matches_reference():
    g = load_golden("blocks_tokenizers.pt")
    for name in ("vq_plain", "vq_l2norm", "vq_wide", "vq_cluster"):
        case = g[name]
        code, z, dy = vq_case_tensors(case)
        code, z = code.clone().requires_grad_(True), z.requires_grad_(True)
        l2 = bool(case["kwargs"].get("use_l2_norm", False))
        zq, loss, commit, cbl, idx = O.vector_quantizer(z, code, 0.25, l2)
        assert torch.equal(idx, case["indices"]), name
        assert abs(float(loss) - case["quantizer_loss"]) < 1e-6 * max(1.0, abs(case["quantizer_loss"])), name
        assert abs(float(commit) - case["commitment_loss"]) < 1e-7 and abs(float(cbl) - case["codebook_loss"]) < 1e-7
        dz, dcode = torch.autograd.grad((zq * dy).sum() + loss, [z, code])
        assert _err(zq.detach(), case["zq"]) < 1e-6 and _err(dz, case["dz"]) < 1e-5 and _err(dcode, case["dcodebook"]) < 1e-5, name
        if case["kwargs"].get("clustering_vq"):
            new_code, prob = O.vq_cluster_update(z.detach(), code.detach(), torch.zeros(code.shape[0]), 0.99, l2)
            assert case["codebook_moved"] > 0.1            # the refresh really changed the codebook in the reference run
            assert _err(new_code, case["codebook_after"]) < 1e-6
            assert torch.allclose(prob, case["embed_prob_after"], atol=1e-7)
