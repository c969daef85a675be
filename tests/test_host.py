"""CPU-side checks (no GPU): module surface / checkpoint-key contract, config derivations, the LR
schedule against the reference's trace, the C-ABI library's exported symbols, and that the
product path refuses to run without a device instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from vitamd import lib
    header = open(os.path.join(ROOT, "include", "vitamd.h")).read()
    declared = set(re.findall(r"\b(?:int|long)\s+(vitamd_\w+)\s*\(", header))
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(dll, name), name
    assert lib.load().vitamd_abi_version() == lib.ABI_VERSION
    # ... with as many parameters in the binding as in the header (a dropped or added argument would shift every later one silently)
    for name, params in re.findall(r"\b(?:int|long)\s+(vitamd_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.S):
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        assert n == len(lib.SIGNATURES[name]), (name, n, len(lib.SIGNATURES[name]))


def test_weight_gradient_form_policy():
    """functions.TN_FORM_POLICY: "auto" = the 12-wave exclusive kernel for the fc2 weight gradient (beside GEMMs that fill their CUs anyway) and for
    every weight gradient when there is no second stream; the 8-wave shared kernel where LayerNorm waves share the CUs."""
    from vitamd import functions as F, ops
    keep_p, keep_s = F.TN_FORM_POLICY, F.SIDE.enabled
    try:
        F.TN_FORM_POLICY, F.SIDE.enabled = "auto", True
        assert [F._tn_form(w) for w in ("fc2", "fc1", "qkv")] == [ops.TN_FORM_EXCLUSIVE, ops.TN_FORM_SHARED, ops.TN_FORM_SHARED]
        F.SIDE.enabled = False
        assert {F._tn_form(w) for w in ("fc2", "fc1", "qkv")} == {ops.TN_FORM_EXCLUSIVE}
        F.SIDE.enabled = True
        F.TN_FORM_POLICY = "shared"
        assert {F._tn_form(w) for w in ("fc2", "fc1", "qkv")} == {ops.TN_FORM_SHARED}
        F.TN_FORM_POLICY = "exclusive"
        assert {F._tn_form(w) for w in ("fc2", "fc1", "qkv")} == {ops.TN_FORM_EXCLUSIVE}
    finally:
        F.TN_FORM_POLICY, F.SIDE.enabled = keep_p, keep_s


def test_state_dict_contract_matches_reference_keys_and_shapes():
    import train_vit as TV
    for fixture, size, preset, classes in (("vit_s32.pt", 32, "S", 10), ("vit_b224.pt", 224, "B", 1000)):
        g = load_golden(fixture)
        m = TV.ViTClassifier(TV.ViTConfig(size, 3, 16, preset, 1, 0.0), num_classes=classes)
        sd = m.state_dict()
        assert sorted(sd.keys()) == g["state_keys"]
        assert {k: list(v.shape) for k, v in sd.items()} == g["state_shapes"]
        assert sum(p.numel() for p in m.parameters()) == g["n_params"]


def test_transformer_surface():
    import transformer as T
    cfg = T.transformer_configs["B"](block_size=197, dropout=0.0)
    assert (cfg.n_layers, cfg.n_heads, cfg.n_embd, cfg.head_dim, cfg.causal) == (12, 12, 768, 64, False)
    assert (T.S(block_size=1).n_embd, T.L(block_size=1).n_layers) == (512, 24)
    c = T.TransformerConfig(n_layers=1, n_heads=2, n_embd=128, block_size=9, causal=True)
    m = T.Transformer(c)
    assert m.n_embd == 128 and m.layers[0].multi_attn.n_heads == 2 and m.layers[0].causal   # config fields copied onto modules
    keys = set(m.state_dict().keys())
    assert keys == {"layers.0.multi_attn.qkv.weight", "layers.0.multi_attn.qkv.bias", "layers.0.multi_attn.mask",
                    "layers.0.mlp.0.weight", "layers.0.mlp.0.bias", "layers.0.mlp.2.weight", "layers.0.mlp.2.bias"}
    mask = m.state_dict()["layers.0.multi_attn.mask"]
    assert mask.shape == (9, 9) and mask[0, 1] == float("-inf") and mask[1, 0] == 0 and mask[3, 3] == 0
    old = T.TransformerConfig(n_layers=1, n_heads=2, n_embd=128, block_size=9)
    del old.__dict__["causal"]          # configs saved before `causal` existed (reference transformer.py:19)
    assert T.Attention(old).causal is False


def test_vit_config_derivations_and_mutation():
    import train_vit as TV
    c = TV.ViTConfig(224, 3, 16, "B", 1, 0.0)
    assert (c.n_patches, c.patch_dim, c.trans_config.block_size) == (196, 768, 197)
    # the TiTok decoder pattern (reference train_titok.py:31-32): 1x1 "patches" over latents, n_patches overridden
    d = TV.ViTConfig(32, 512, 1, "S", 256, 0.0)
    d.n_patches = 32
    v = TV.ViT(d)
    assert v.pos_emb.weight.shape == (32, 512) and v.extra_emb.weight.shape == (256, 512)
    assert v.patch_proj.weight.shape == (512, 512, 1, 1)
    e = TV.ViT(TV.ViTConfig(256, 3, 16, "B", 0, 0.0))   # ViT-VQGAN: zero extra tokens (reference train_vit_vqgan.py:29)
    assert e.extra_emb.weight.shape == (0, 768)


def test_no_cpu_fallback():
    import train_vit as TV
    from vitamd import lib
    m = TV.ViTClassifier(TV.ViTConfig(32, 3, 16, "S", 1, 0.0), num_classes=10)
    with pytest.raises(lib.VitamdError):
        m(torch.randn(2, 3, 32, 32))
    import transformer as T
    with pytest.raises(lib.VitamdError):
        T.Transformer(T.S(block_size=5, dropout=0.1))(torch.randn(1, 5, 512))   # dropout is supported, host tensors are not
    with pytest.raises(ValueError):
        T.Transformer(T.S(block_size=5, dropout=1.5))(torch.randn(1, 5, 512))


def test_lr_scheduler_reproduces_reference_trace():
    import utils as U
    g = load_golden("lr_schedule.pt")
    p = torch.nn.Parameter(torch.zeros(1))
    optim = torch.optim.AdamW([p], lr=g["base_lr"])
    sched = U.get_lr_scheduler(optim, g["warmup_steps"], g["train_steps"], g["min_lr"])
    for s, lr in enumerate(g["lrs"].tolist()):
        assert abs(optim.param_groups[0]["lr"] - lr) < 1e-12, (s, optim.param_groups[0]["lr"], lr)
        optim.step()
        sched.step()
    assert U.get_params_str(torch.nn.Linear(1000, 1000)) == "1.0M"


def test_shard_batch():
    from vitamd.ddp import shard_batch
    for n, w in ((2048, 8), (10, 4), (7, 8)):
        spans = [shard_batch(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_tokenizer_surfaces_match_reference_checkpoints():
    import train_titok as TT
    import train_vit_vqgan as TQ
    g = load_golden("titok_s256.pt")
    m = TT.TiTok(TT.TiTokConfig(256, 16, 32, 2048, 12, "S"))
    assert sorted(m.state_dict().keys()) == g["state_keys"] and sum(p.numel() for p in m.parameters()) == g["n_params"] == 36_034_828
    assert m.config.dec_vit_config.n_patches == 32 and m.dec.vit.extra_emb.weight.shape == (256, 512)
    g = load_golden("vitvqgan_b256.pt")
    v = TQ.ViTVQGAN(TQ.ViTVQGANConfig(256, 16, 2048, 12, "B"))
    assert sorted(v.state_dict().keys()) == g["state_keys"] and sum(p.numel() for p in v.parameters()) == g["n_params"] == 158_069_772
    assert {k: list(t.shape) for k, t in v.state_dict().items()} == g["state_shapes"]
    assert float(m.quant.codebook.weight.abs().max()) <= 1.0 / 2048 + 1e-9          # reference init (train_titok.py:49)


def test_blocks_surface_state_dict_contract():
    import blocks as BK
    g = load_golden("blocks_tiny.pt")
    ctors = {"rab": ("ResidualAttentionBlock", dict(d_model=128, n_head=2)), "rab_nomlp": ("ResidualAttentionBlock", dict(d_model=128, n_head=2, mlp_ratio=0)),
             "uvit_skip": ("UViTBlock", dict(dim=128, num_heads=2, skip=True)), "uvit_bias": ("UViTBlock", dict(dim=128, num_heads=2, qkv_bias=True)),
             "attn": ("Attention", dict(dim=128, num_heads=2, qkv_bias=True)), "mlp": ("Mlp", dict(in_features=128, hidden_features=512))}
    for name, (cls, kw) in ctors.items():
        m = getattr(BK, cls)(**kw)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == g[name]["shapes"], name
    assert isinstance(BK.ATTENTION_MODE, str)
    blk = BK.UViTBlock(128, 2, drop=0.1, drop_path=0.1)        # rates accepted (configs carry them) ...
    assert isinstance(blk.drop_path, BK.DropPath) and blk.mlp.drop.p == 0.1
    from vitamd.lib import VitamdError
    with pytest.raises(VitamdError):                            # ... and training with them runs on the kernels: a host tensor is refused
        blk(torch.zeros(1, 4, 128))
    with pytest.raises(NotImplementedError):
        BK.Attention(96, 2)            # head_dim 48


def _tok_cfg(c):
    from types import SimpleNamespace as NS
    return NS(image_size=c["image_size"], patch_size=c["patch_size"], transformer=c["transformer"], latent_tokens=c["latent_tokens"],
              latent_dim=c["latent_dim"], model=NS(vq_model={"text_context_length": c["text_context_length"], "text_embed_dim": c["text_embed_dim"]}))


def test_blocks_tokenizer_wrappers_state_dict_contract():
    """blocks.py:209,286,365,406 - same class names, ctor signatures and checkpoint keys/shapes as the reference
    (shapes recorded from the reference's own state_dict in tests/golden/blocks_tokenizers.pt)."""
    import blocks as BK
    g = load_golden("blocks_tokenizers.pt")
    cfg = _tok_cfg(g["config"])
    for name, cls in (("encoder", "TiTokEncoder"), ("decoder", "TiTokDecoder"), ("tatitok_decoder", "TATiTokDecoder")):
        m = getattr(BK, cls)(cfg)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == g[name]["shapes"], name
    for name in ("vq_plain", "vq_l2norm", "vq_wide", "vq_cluster"):
        vq = BK.VectorQuantizer(**g[name]["kwargs"])
        shapes = {k: list(v.shape) for k, v in vq.state_dict().items() if k != "embed_prob"}
        assert shapes == g[name]["shapes"], name
        assert float(vq.embedding.weight.abs().max()) <= 1.0 / vq.codebook_size + 1e-9       # reference init (blocks.py:421)
    assert "embed_prob" in BK.VectorQuantizer(clustering_vq=True).state_dict()
    # no CPU path: the wrappers raise instead of silently computing on the host
    from vitamd.lib import VitamdError
    enc = BK.TiTokEncoder(cfg)
    with pytest.raises((VitamdError, RuntimeError)):
        enc(torch.zeros(1, 3, 32, 32), torch.zeros(8, 512))


def test_gemm_nt_plan_reports_the_dispatch_rule_without_a_gpu():
    """vitamd_gemm_nt_plan (ABI 8): the kernel form a launch would take - form | rows << 8 - straight from the dispatcher's own rule (plan_single in
    csrc/gemm_nt.hip executes for real launches too).  No GPU is needed: with no current device the rule assumes 256 CUs."""
    from vitamd import lib
    plan = lib.load().vitamd_gemm_nt_plan
    SMALL, PP, PERS, SEAM, LOADER = 1, 2, 3, 4, 5
    M = 256 * 197
    f = lambda *a: (plan(*a) & 0x7f, plan(*a) >> 8)
    assert f(M, 2304, 768, 2304, 0, 0) == (LOADER, 256)               # QKV forward: short K loop, 6.9 tiles per CU: the loader-wave form (round 4)
    assert f(M, 3072, 768, 3072, 6, 0) == (LOADER, 256)               # fc1 + GELU (stored derivative), table in LDS
    assert f(M, 3072, 768, 3072, 7, 0) == (LOADER, 256)               # dgrad-fc2 x gelu'
    assert f(M, 2304, 704, 2304, 0, 0) == (SEAM, 256)                 # an odd number of K-tiles: the seam kernel
    assert f(M, 768, 3072, 768, 0, 0) == (PERS, 320)                  # N = 768: two rounds of 320 rows instead of three of 256
    assert f(M, 768, 3072, 768, 2, 0) == (PERS, 320)                  # fc2 + fp32 residual
    assert f(25216, 2304, 768, 2304, 0, 0) == (LOADER, 256)           # batch 128: 891 tiles >= 3 per CU (the host pre-filter missed it)
    assert f(25216, 2304, 768, 2304, 0, 1024) == (PERS, 320)          # code 1024: persistent, no seam form (3 rounds of 320 rows <= 4 of 256)
    assert f(M, 2304, 768, 2304, 0, 512) == (PP, 256)                 # code 512: one workgroup per tile
    assert f(M, 768, 3072, 768, 0, 2048) == (LOADER, 256)             # the loader-wave form on request
    assert plan(M, 768, 192, 768, 0, 2048) == -1                      # ... refused (VITAMD_ERR_SHAPE) for an odd number of K-tiles
    assert plan(M, 768, 3072, 768, 2, 2048) == -1                     # ... and for the fp32-residual epilogue
    assert f(300, 200, 64, 200, 5, 0) == (SMALL, 128)
    assert plan(0, 8, 64, 8, 0, 0) == -1 and plan(M, 768, 3072, 768, 0, 77) == -2 and plan(M, 768, 3072, 768, 9, 0) == -2


def test_init_and_compute_fail_loudly_without_a_device():
    """No GPU in this process: vitamd_init reports a HIP failure instead of crashing, and nothing falls back to the CPU."""
    if torch.cuda.is_available():
        pytest.skip("needs a process without a GPU")
    from vitamd import lib
    assert lib.load().vitamd_init(-1, None) == 3                      # VITAMD_ERR_LAUNCH
    assert lib.ERRORS[4].startswith("vitamd_init")


def test_two_hip_runtimes_are_detected(tmp_path):
    """lib.load() scans /proc/self/maps after dlopen: a second libamdhip64 (round 3's `HIP launch failure` when libvitamd.so was loaded before
    torch) raises instead of failing at the first launch.  Driven here with stub maps files."""
    from vitamd import lib
    one = tmp_path / "maps_one"
    one.write_text("7f00-7f10 r-xp 00000000 fd:01 1 /usr/lib/torch/lib/libamdhip64.so\n"
                   "7f10-7f20 r--p 00010000 fd:01 1 /usr/lib/torch/lib/libamdhip64.so\n"
                   "7f20-7f30 r-xp 00000000 fd:01 2 /usr/lib/libc.so.6\n7f30-7f40 rw-p 00000000 00:00 0 \n")
    assert lib.check_single_hip_runtime(str(one)) == ["/usr/lib/torch/lib/libamdhip64.so"]
    two = tmp_path / "maps_two"
    two.write_text(one.read_text() + "7f40-7f50 r-xp 00000000 fd:01 3 /opt/rocm-7.2.0/lib/libamdhip64.so.7.2.0\n")
    with pytest.raises(lib.VitamdError) as e:
        lib.check_single_hip_runtime(str(two))
    assert "/opt/rocm-7.2.0/lib/libamdhip64.so.7.2.0" in str(e.value) and "/usr/lib/torch/lib/libamdhip64.so" in str(e.value)
    assert len(lib.check_single_hip_runtime()) <= 1                  # this very process: one runtime (torch's), or none mapped yet


def test_pmc_traffic_keeps_template_instantiations_apart_and_bench_weights_them(tmp_path, monkeypatch):
    """VERDICT r3 item 2.  tools/pmc_traffic.py keys kernels by the full name up to the parameter list (round 3 cut at 60 characters and folded
    gemm_nt_seam_kernel<0|1|3, ...> into one entry); bench.pmc_profile sets every instantiation of the step against ITS OWN algorithmic bytes
    and reports the launch-weighted family figure.  Driven with a synthetic pair of rocprofv3 counter_collection.csv passes."""
    import csv
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    import pmc_traffic
    import bench
    ns = "(anonymous namespace)::"
    table = bench.nt_step_table()
    assert set(table) == {"gemm_nt_ld_kernel<0, false, 1>", "gemm_nt_ld_kernel<1, true, 1>", "gemm_nt_ld_kernel<3, false, 1>",
                          "gemm_nt_pp_kernel<0, 10, 4, 6, true>", "gemm_nt_pp_kernel<2, 10, 4, 6, true>"}
    assert sum(v["launches_per_step"] for v in table.values()) == 72
    M = 256 * 197
    fc2, dqkv = bench.nt_algorithmic_bytes(M, 768, 3072, 0), bench.nt_algorithmic_bytes(M, 768, 2304, 0)
    assert (fc2, dqkv) == ((M * 3072 + 768 * 3072 + M * 768) * 2, (M * 2304 + 768 * 2304 + M * 768) * 2)
    assert table["gemm_nt_pp_kernel<0, 10, 4, 6, true>"]["algorithmic_bytes_per_launch"] == (23 * fc2 + 12 * dqkv) // 35
    ratio = {"gemm_nt_ld_kernel<0, false, 1>": 1.8, "gemm_nt_ld_kernel<1, true, 1>": 1.5, "gemm_nt_ld_kernel<3, false, 1>": 1.6,
             "gemm_nt_pp_kernel<0, 10, 4, 6, true>": 1.25, "gemm_nt_pp_kernel<2, 10, 4, 6, true>": 1.2}
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        d = tmp_path / "csv" / sub
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            i = 0
            for inst, r in ratio.items():
                full = f"void {ns}{ns if '_ld_' in inst else ''}{inst}({ns if '_ld_' in inst else ''}GemmNtArgs)"
                assert len(full) > 60
                total_kb = r * table[inst]["algorithmic_bytes_per_launch"] / 1024
                kb = total_kb / 4 if counter == "FETCH_SIZE" else total_kb / 2            # 2 x fetch + write = total
                for _ in range(3):
                    i += 1
                    w.writerow([i, full, counter, kb])
            w.writerow([i + 1, f"{ns}splitk_reduce_kernel(float const*, float*, int, int, int, int, int, int, int)", counter, 100.0])
    folded = pmc_traffic.fold(str(tmp_path / "csv"))
    assert set(ratio) <= set(folded) and "splitk_reduce_kernel" in folded and folded["gemm_nt_ld_kernel<1, true, 1>"]["launches"] == 3
    out = tmp_path / "prof"
    out.mkdir()
    json.dump(folded, open(out / "final_pmc_hbm_traffic.json", "w"))
    json.dump({"csrc_sha16": bench.csrc_sha16()}, open(out / "final_pmc_meta.json", "w"))
    monkeypatch.setattr(bench, "PMC_DIR", str(out))
    prof = bench.pmc_profile(table)
    assert prof["stale"] is False
    nt = prof["traffic"]["gemm_nt"]
    assert nt["launches_covered"] == 72 and set(nt["per_instantiation"]) == set(ratio)
    for inst, r in ratio.items():
        assert abs(nt["per_instantiation"][inst]["ratio"] - r) < 2e-3
    want = sum(r * table[i]["algorithmic_bytes_per_launch"] * table[i]["launches_per_step"] for i, r in ratio.items()) / 72
    assert abs(nt["bytes_per_launch"] - want) / want < 1e-3 and nt["bytes_per_launch"] > nt["algorithmic_bytes_per_launch"]
