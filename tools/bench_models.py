"""Forward+backward throughput of the three presets of transformer.py:56-58 as ViT classifiers (224 px, patch 16) - a scale
check of the kernels at D = 512 / 768 / 1024 beyond the headline configuration."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
FL = {"S": None, "B": 96.786, "L": None}
import sys as _s
ORDER = _s.argv[1] if len(_s.argv) > 1 else "SBL"
BATCH = {"S": 512, "B": 256, "L": 128}
for preset, batch in ((p, BATCH[p]) for p in ORDER):
    torch.manual_seed(0)
    m = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, preset, 1, 0.0)).to(dev)
    nparam = sum(p.numel() for p in m.parameters())
    x = torch.randn(batch, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
    def step():
        m.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        loss = torch.nn.functional.cross_entropy(m(x), y); loss.backward(); return loss
    for _ in range(3): l = step()
    torch.cuda.synchronize(); st0 = torch.cuda.memory_stats(); t0 = time.perf_counter()
    for _ in range(10): l = step()
    host = (time.perf_counter() - t0) / 10 * 1e3
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    st1 = torch.cuda.memory_stats()
    print(f"   host enqueue {host:.2f} ms/step, device mallocs in the timed steps {st1['num_device_alloc'] - st0['num_device_alloc']}, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")
    tc = m.vit.config.trans_config
    D, L, N = tc.n_embd, tc.n_layers, 197
    gflop_img = 3 * 2 * (L * (N * D * 3 * D + 2 * N * N * D + 2 * N * D * 4 * D) + 196 * 768 * D) / 1e9   # fwd + 2x bwd, MACs*2
    print(f"ViT-{preset}/16 224px batch {batch}: {nparam / 1e6:6.1f} M params  {ms:7.2f} ms/step  {batch / ms * 1e3:8.0f} img/s  "
          f"~{batch / ms * 1e3 * gflop_img / 1e3:6.0f} TFLOP/s  loss {float(l):.3f}", flush=True)
    del m, x, y
    # (no torch.cuda.empty_cache() here: handing the blocks back and re-allocating them between models was seen to leave the next model up to 5x slower - memory placement - while the cached blocks are simply reused)
