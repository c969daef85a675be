"""ViT-B/16 at other resolutions (token counts 65 .. 1025: the three attention regimes) - functional and throughput check."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
for img, batch in ((128, 512), (224, 256), (256, 192), (320, 128), (384, 64), (512, 32)):
    torch.manual_seed(0)
    m = TV.ViTClassifier(TV.ViTConfig(img, 3, 16, "B", 1, 0.0)).to(dev)
    x = torch.randn(batch, 3, img, img, device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
    def step():
        m.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        l = torch.nn.functional.cross_entropy(m(x), y); l.backward(); return l
    for _ in range(3): l = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): l = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 8 * 1e3
    N = (img // 16) ** 2 + 1
    gf = 3 * 2 * (12 * (N * 768 * 2304 + 2 * N * N * 768 + 2 * N * 768 * 3072) + (N - 1) * 768 * 768) / 1e9
    finite = all(torch.isfinite(p.grad).all().item() for p in m.parameters())
    print(f"{img:4d} px ({N:5d} tokens) batch {batch:4d}: {ms:7.2f} ms/step {batch / ms * 1e3:7.0f} img/s ~{batch / ms * gf:5.0f} TFLOP/s  loss {float(l):.3f} grads finite {finite}", flush=True)
    del m, x, y
