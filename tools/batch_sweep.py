"""ViT-B/16 forward+backward throughput across batch sizes (tile-selection rules must not leave cliffs)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
m = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
for batch in (8, 16, 32, 64, 96, 128, 192, 256, 320, 384, 512):
    x = torch.randn(batch, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
    def step():
        m.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        torch.nn.functional.cross_entropy(m(x), y).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"batch {batch:4d}: {ms:7.2f} ms/step  {batch / ms * 1e3:7.0f} img/s  {batch / ms * 96.786:6.0f} TFLOP/s", flush=True)
