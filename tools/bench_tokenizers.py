"""Forward+backward throughput of the TiTok-S and ViT-VQGAN-B tokenizers (BASELINE configs[3], [4]
models; MSE + quantiser loss, no perceptual term) on one MI355X."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_titok as TT, train_vit_vqgan as TQ
from vitamd.functions import WEIGHTS
dev = torch.device("cuda")
for name, model, bs in (("TiTok-S 256px/32 latents", TT.TiTok(TT.TiTokConfig(256, 16, 32, 2048, 12, "S")), 256),
                        ("ViT-VQGAN-B 256px", TQ.ViTVQGAN(TQ.ViTVQGANConfig(256, 16, 2048, 12, "B")), 128)):
    model = model.to(dev)
    x = torch.rand(bs, 3, 256, 256, device=dev)
    def step():
        model.zero_grad(set_to_none=True); WEIGHTS.clear()
        recon, idx, ql = model(x)
        (torch.nn.functional.mse_loss(recon, x) + ql).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 8
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:28s} batch {bs:4d}  {dt*1e3:8.2f} ms/step  {bs/dt:9.1f} img/s", flush=True)
