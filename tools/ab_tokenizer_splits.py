"""TiTok-S / ViT-VQGAN-B forward+backward against the weight-gradient cut (functions.TN_TARGET_WGS: workgroups per dW GEMM = tiles x split-K factor):
small dW matrices (TiTok-S: 512-wide) at 252 workgroups mean 16-63 splits, i.e. as many fp32 partial tiles for the reduce pass as the GEMM reads operands."""
import os, sys, time, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_titok as TT, train_vit_vqgan as TQ
from vitamd import functions as F
dev = torch.device("cuda")
for name, make, bs in (("TiTok-S", lambda: TT.TiTok(TT.TiTokConfig(256, 16, 32, 2048, 12, "S")), 256), ("ViT-VQGAN-B", lambda: TQ.ViTVQGAN(TQ.ViTVQGANConfig(256, 16, 2048, 12, "B")), 128)):
    torch.manual_seed(0); model = make().to(dev); x = torch.rand(bs, 3, 256, 256, device=dev)
    def step():
        model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        recon, idx, ql = model(x); (torch.nn.functional.mse_loss(recon, x) + ql).backward()
    def timed(n=5):
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    for _ in range(2): step()
    cfgs = [None, 192, 160, 128, 96]
    res = {c: [] for c in cfgs}
    for r in range(3):
        for c in cfgs:
            F.TN_TARGET_WGS = c; res[c].append(timed())
    F.TN_TARGET_WGS = None
    print(name, "  ".join(f"{'252(default)' if c is None else c}: {statistics.median(v):.2f} ms" for c, v in res.items()), flush=True)
    del model, x
