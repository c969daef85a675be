"""Whole-step A/B of the loader-wave NT GEMM per launch class (experimental library: vitamd_set_debug2 bits 0-3, gemm_nt.hip::ld_auto), interleaved, medians.
usage: ab_ld.py   (AB_NO_SIDE=1: weight-gradient GEMMs on the main stream)"""
import os, sys, time, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug2.argtypes = [ctypes.c_int]
if os.environ.get("AB_NO_SIDE") == "1": F.SIDE.enabled = False
# the defaults since round 4: loader form ON for the three short-K classes (bits 0-2 flip them OFF), TN loader requests split (bit 6 = round-3 loaders); bit 7 (timing only): no split-K reduce pass
cfgs = {"production": 0, "seam_qkv": 1, "seam_gelu": 2, "seam_dgelu": 4, "seam_all(r3)": 7, "ld_n768_too": 8, "ld_dgrads768_too": 32, "tn_r3_loaders": 64, "no_splitk_reduce(!)": 128}
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    loss = torch.nn.functional.cross_entropy(model(x), y); loss.backward(); return loss
def timed(n=5):
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
res = {k: [] for k in cfgs}; losses = {}
for r in range(5):
    for k, bits in cfgs.items():
        L.vitamd_set_debug2(bits); res[k].append(timed())
        if r == 0: losses[k] = float(step())
L.vitamd_set_debug2(0)
for k in cfgs: print("%-24s median %.2f ms/step  %s  loss %.6f" % (k, statistics.median(res[k]), ["%.2f" % v for v in res[k]], losses[k]), flush=True)
