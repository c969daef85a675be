"""Patch-embed backward reduction (g fp32 [B, 197, 768] -> dpos, dextra, bf16 rows, bias gradient) at the headline shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops
dev = torch.device("cuda")
B, seq, extra, D = 256, 197, 1, 768
g = torch.randn(B * seq, D, device=dev)
ops.embed_bwd(g, B, seq, extra, D); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.embed_bwd(g, B, seq, extra, D)
e.record(); torch.cuda.synchronize()
print(f"embed_bwd (+ zero fills, bias reduce): {s.elapsed_time(e) / 20 * 1e3:.1f} us per call; the pass moves {(g.numel() * 6) / 1e6:.0f} MB")
