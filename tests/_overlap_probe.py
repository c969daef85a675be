"""Helper of test_gpu_ddp.py (run as a fresh process): do the main stream and the weight-gradient side stream run kernels side by
side once RCCL is up?  argv[1] = "claim": vitamd.functions.claim_streams() before init_process_group (the documented order);
"late": the side stream is first used after RCCL's streams exist; "heal": as "late", then vitamd.functions.ensure_side_overlap()
(what DataParallel does for multi-rank jobs).  Prints `ratio <two streams / one stream>`."""
import os, socket, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, functions as F
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if sys.argv[1] == "claim":
    F.claim_streams(dev)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 16, device=dev)
dist.all_reduce(t)                                    # RCCL's own stream has run
R, P, Q = 32768, 512, 512                             # 4 output tiles, one workgroup each (splits=1): a long kernel on 4 of 256 CUs
l = torch.randn(R, P, device=dev).to(torch.bfloat16)
r = torch.randn(R, Q, device=dev).to(torch.bfloat16)
o1, o2 = torch.empty(P, Q, device=dev), torch.empty(P, Q, device=dev)
if sys.argv[1] == "heal":
    F.SIDE.stream(dev)
    print(f"ensure_side_overlap -> {F.ensure_side_overlap(dev):.3f}")
side = F.SIDE.stream(dev)


def run(both):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if both:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):
                ops.gemm_tn(l, r, o2, accumulate=False, splits=1)
    for _ in range(4):
        ops.gemm_tn(l, r, o1, accumulate=False, splits=1)
    if both:
        torch.cuda.current_stream().wait_stream(side)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


run(True)
one = min(run(False) for _ in range(3))
two = min(run(True) for _ in range(3))
dist.destroy_process_group()
print(f"one stream {one:.3f} ms, two streams {two:.3f} ms")
print(f"ratio {two / one:.3f}")
