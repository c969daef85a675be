"""Headline benchmark: images/sec of ViT-B/16 224x224 bf16 forward + loss + backward
(BASELINE.json metric / configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of 256 synthetic images per GPU: per-step bf16
weight cast, patch embed, 12 transformer layers, head, cross-entropy, full backward to every
parameter gradient and (N > 1) the bucketed RCCL gradient all-reduce, overlapped with backward.
Inputs are resident in HBM before timing.  The optimiser step is outside the metric.
Weak scaling: 256 images per GPU at every N.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))

GFLOP_PER_IMAGE = 96.786        # SURVEY.md section 8d / BASELINE.md section 3 (fwd 32.339 + bwd 64.447)
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PER_GPU_BATCH = 256


def _timed(fns, reps):
    for f in fns:
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()                      # our kernels launch on torch's current stream: events see them
    for _ in range(reps):
        for f in fns:
            f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def kernel_roofline(dev):
    """Live HIP-event timing of the two dominant kernels at the shapes one training step launches them with:
      gemm_nt_pp_kernel - every forward Linear and every input-gradient GEMM (6 launches per layer, ~2/3 of the step's FLOPs),
      gemm_tn_pp_kernel / gemm_tn_ld_kernel - every weight-gradient GEMM (3 launches per layer, split-K reduce pass included; the form each
      launch takes in the step: functions._tn_form).
    achieved = algorithmic FLOPs (2*M*N*K per launch) / average launch duration over the launches of one layer, timed in the order
    the layer issues them.  `roofline` itself describes the NT family (the larger share); `kernels` carries both."""
    from vitamd import ops, functions as F
    M, D = PER_GPU_BATCH * 197, 768
    g = torch.Generator(device="cpu").manual_seed(1)

    def rb(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)

    x1, x3, x4 = rb(M, D), rb(M, 3 * D), rb(M, 4 * D)
    wqkv, w1, w2 = rb(3 * D, D, scale=0.03), rb(4 * D, D, scale=0.03), rb(D, 4 * D, scale=0.03)
    wqkv_t, w1_t, w2_t = rb(D, 3 * D, scale=0.03), rb(D, 4 * D, scale=0.03), rb(4 * D, D, scale=0.03)
    b3, b4, b1 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev), torch.randn(D, device=dev)
    res = torch.randn(M, D, device=dev)
    cs = torch.zeros(4 * D, device=dev)
    y2 = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
    nt_calls = [       # (name, launch, FLOPs, launches per step): inside the stack fc2 writes bf16 (DEFER_RESID), only the last layer's adds the residual
        ("qkv", lambda: ops.gemm_nt(x1, wqkv, ops.EPI_BIAS_BF16, bias=b3), 2.0 * M * D * 3 * D, 12),
        ("fc1+gelu", lambda: ops.gemm_nt(x1, w1, ops.EPI_GELU_DG, bias=b4), 2.0 * M * D * 4 * D, 12),
        ("fc2", lambda: ops.gemm_nt(x4, w2, ops.EPI_BIAS_BF16, bias=b1, out=y2), 2.0 * M * D * 4 * D, 11),
        ("fc2+resid", lambda: ops.gemm_nt(x4, w2, ops.EPI_RESID_F32, bias=b1, aux=res), 2.0 * M * D * 4 * D, 1),
        ("dgrad_fc2", lambda: ops.gemm_nt(x1, w2_t, ops.EPI_DMUL, aux=x4, colsum=cs), 2.0 * M * D * 4 * D, 12),
        ("dgrad_fc1", lambda: ops.gemm_nt(x4, w1_t, ops.EPI_BIAS_BF16), 2.0 * M * D * 4 * D, 12),
        ("dgrad_qkv", lambda: ops.gemm_nt(x3, wqkv_t, ops.EPI_BIAS_BF16), 2.0 * M * D * 3 * D, 12),
    ]
    dWqkv, dW1, dW2 = (torch.empty(s, device=dev) for s in ((3 * D, D), (4 * D, D), (D, 4 * D)))
    tn_calls = [       # split-K factors as the step chooses them (vitamd.functions._tn_splits)
        ("dW_fc2", lambda: ops.gemm_tn(x1, x4, dW2, accumulate=False, splits=F._tn_splits(dW2), form=F._tn_form("fc2")), 2.0 * M * D * 4 * D, 12),
        ("dW_fc1", lambda: ops.gemm_tn(x4, x1, dW1, accumulate=False, splits=F._tn_splits(dW1), form=F._tn_form("fc1")), 2.0 * M * D * 4 * D, 12),
        ("dW_qkv", lambda: ops.gemm_tn(x3, x1, dWqkv, accumulate=False, splits=F._tn_splits(dWqkv), form=F._tn_form("qkv")), 2.0 * M * D * 3 * D, 12),
    ]
    step_table = nt_step_table(M)
    prof = pmc_profile(step_table)

    def family(calls, kernel):
        # per shape (informational): five launches back to back, median of three; reported: ONE STEP's launches of the family in layer order
        # (12 layers: every call whose per-step count exceeds the layer index) between one pair of events, twice
        detail = {name: round(flops / sorted(_timed([fn], 5) for _ in range(3))[1] / 1e9, 1) for name, fn, flops, _ in calls}
        seq = [fn for layer in range(12) for name, fn, _, n in calls if (layer < n if n != 1 else layer == 11)]
        ms = _timed(seq, 2)
        flops = sum(f * n for _, _, f, n in calls)
        ach = flops / ms / 1e9
        return {"kernel": kernel, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                "launches_per_step": len(seq), "avg_launch_us": round(ms * 1e3 / len(seq), 1), "per_shape_tflops": detail,
                "recorded_mfma_util": (prof.get("mfma_util") or {}).get(kernel), "recorded_traffic": (prof.get("traffic") or {}).get(kernel)}

    nt = family(nt_calls, "gemm_nt")          # gemm_nt_pp_kernel, gemm_nt_ld_kernel (and gemm_nt_seam_kernel for odd K-tile counts)
    tn = family(tn_calls, "gemm_tn")          # gemm_tn_pp_kernel (8 waves) and gemm_tn_ld_kernel (12 waves, loader waves)
    # (informational) the same weight-gradient GEMMs cut for the whole chip (252 workgroups) instead of the ~128 the step uses so that
    # they leave half the CUs to the main stream's kernels
    full = [(n, (lambda l=l, r=r, o=o: ops.gemm_tn(l, r, o, accumulate=False, splits=0)), f) for (n, _, f, _), (l, r, o) in
            zip(tn_calls, ((x1, x4, dW2), (x4, x1, dW1), (x3, x1, dWqkv)))]
    tn["whole_chip_split_per_shape_tflops"] = {name: round(flops / _timed([fn], 10) / 1e9, 1) for name, fn, flops in full}
    # `achieved` / `per_shape_tflops` are measured live in this run.  `recorded_*` and `traffic` are NOT: they come from rocprofv3 PMC passes of
    # this same command committed under profiles/ (collected with --pmc in separate runs, as the guide prescribes), and say so: "static": true;
    # "stale": true when the kernel sources have changed since those passes were made (then `traffic` is null).
    rec_ok = prof.get("source") is not None and not prof.get("stale")
    out = {"bound": "mfma", "kernel": "gemm_nt_pp_kernel (320x256x64 ping-pong tiles: the N = 768 GEMMs) + gemm_nt_ld_kernel (256x256x64, 8 compute + 4 loader waves: QKV, fc1+GELU, dgrad-fc2); the 72 large NT GEMM launches of one step, in layer order",
           "instantiations": step_table,
           "achieved": nt["achieved"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": nt["frac"],
           "traffic": (nt["recorded_traffic"] or {}).get("bytes_per_launch") if rec_ok else None,
           "recorded": {"static": True, "stale": bool(prof.get("stale")), "source": prof.get("source"), "made_at": prof.get("meta"),
                        "traffic_detail": nt["recorded_traffic"], "mfma_util": nt["recorded_mfma_util"]},
           "per_shape_tflops": nt["per_shape_tflops"], "kernels": {"gemm_nt": nt, "gemm_tn": tn},
           # what these main loops are observed to get: ~17 B per clock per CU of operand fill (one 1-KiB LDS-DMA piece per ~60
           # cycles; the bare path does 52-58 from L2), where a 256x256 (320x256) tile needs 32 (28.8) B per clock at full MFMA rate
           "feed_ceiling": {"bytes_per_clk_per_cu": {"requests_from_compute_waves": 17, "four_loader_waves": 21.5}, "frac_of_mfma_peak": {"256x256": 0.53, "320x256": 0.59, "256x256_loader_waves": 0.67},
                            "source": "DESIGN.md 4.6 / 4.7, profiles/r03/tn_loader_ring_variants.log, profiles/r04/nt_loader_vs_auto_ablations.log"}}
    return out


PMC_DIR = os.path.join("profiles", "r04")

# The 72 large NT GEMM launches of one training step at the headline config (M = 256 x 197 token rows), as vitamd.functions issues them:
# (name, ABI epilogue code, N, K, launches per step).  Inside the stack fc2 writes bf16 (the next LayerNorm adds the residual: DEFER_RESID);
# only the last layer's fc2 carries the fused fp32 residual.
NT_STEP_LAUNCHES = [("qkv", 0, 2304, 768, 12), ("fc1+gelu", 6, 3072, 768, 12), ("fc2", 0, 768, 3072, 11), ("fc2+resid", 2, 768, 3072, 1),
                    ("dgrad_fc2", 7, 3072, 768, 12), ("dgrad_fc1", 0, 768, 3072, 12), ("dgrad_qkv", 0, 768, 2304, 12)]


def nt_algorithmic_bytes(M, N, K, epi):
    """Bytes a launch must move once: both operands, the output(s), the epilogue's auxiliary input (SURVEY.md section 8d shapes)."""
    b = (M * K + N * K) * 2
    if epi == 2:
        return b + 2 * M * N * 4            # fp32 residual in, fp32 stream out
    return b + M * N * 2 * (2 if epi in (1, 6, 3, 7) else 1)      # bf16 out (+ the second GELU output / the dGELU factor)


def nt_instantiation(plan, epi):
    """The kernel instantiation (name as tools/pmc_traffic.py::kernel_key prints it) behind a vitamd_gemm_nt_plan code."""
    e = {6: 1, 7: 3}.get(epi, epi)
    form, rows = plan & 0x7f, plan >> 8
    mt = rows // 32
    tab = "true" if e == 1 else "false"
    if form == 5:
        return f"gemm_nt_ld_kernel<{e}, {tab}, 1>"
    if form == 4:
        return f"gemm_nt_seam_kernel<{e}, {mt}, 0, {tab}>"
    if form in (2, 3):
        return f"gemm_nt_pp_kernel<{e}, {mt}, 4, 6, {'true' if form == 3 else 'false'}>"
    return "gemm_nt_kernel<128, 128, 2, 2, %d>" % e


def nt_step_table(M=PER_GPU_BATCH * 197):
    """instantiation -> {launches per step, algorithmic bytes per launch (launch-weighted over the shapes it serves), shapes}: which kernel each
    launch class takes comes from the library's own dispatch rule (vitamd_gemm_nt_plan), so the table follows the code."""
    from vitamd import lib
    plan = lib.load().vitamd_gemm_nt_plan
    out = {}
    for name, epi, N, K, n in NT_STEP_LAUNCHES:
        inst = nt_instantiation(plan(M, N, K, N, epi, 0), epi)
        rec = out.setdefault(inst, {"launches": 0, "bytes": 0.0, "shapes": []})
        rec["launches"] += n
        rec["bytes"] += n * nt_algorithmic_bytes(M, N, K, epi)
        rec["shapes"].append(name)
    return {k: {"launches_per_step": v["launches"], "algorithmic_bytes_per_launch": int(v["bytes"] / v["launches"]), "shapes": v["shapes"]} for k, v in out.items()}


def csrc_sha16():
    """hash of the kernel sources: a recorded PMC profile belongs to exactly one state of them"""
    import glob, hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "vit-is-all-you-need_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(step_table=None):
    """Per-kernel MFMA utilisation and HBM bytes per launch RECORDED by rocprofv3 PMC passes of this bench command and committed under
    profiles/ (final_pmc_mfma.json: tools/pmc_mfma.py; final_pmc_hbm_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes folded by
    tools/pmc_traffic.py with the gfx950 correction 2 x FETCH_SIZE + WRITE_SIZE; final_pmc_meta.json: the source hash they were made at).
    Missing files -> empty; a source hash that differs from the tree's -> "stale"."""
    out = {}
    try:
        meta = json.load(open(os.path.join(ROOT, PMC_DIR, "final_pmc_meta.json")))
        out["meta"] = meta
        out["stale"] = meta.get("csrc_sha16") != csrc_sha16()
    except Exception:
        out["stale"] = True
    try:
        mf = json.load(open(os.path.join(ROOT, PMC_DIR, "final_pmc_mfma.json")))
        util = {}
        for fam in ("gemm_nt", "gemm_tn"):
            sel = {k: v for k, v in mf.items() if (("gemm_nt_pp_kernel" in k or "gemm_nt_seam_kernel" in k or "gemm_nt_ld_kernel" in k) if fam == "gemm_nt" else ("gemm_tn_pp_kernel" in k or "gemm_tn_ld_kernel" in k)) and v["avg_us_profiled"] > 50}
            rows = [(v["launches"], v["avg_us_profiled"], v["mfma_util"]) for v in sel.values()]
            if rows:       # time-weighted over the family's launches
                util[fam] = {"mfma_busy_frac": round(sum(n * t * u for n, t, u in rows) / sum(n * t for n, t, _ in rows), 4),
                             "per_instantiation": {k.split("::")[-1].split("(")[0]: v["mfma_util"] for k, v in sel.items()}}
        out["mfma_util"] = util
        out["source"] = os.path.join(PMC_DIR, "final_pmc_mfma.json")
    except Exception:
        pass
    try:
        tr = json.load(open(os.path.join(ROOT, PMC_DIR, "final_pmc_hbm_traffic.json")))
        key = "hbm_MB_avg_corrected(2*fetch+write)"
        traffic = {}
        # NT family: every instantiation the step launches, each against ITS OWN algorithmic bytes; the family figure weights them by launches per step
        per, tot_n, tot_b, tot_a = {}, 0, 0.0, 0.0
        for inst, rec in (step_table if step_table is not None else nt_step_table()).items():
            if inst not in tr:
                continue
            meas, algo, n = tr[inst][key] * 1e6, rec["algorithmic_bytes_per_launch"], rec["launches_per_step"]
            per[inst] = {"bytes": int(meas), "algorithmic": algo, "ratio": round(meas / algo, 3), "launches_per_step": n, "shapes": rec["shapes"]}
            tot_n += n; tot_b += n * meas; tot_a += n * algo
        if tot_n:
            traffic["gemm_nt"] = {"bytes_per_launch": int(tot_b / tot_n), "algorithmic_bytes_per_launch": int(tot_a / tot_n),
                                  "ratio": round(tot_b / tot_a, 3), "launches_covered": tot_n, "per_instantiation": per}
        tn = [v for k, v in tr.items() if k.startswith("gemm_tn_pp_kernel<") or k.startswith("gemm_tn_ld_kernel<")]
        if tn:
            tot_n = sum(v["launches"] for v in tn)
            traffic["gemm_tn"] = {"bytes_per_launch": int(sum(v["launches"] * v[key] for v in tn) / tot_n * 1e6),
                                  "algorithmic_bytes_per_launch": int((387 + 387 + 309) / 3 * 1e6 + 9.4e6)}
        out["traffic"] = traffic
    except Exception:
        pass
    return out


def box_identity(dev):
    """What identifies the device a line was measured on, so that an outlier box (round 3 met one whose in-GEMM buffer stores ran 4x slower,
    profiles/r03/store_trickle_README.md) is visible in BENCH_r*.json: architecture, CU count, the clock attributes HIP reports, memory size."""
    p = torch.cuda.get_device_properties(dev)
    out = {"name": p.name, "gcnArchName": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
           "total_memory_GiB": round(p.total_memory / 2**30, 1), "l2_cache_MiB": round(getattr(p, "L2_cache_size", 0) / 2**20, 1)}
    for k in ("clock_rate", "memory_clock_rate", "memory_bus_width"):       # kHz / kHz / bits where torch exposes them
        if hasattr(p, k):
            out[k] = getattr(p, k)
    out["hip"] = getattr(torch.version, "hip", None)
    return out


def also_models(dev):
    """Driver-visible numbers for BASELINE configs[3] / configs[4]: forward + backward (MSE + quantiser loss; the perceptual term
    needs downloaded ConvNeXt weights and stays out) of TiTok-S (train_titok.py:79-93) and ViT-VQGAN-B (train_vit_vqgan.py:34-91) on this GPU,
    3 warm-up + 5 timed steps each, after the headline timing; not part of `value`."""
    import train_titok as TT, train_vit_vqgan as TQ
    from vitamd.functions import WEIGHTS
    out = {}
    for key, make, bs in (("titok_s256", lambda: TT.TiTok(TT.TiTokConfig(256, 16, 32, 2048, 12, "S")), 256),
                          ("vitvqgan_b256", lambda: TQ.ViTVQGAN(TQ.ViTVQGANConfig(256, 16, 2048, 12, "B")), 128)):
        torch.manual_seed(0)
        model = make().to(dev)
        x = torch.rand(bs, 3, 256, 256, device=dev)

        def step():
            model.zero_grad(set_to_none=True)
            WEIGHTS.clear()
            recon, _idx, ql = model(x)
            (torch.nn.functional.mse_loss(recon, x) + ql).backward()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out[key] = {"img_s": round(bs / dt, 1), "batch": bs, "ms_per_step": round(dt * 1e3, 2), "what": "forward + backward, MSE + quantiser loss, 256x256, bf16 kernels"}
        del model, x
    return out


def cpu_baseline():
    """The CPU oracle (fp32 restatement of the reference, oracle/vit_oracle.py) timed on this
    host's cores on a bounded sample: 5 forward+backward iterations of batch 16 after one warm-up (~10 s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vit_oracle as O
    import weights as W
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)          # the GPU box grants a 16-core CPU share per GPU; more threads only thrash
    torch.set_num_threads(cores)
    cfg = O.OracleViTConfig.preset(224, 3, 16, "B", 1)
    sd = W.classifier_state(0, 3, 16, cfg.n_patches, 1, cfg.n_layers, cfg.n_embd, 1000)
    bs, iters = 16, 5
    images = W.normal(0, "images", (bs, 3, 224, 224))
    labels = W.randint(0, "labels", (bs,), 1000)
    O.classifier_loss_and_grads(images[:4], labels[:4], sd, cfg)
    t0 = time.time()
    for _ in range(iters):
        O.classifier_loss_and_grads(images, labels, sd, cfg)
    dt = time.time() - t0
    return {"value": round(bs * iters / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{iters} x (batch {bs} ViT-B/16 224 fp32 forward+loss+backward), CPU oracle, {dt:.1f} s"}


def gather_dist_info(device_index, own_ms, finish_wait_ms, diagnostics):
    """What an N > 1 line carries so that a bad scaling number can be diagnosed from the JSON alone: every rank's own step time, the
    time each rank's main stream spent waiting in finish() for the gradient all-reduces (event-timed, GPU timeline; ~0 = fully
    hidden under backward), and rank 0's DataParallel diagnostics (side-stream overlap probe, measured NT launch form, buckets).
    Works on any backend (objects are gathered): tests/_rank_probe.py drives it over gloo."""
    world = dist.get_world_size()
    rows = [None] * world
    dist.all_gather_object(rows, {"device": device_index, "ms": round(float(own_ms), 3), "finish_wait_ms": round(float(finish_wait_ms), 3)})
    info = {"world_size": world, "backend": dist.get_backend(), "devices": [r["device"] for r in rows],
            "per_rank_ms": [r["ms"] for r in rows], "finish_wait_ms": [r["finish_wait_ms"] for r in rows]}
    info.update(diagnostics)
    return info


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, script, script_args, env=None, capture=False):
    """Start `n` fresh rank processes of `script` on this node (one per GPU) through torch.distributed.run and relay their
    output; returns the launcher's exit code.  Called by a parent that has NOT touched the GPU: the ranks are new child
    processes (never an exec of an initialised one), rendezvous on 127.0.0.1."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, *script_args]
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "4")
    if capture:                                     # tests: (exit code, everything the ranks printed)
        r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return r.returncode, r.stdout
    return subprocess.run(cmd, env=e).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the TiTok-S / ViT-VQGAN-B side measurements")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this parent only launches the N ranks (no GPU call happens in it) and exits with their code
        sys.exit(spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to mislabel the result")
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = max(1, torch.cuda.device_count())
    if world > ndev and os.environ.get("GPU_MAX_HW_QUEUES"):
        # several ranks per card AND a raised hardware-queue count: the combination that hung in round 2 (vitamd.ddp.check_hw_queues)
        sys.exit(f"bench.py: {world} ranks on {ndev} GPU(s) with GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']} set: refusing "
                 "(oversubscribed hardware queues hang this step; unset the variable or run one rank per GPU)")
    dev = torch.device("cuda", local % ndev)       # (a rehearsal on a 1-GPU box may run several ranks on one card)
    torch.cuda.set_device(dev)
    if world > 1:
        # The step uses two HIP streams (input-gradient chain / weight-gradient GEMMs) and RCCL brings its own.  HIP multiplexes streams
        # onto 4 hardware queues in order of first use; two streams on one queue are serialised with barrier packets.  Measured on one
        # MI355X (tools/ddp_bisect.py): with RCCL initialised BEFORE the side stream's first use the side stream shared the main stream's
        # queue - 35.5 instead of 31.4 ms/step on every rank.  So the side stream runs its first kernel here, before RCCL exists.
        # (GPU_MAX_HW_QUEUES=8 cures it too with one rank per GPU, but hangs ranks that share a GPU: refused above and in vitamd.ddp.)
        from vitamd import functions as _F
        _F.claim_streams(dev)
        backend = os.environ.get("VITAMD_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import train_vit as TV
    from vitamd.ddp import DataParallel
    from vitamd.functions import WEIGHTS

    torch.manual_seed(0)
    model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0), num_classes=1000).to(dev)
    net = DataParallel(model) if world > 1 else model
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)   # each rank owns its shard of the global batch
    images = torch.randn(PER_GPU_BATCH, 3, 224, 224, generator=g).to(dev)
    labels = torch.randint(0, 1000, (PER_GPU_BATCH,), generator=g).to(dev)

    finish_events = []                       # (before, after) net.finish() on the main stream: how long backward's tail waits for the all-reduces

    def step(timed=False):
        model.zero_grad(set_to_none=True)
        WEIGHTS.clear()                      # weights are re-cast to bf16 every step, as in training
        loss = torch.nn.functional.cross_entropy(net(images), labels)
        loss.backward()
        if world > 1:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            net.finish()
            if timed:
                e1.record()
                finish_events.append((e0, e1))
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(timed=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_ms = elapsed / args.steps * 1e3
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())
    dist_info = {"world_size": 1, "backend": backend, "devices": [dev.index]}
    if world > 1:
        wait_ms = sum(a.elapsed_time(b) for a, b in finish_events) / max(1, len(finish_events))
        dist_info = gather_dist_info(dev.index, own_ms, wait_ms, net.diagnostics())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = PER_GPU_BATCH * world / (ms / 1e3)
        out = {
            "metric": "images/sec ViT-B/16 224px bf16 fwd+bwd", "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ViT-B/16 224x224 bf16 forward+loss+backward, batch 256 per GPU (BASELINE configs[1]; "
                                   "configs[2] at 8 GPUs), random-init weights, 197 tokens, 79.44 M params",
                       "global_batch": PER_GPU_BATCH * world, "parallelism": f"dp{world}"},
            "model_flops_frac_of_peak": round(value / world * GFLOP_PER_IMAGE / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(loss_val, 4), "dist": dist_info,
            "nt_seam_probe": __import__("vitamd.ops", fromlist=["SEAM_PROBE"]).SEAM_PROBE.get(dev.index),     # start-up A/B of the seam form on this device (vitamd.ops.seam_probe)
        }
        try:                                   # side information must never cost the headline line
            out["device"] = box_identity(dev)
        except Exception as e:                 # noqa: BLE001
            out["device"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_also:
            try:
                out["also"] = also_models(dev)
            except Exception as e:             # noqa: BLE001
                out["also"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_roofline:
            try:
                out["roofline"] = kernel_roofline(dev)
            except Exception as e:             # noqa: BLE001
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:             # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
