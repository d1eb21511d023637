"""bench.py -- images/sec of one `train_hallucidet` step (Faster R-CNN, LLVIP geometry 512x640, batch 8 / GPU, fp16) on
N MI355X of one node, one process per GPU (RCCL all-reduce of the hallucination-net gradients), synthetic inputs already
resident in HBM.  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`roofline`: the dominant kernel is the convolution behind hd_conv2d (conv_igemm_kernel, and conv3x3_small_kernel for the
16/32-channel 3x3 layers: every conv / data-gradient / FC of the U-Net and the detector).  Its launches of one step are recorded and re-issued back to back between HIP events on
the launch stream; achieved = algorithmic FLOPs of those launches / their summed duration (DESIGN.md "Measurement").
`cpu_baseline`: the CPU oracle (oracle/step.py, "port") timed on this host's cores on a bounded sample of the same
workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F16_PEAK_TFLOPS = 2500.0     # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
BATCH_PER_GPU = 8
H, W = 512, 640


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in two separate
    runs, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py) committed under profiles/ for this round.
    bench.py itself cannot collect PMCs (they need rocprofv3 around the process): null if the file is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_conv_traffic.json")
    try:
        return round(json.load(open(path))["traffic_bytes_per_launch"])
    except Exception:
        return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def conv_roofline(lit, batch, reps=5):
    """Record every hd_conv2d launch of one (eager) training step, then time the recorded launches back to back."""
    import ctypes as C
    from hallucidet_amd import _abi, ops
    lib = _abi.load()
    rec = []
    orig = ops.conv2d

    def spy(x, w, KH, KW, **kw):
        out = orig(x, w, KH, KW, **kw)
        y = out[0] if isinstance(out, tuple) else out
        C1 = x.shape[3]
        C2 = 0 if kw.get("x2") is None else kw["x2"].shape[3]
        if kw.get("out_nchw_f32"):
            n, co, ho, wo = y.shape
        else:
            n, ho, wo, co = y.shape
        dil = kw.get("in_dil", 1)
        # algorithmic FLOPs: a data-gradient over a zero-dilated input only multiplies the non-zero taps
        flops = 2.0 * n * ho * wo * co * KH * KW * (C1 + C2) / (dil * dil)
        # algorithmic bytes: every operand once (x, x2, w, y, residual, mask)
        by = x.numel() * 2 + (0 if kw.get("x2") is None else kw["x2"].numel() * 2) + w.numel() * 2 + y.numel() * y.element_size()
        by += sum(t.numel() * 2 for t in (kw.get("res"), kw.get("mask")) if t is not None)
        rec.append((dict(kw), (x, w, KH, KW), flops, by, where[0]))
        return out

    # weight-gradient launches of the hallucination network (hd_wgrad): part of its "conv blocks" (BASELINE north_star)
    wrec = []
    orig_wg = ops.wgrad

    def spy_wg(x, dy, KH, KW, **kw):
        out = orig_wg(x, dy, KH, KW, **kw)
        C2 = 0 if kw.get("x2") is None else kw["x2"].shape[3]
        wrec.append(((x, dy, KH, KW), dict(kw), 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * KH * KW * (x.shape[3] + C2)))
        return out

    where = ["detector"]
    rr = lit.encoder_decoder.runner
    o_fwd, o_bwd = rr.forward, rr.backward

    def tag(fn):
        def g(*a, **k):
            where[0] = "unet"
            try:
                return fn(*a, **k)
            finally:
                where[0] = "detector"
        return g
    rr.forward, rr.backward = tag(o_fwd), tag(o_bwd)
    ops.conv2d = spy
    ops.wgrad = spy_wg
    import hallucidet_amd.models.detection as det_mod
    import hallucidet_amd.segmentation_models.unet as unet_mod
    r = lit.encoder_decoder.runner
    was = r.use_graphs
    r.enable_graphs(False)
    try:
        lit.fit_step(batch)
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        ops.wgrad = orig_wg
        rr.forward, rr.backward = o_fwd, o_bwd
        r.enable_graphs(was)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot_ms, tot_fl, tot_by = 0.0, 0.0, 0.0
    dump = os.environ.get("HD_BENCH_DUMP")
    rows = []
    grp = {"unet": [0.0, 0.0], "detector": [0.0, 0.0]}
    for kw, (x, w, KH, KW), fl, by, origin in rec:
        tot_by += by
        kw = dict(kw)                                 # same epilogue (bias / res / mask / BN statistics) as in the step
        orig(x, w, KH, KW, **kw)                      # warm
        e0.record()
        for _ in range(reps):
            orig(x, w, KH, KW, **kw)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot_ms += ms
        tot_fl += fl
        grp[origin][0] += ms
        grp[origin][1] += fl
        if dump:
            rows.append((ms * 1e3, fl / 1e9, tuple(x.shape), 0 if kw.get("x2") is None else kw["x2"].shape[3], tuple(w.shape), KH,
                         kw.get("stride", 1), kw.get("in_dil", 1), bool(kw.get("up1")), bool(kw.get("want_stats")),
                         kw.get("mask") is not None, kw.get("res") is not None))
    if dump:
        with open(dump, "w") as f:
            for r_ in rows:
                f.write("%8.1f us %8.2f GF %7.1f TF  x=%s c2=%d w=%s k=%d s=%d dil=%d up=%d stats=%d mask=%d res=%d\n"
                        % (r_[0], r_[1], r_[1] / r_[0] * 1e-3 * 1e3, *r_[2:]))
    n = len(rec)
    achieved = tot_fl / (tot_ms * 1e-3) / 1e12
    wg_ms = wg_fl = 0.0
    for (x, dy, KH, KW), kw, fl in wrec:
        orig_wg(x, dy, KH, KW, **kw)
        e0.record()
        for _ in range(reps):
            orig_wg(x, dy, KH, KW, **kw)
        e1.record()
        e1.synchronize()
        wg_ms += e0.elapsed_time(e1) / reps
        wg_fl += fl
    tf = lambda fl, ms: round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None
    u_ms, u_fl = grp["unet"][0] + wg_ms, grp["unet"][1] + wg_fl
    groups = {
        "unet_conv_fwd_dgrad": {"ms": round(grp["unet"][0], 3), "tflops": tf(grp["unet"][1], grp["unet"][0])},
        "unet_wgrad": {"ms": round(wg_ms, 3), "tflops": tf(wg_fl, wg_ms), "kernel": "wgrad_kernel"},
        "unet_conv_blocks_total": {"ms": round(u_ms, 3), "tflops": tf(u_fl, u_ms), "frac": round(u_fl / (u_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4) if u_ms else None,
                                   "gflop_per_image": round(u_fl / 1e9 / BATCH_PER_GPU, 1)},
        "detector_conv": {"ms": round(grp["detector"][0], 3), "tflops": tf(grp["detector"][1], grp["detector"][0])},
    }
    return {"bound": "mfma", "groups": groups, "achieved": round(achieved, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": _pmc_traffic(), "traffic_unit": "HBM bytes per launch (PMC)",
            "alg_bytes_per_launch": round(tot_by / max(n, 1)),
            "kernel": "hd_conv2d: conv_igemm_kernel (implicit-GEMM conv / dgrad / FC) + conv3x3_small_kernel (16/32-channel 3x3 layers)", "launches_per_step": n,
            "avg_launch_us": round(tot_ms * 1e3 / max(n, 1), 2), "avg_launch_gflop": round(tot_fl / max(n, 1) / 1e9, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--detector", default="fasterrcnn", choices=["fasterrcnn", "retinanet"],
                    help="fasterrcnn = the configuration BASELINE.json's metric is quoted on (default); retinanet = configs[3]")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU path")
    torch.cuda.set_device(local)
    force_dist = os.environ.get("HD_FORCE_DIST") == "1"      # exercise the RCCL path on one GPU (world size 1)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from hallucidet_amd import synthetic
    dev = "cuda:%d" % local
    lit = synthetic.make_module(seed=123, device=dev, precision=16, detector_name=args.detector)
    batch = synthetic.make_batch(BATCH_PER_GPU, H, W, seed=123 + rank, device=dev)   # per-rank shard, resident in HBM

    for _ in range(args.warmup):
        lit.fit_step(batch)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = lit.fit_step(batch)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    if not torch.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")

    if rank == 0:
        # everything below is rank-0-only measurement: no collective may be issued from here on (the other ranks are
        # already waiting at the final barrier), so the gradient averager is detached for the extra steps
        lit.averager.start = lambda g: None
        lit.averager.finish = lambda g: None
        value = BATCH_PER_GPU * world * args.steps / elapsed
        out = {
            "metric": "images/sec train_hallucidet (640x512, batch 8/GPU)",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": ("train_hallucidet.py fasterrcnn LLVIP batch=8 fp16 on 1xMI355X (BASELINE configs[1]); "
                                    "U-Net resnet34 fwd+bwd, 3 frozen Faster R-CNN R50-FPN passes @300x300, loss scaling, "
                                    "value clip 0.5, Adam") if args.detector == "fasterrcnn" else
                                   ("train_hallucidet.py retinanet batch=8 fp16 (BASELINE configs[3], NOT the headline config); "
                                    "U-Net resnet34 fwd+bwd, 3 frozen RetinaNet R50-FPN passes @300x300, loss scaling, clip, Adam"),
                       "global_batch": BATCH_PER_GPU * world, "image": "1x512x640 IR + 3x512x640 RGB",
                       "parallelism": "dp%d" % world, "alg_gflop_per_image": 428.5 if args.detector == "fasterrcnn" else None,
                       "step_alg_tflops": round(428.5e9 * value / 1e12, 1) if args.detector == "fasterrcnn" else None},
            "final_loss": round(float(loss), 5),
        }
        if world == 1:
            # PCIe-inclusive rate (never `value`): the same step with the batch copied from pinned host memory every step
            hb = [t.cpu().pin_memory() if torch.is_tensor(t) else [{k: v.cpu().pin_memory() for k, v in d.items()} for d in t] for t in batch]
            def h2d_step():
                db = [t.to(dev, non_blocking=True) if torch.is_tensor(t) else [{k: v.to(dev, non_blocking=True) for k, v in d.items()} for d in t] for t in hb]
                lit.fit_step(db)
            for _ in range(2):                     # first copies out of freshly pinned pages are not representative
                h2d_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                h2d_step()
            torch.cuda.synchronize()
            out["pcie_inclusive_images_per_s"] = round(BATCH_PER_GPU * 10 / (time.perf_counter() - t1), 2)
        if not args.no_roofline:
            out["roofline"] = conv_roofline(lit, batch)
        if world == 1 and not args.no_cpu_baseline:
            from oracle.step import time_cpu_step
            cb = synthetic.make_batch(2, H, W, seed=123, device="cpu")
            v, steps, cores = time_cpu_step(cb, budget_s=25.0, max_steps=2)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "images/s", "cores": cores, "kind": "port",
                                   "sample": "%d full training step(s) of the CPU oracle (oracle/step.py, fp32 torch) on a batch of 2 "
                                             "synthetic 512x640 images; host: %s" % (steps, _cpu_model())}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()                 # ranks > 0 wait here while rank 0 finishes its roofline / baseline legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
