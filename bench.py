#!/usr/bin/env python3
"""Benchmark of the extraction hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: BASELINE.json configs[1] - standard TDNN x-vector, 80-dim fbank,
256 utterances x 200 frames per GPU, bf16 MFMA (f32 accumulate, f32 pooled tail).
Weak scaling: every rank extracts its own 256-utterance shard; with N > 1 the embeddings are
collected with one RCCL all-gather per step (the path's only exchange).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, hipEvent-timed inside libasv_amd.so on the extract stream, during the timed steps)
and `cpu_baseline` (torch-CPU port of the reference's per-utterance path, N=1 only).
"""

import argparse
import json
import os
import sys
import time

# RCCL / device-tensor sharing across the ranks of one node needs dmabuf IPC on these hosts (already exported by the
# launch environment; set here too so a bare `torchrun bench.py` works)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(REPO, "asv-subtools_amd", "pytorch"), REPO]

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=None,
                    help="utterances per GPU per step (default: 640 for the x-vector - 640 x 204 padded rows = 1020 row tiles of 128, two "
                         "column tiles each, fill the 512 workgroup slots of the device in whole rounds, +15 %% over 256; 256 for the others)")
    ap.add_argument("--frames", type=int, default=None, help="frames per utterance (default: 200; 300 for --model ecapa, BASELINE configs[2])")
    ap.add_argument("--feat-dim", type=int, default=80)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel hipEvents in the timed region")
    ap.add_argument("--event-stride", type=int, default=8, help="record the per-GEMM hipEvents on every k-th timed step")
    ap.add_argument("--per-op", action="store_true", help="also print a per-op timing table to stderr")
    ap.add_argument("--settle-seconds", type=float, default=0.5,
                    help="untimed steps run for this long before the W warmup steps: the power management needs ~100 ms of sustained load "
                         "to reach the steady clock; the first 100 steps after idle are 30 %% slower than the steady state")
    ap.add_argument("--from-wav", action="store_true",
                    help="start every step from 16-bit PCM in HBM: asv_fbank_pcm16 (log-mel, --feat-dim bins) + asv_cmvn + extraction "
                         "(SURVEY.md 8(f) rank 2; the headline metric starts from feature matrices, this is a supplementary line)")
    ap.add_argument("--model", default="xvector", choices=["xvector", "ecapa", "resnet"],
                    help="xvector = BASELINE configs[1] (the default, the contract's workload); ecapa = configs[2] (C=1024, 300 frames); "
                         "resnet = the configs[4] extractor (ResNet34-SE)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import libs.support.utils as utils
    from libs.amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    # ---- model: the reference's standard x-vector blueprint, synthetic weights ---------------
    if args.model == "xvector":
        blueprint, creation = "xvector.py", "Xvector(%d,10,training=False)" % args.feat_dim
    elif args.model == "ecapa":
        blueprint, creation = "ecapa_tdnn_xvector.py", "ECAPA_TDNN(%d,10,training=False)" % args.feat_dim
    else:
        blueprint = "resnet_xvector.py"
        creation = ("ResNetXvector(%d,10,training=False,resnet_params={'use_se':True,'se_ratio':4,'full_pre_activation':False},"
                    "fc2_params={'nonlinearity':'','bn_params':{'momentum':0.5,'affine':False,'track_running_stats':True}})" % args.feat_dim)
    model = utils.create_model_from_py(os.path.join(REPO, "asv-subtools_amd", "pytorch", "model", blueprint), creation)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, 0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    model.cuda()
    model.amd_precision = args.precision
    eng = model._amd_engine()

    # ---- synthetic batch, device resident ----------------------------------------------------
    if args.frames is None:
        args.frames = 300 if args.model == "ecapa" else 200
    if args.batch is None:
        args.batch = 640 if args.model == "xvector" else 256
    B, T, D = args.batch, args.frames, args.feat_dim
    mats = [synth.synth_feats(T, D, 10_000 * rank + i) for i in range(B)]
    feats = torch.from_numpy(np.concatenate(mats, axis=0)).to(dev)
    offsets = (np.arange(B + 1) * T).astype(np.int32)
    # two output / gather buffer pairs: the all-gather of step i runs on RCCL's stream while step i+1 computes
    outs = [torch.empty((B, eng.embed_dim), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * B, eng.embed_dim), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    counter = [0]

    if args.from_wav:
        from libs.amd import frontend
        n_samp = 400 + (T - 1) * 160                                # 25 ms windows, 10 ms shift at 16 kHz: exactly T frames
        wave_dev = torch.from_numpy(np.concatenate([synth.synth_wave(n_samp, 10_000 * rank + i).astype(np.int16) for i in range(B)])).to(dev)
        sample_off = np.arange(B + 1, dtype=np.int64) * n_samp
        fe_kw = dict(num_mel_bins=D, energy_floor=0.0, mean_norm=True)

    def extract_once(out):
        if args.from_wav:
            f, _ = frontend.fbank_device(wave_dev, sample_off, **fe_kw)
            eng.extract_device(f, offsets, out=out)
        else:
            eng.extract_device(feats, offsets, out=out)

    def step():
        k = counter[0] & 1
        counter[0] += 1
        if pending[k] is not None:                                  # buffer pair k is free once its gather has finished
            pending[k].wait()                                       # (stream-side wait, the host does not block)
            pending[k] = None
        extract_once(outs[k])
        if world > 1:
            pending[k] = dist.all_gather_into_tensor(gathered[k], outs[k], async_op=True)

    def barrier():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(n, sample_events=0):
        """sample_events = k > 0: per-GEMM hipEvents are recorded on every k-th step of the timed region
        (each recorded event is a barrier packet between kernels; sampling keeps that perturbation small)."""
        barrier()
        t0 = time.perf_counter()
        for i in range(n):
            if sample_events:
                eng.set_profiling(4 if i % sample_events == 0 else 0)
            step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle_seconds:       # untimed: bring the device to its steady clock
        for _ in range(50):
            extract_once(outs[0])                                    # local work only: the count differs between ranks,
        torch.cuda.synchronize(dev)                                   # so no collective may be issued here
    for _ in range(args.warmup):
        step()
    barrier()

    profile = not args.no_profile
    if profile:
        eng.set_profiling(4)                                        # one hipEvent pair around each run of frame-level GEMM launches
        # create the whole event pool outside the timed region: one profiled step per step that will be sampled
        # (hipEventCreate inside the timed steps cost 5-30 % of the measured rate, erratically)
        for _ in range((args.steps + args.event_stride - 1) // args.event_stride):
            step()
        barrier(); eng.get_profile()
    dt = timed(args.steps, sample_events=args.event_stride if profile else 0)
    rows = eng.get_profile() if profile else []
    eng.set_profiling(False)
    if args.per_op and rank == 0:
        eng.set_profiling(2)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        ops = eng.graph.ops
        for r in sorted(eng.get_profile(), key=lambda r: r["op_index"]):
            i = r["op_index"]
            desc = ""
            if 0 <= i < len(ops) and ops[i].kind == "tdnn":
                desc = "%d->%d taps=%s" % (ops[i].inp.channels, ops[i].out.channels, ops[i].taps)
            us = 1e3 * r["total_ms"] / max(r["launches"], 1)
            tf = r["flops"] / (r["total_ms"] * 1e-3) / 1e12 if r["total_ms"] > 0 else 0.0
            print("  op %3d %-14s %-28s %9.1f us  %8.1f TFLOP/s" % (i, r["name"], desc, us, tf), file=sys.stderr)
        eng.set_profiling(False)
    dt_plain = timed(args.steps)                                      # same steps without event recording, for reference

    utts = world * B * args.steps
    value = utts / dt
    res = {
        "metric": "utterances/sec (200-frame) embedding extraction + EER, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "utterances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "%s: %s, %d-dim fbank, %d utterances x %d frames per GPU per step, "
                               "%s resident in HBM, f32 embeddings out%s" % ({"xvector": "BASELINE configs[1]: standard TDNN x-vector", "ecapa": "BASELINE configs[2]: ECAPA-TDNN C=1024",
                                                                                   "resnet": "BASELINE configs[4] extractor: ResNet34-SE"}[args.model],
                                                                                  creation, D, B, T, "16-bit PCM (fbank + CMN computed on the device in every step)" if args.from_wav else "features",
                                                                                  ", + RCCL all-gather of embeddings" if world > 1 else ""),
                   "global_batch_utts": world * B, "frames_per_utt": T, "parallelism": "utterance shards x%d" % world},
        "value_without_event_recording": round(utts / dt_plain, 1),
        "settle_seconds": args.settle_seconds,
    }
    gemm = next((r for r in rows if r["name"] == "tdnn_gemm"), None)
    if gemm and gemm["total_ms"] > 0:
        achieved = gemm["flops"] / (gemm["total_ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        per_frame, per_utt = eng.graph.flops_per_frame()
        kname = {"xvector": "tdnn_gemm_big3_kernel (the frame-level layers tdnn1-5: 5 launches per step, the last one with the fused pooling epilogue)",
                 "ecapa": "frame-level GEMM launches (tdnn_gemm_big3_kernel for the wide layers, tdnn_gemm_kernel for the 128-channel ones)",
                 "resnet": "frame-level GEMM launches (grid_conv_narrow_kernel / tdnn_gemm_kernel)"}[args.model]
        res["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                           "launches": gemm["launches"], "avg_launch_us": round(1e3 * gemm["total_ms"] / gemm["launches"], 2),
                           "algorithmic_gflop_per_utt": round((per_frame * T + per_utt) / 1e9, 4)}
        sampled_steps = (args.steps + args.event_stride - 1) // args.event_stride
        res["roofline"]["sampled_steps"] = sampled_steps
        res["gemm_ms_per_step"] = round(gemm["total_ms"] / sampled_steps, 4)
        pmc = os.path.join(REPO, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and args.model == "xvector":
            # HBM traffic of the dominant kernel from a separate rocprofv3 --pmc pass of this same command
            # (tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
            with open(pmc) as f:
                info = json.load(f)
            res["roofline"]["traffic"] = info.get("traffic_bytes_per_launch")
            res["roofline"]["traffic_source"] = info.get("source")

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import torch_cpu_port as P                        # cpu_baseline leg only
        # torch's default (one thread per core) collapses on many-core hosts for these small
        # convolutions: probe a few thread counts briefly and time the sample with the best one
        if args.model != "xvector":
            raise SystemExit("cpu_baseline is implemented for the default workload only; pass --cpu-seconds 0 with --model %s" % args.model)
        ex = P.XvectorCpu(sd, "far")
        host = os.cpu_count() or 1
        best, cores = 0.0, 1
        for th in sorted({1, min(8, host), min(16, host), min(32, host)}):
            torch.set_num_threads(th)
            ups, _, _ = P.time_cpu_baseline(ex, mats[:8], budget_s=min(1.5, args.cpu_seconds / 6.0), min_utts=2)
            if ups > best:
                best, cores = ups, th
        torch.set_num_threads(cores)
        ups, n, secs = P.time_cpu_baseline(ex, mats[:64], budget_s=args.cpu_seconds)
        res["cpu_baseline"] = {"value": round(ups, 2), "unit": "utterances/s", "cores": cores, "kind": "port",
                               "sample": "%d utterances of the same %dx%d workload, batch=1 loop as pipeline/onestep/extract_embeddings.py:73-83, "
                                         "torch %s CPU with the best of {1,8,16,32} threads on a %d-core host, %.1f s" % (n, T, D, torch.__version__, host, secs)}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
