#!/usr/bin/env python3
"""Headline benchmark: trials/sec end-to-end on synthetic EEG (BASELINE.json metric).

One step = one pass of the hot path over one batch that is already resident in HBM:
  extract_features (Butterworth filterbank -> STFT 64/32 -> log band power, HIP)
  -> Conv4Layers(nb*C, 32) + Linear(32, 5) forward -> softmax-CE -> backward (HIP)
  -> [N > 1: one flat-bucket RCCL all-reduce] -> AdamW step (torch, fused).
Workload = BASELINE config 2: 4096 trials per GPU, 64 ch, 2 s @ 256 Hz, 9 bands, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3     # fp32 vector == fp32-input MFMA peak


def synth_trials(B, C, T, fs, seed):
    """SURVEY.md 8d synthetic EEG: unit white noise + 0.5 sin(2 pi f_y t + phi) on the channels of zone y."""
    import isd_amd
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, C, T), dtype=np.float32)
    y = rng.integers(0, 5, B).astype(np.uint8)
    tone = np.array([6.0, 10.0, 18.0, 26.0, 34.0])
    phase = rng.uniform(0.0, 2.0 * np.pi, B)
    t = np.arange(T) / fs
    zones = isd_amd.zone_index_lists()
    for i in range(B):
        ch = [c for c in zones[int(y[i])] if c < C]
        X[i, ch] += (0.5 * np.sin(2.0 * np.pi * tone[y[i]] * t + phase[i])).astype(np.float32)
    return X, y


def cpu_baseline(C, T, fs, n_trials=128, steps=16):
    """The oracle (CPU restatement of the same pipeline) timed on this box's host cores, bounded sample."""
    from oracle import cnn as ocnn, dsp as odsp
    # the GPU box grants a 16-CPU share per GPU (cpu_count reports the whole host)
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(threads)
    X, y = synth_trials(n_trials, C, T, fs, seed=123)
    p = ocnn.init_conv4_params(9 * C, 32, prefix="cnn.", seed=0)
    p["fc.weight"] = torch.randn(5, 32) * 0.1
    p["fc.bias"] = torch.zeros(5)
    for v in p.values():
        v.requires_grad_()
    opt = torch.optim.AdamW(list(p.values()), lr=5e-4)
    yt = torch.from_numpy(y)
    t_feat = t_cnn = 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        feats = torch.from_numpy(odsp.extract_features_scipy(X, fs=fs, bands=odsp.BANDS_9))
        t1 = time.perf_counter()
        opt.zero_grad()
        ocnn.cross_entropy(ocnn.feature_cnn_logits(feats, p), yt).backward()
        opt.step()
        t2 = time.perf_counter()
        t_feat += t1 - t0
        t_cnn += t2 - t1
    total = t_feat + t_cnn
    return {"value": round(n_trials * steps / total, 2), "unit": "trials/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps x {n_trials} trials of the same workload through oracle/ (scipy butter/sosfilt/"
                      f"stft fp64 on 1 thread: {t_feat / steps:.2f} s/step; torch-CPU CNN fwd+bwd+AdamW on {threads} "
                      f"threads: {t_cnn / steps:.2f} s/step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="trials per GPU")
    ap.add_argument("--two-kernel", action="store_true", help="materialise the filtered signals (fb + bandpower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="config 3: bf16 activations/grads in the CNN, fp32 accumulate")
    ap.add_argument("--overlap", action="store_true",
                    help="extract the features of the next batch on a second HIP stream while the CNN trains on the "
                         "current one (measured: no gain on MI355X -- the fused extractor already fills every wave "
                         "slot -- so the default is one stream)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N > 1 only: wait for the gradient all-reduce before extracting the next batch's features "
                         "(default: the all-reduce of step k runs under the feature extraction of batch k+1)")
    args = ap.parse_args()

    import torch.distributed as dist
    import isd_amd
    from isd_amd.classifier import _FeatureModel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    # ISD_DIST_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks share a card)
    backend = os.environ.get("ISD_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    B, C, T, fs = args.batch, 64, 512, 256.0
    nb = len(isd_amd.BANDS_9)
    Xh, yh = synth_trials(B, C, T, fs, seed=rank)               # rank r uses default_rng(r)
    x = torch.from_numpy(Xh).to(dev)
    y = torch.from_numpy(yh).to(dev)
    del Xh

    torch.manual_seed(42)                                        # reference default seed (train_fast.py:275)
    fx = isd_amd.FeatureExtractor(T, fs, isd_amd.BANDS_9)
    model = _FeatureModel(nb * C, 32, 5, 4, "bf16" if args.bf16 else "f32").to(dev)
    trainer = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2, schedule=None)
    feats = torch.empty((B, nb, C, fx.n_frames), dtype=torch.float32, device=dev)
    yfilt = torch.empty((B, nb, C, T), dtype=torch.float32, device=dev) if args.two_kernel else None
    global_batch = B * world
    fused = not args.two_kernel

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    overlap = fused and args.overlap
    main_stream = torch.cuda.current_stream()
    feat_stream = torch.cuda.Stream() if overlap else main_stream
    fbuf = [feats, torch.empty_like(feats)] if overlap else [feats, feats]
    ready = [torch.cuda.Event(), torch.cuda.Event()]       # features of buffer k are complete
    freed = [torch.cuda.Event(), torch.cuda.Event()]       # the CNN step that read buffer k is complete

    def extract(k, e=None):
        """one feature-extraction pass over the resident batch into buffer k, on the feature stream"""
        with torch.cuda.stream(feat_stream):
            if overlap:
                feat_stream.wait_event(freed[k])
            if e:
                e[0].record(feat_stream)
            if fused:
                fx(x, fused=True, out=fbuf[k])
            else:
                fx.fb.forward(x, out=yfilt)
                if e:
                    e[3].record(feat_stream)
                fx.stft.bandpower(yfilt, fx.bins, out=fbuf[k])
            if e:
                e[1].record(feat_stream)
            ready[k].record(feat_stream)

    def train(k, e=None):
        if overlap:
            main_stream.wait_event(ready[k])
        out = trainer.step(fbuf[k].view(B, nb * C, fx.n_frames), y, global_batch=global_batch)
        freed[k].record(main_stream)
        if e:
            e[2].record(main_stream)
        return out

    # N > 1: the features do not depend on the parameters, so the one exchange step of the iteration (the flat
    # gradient all-reduce, RCCL's own stream) is started right after the backward pass and waited for only after
    # the next batch's features are queued: forward/backward(k) -> all-reduce(k) || extract(k+1) -> AdamW(k).
    pipelined = world > 1 and not overlap and not args.no_pipeline

    def pipelined_step(e=None):
        out = trainer.step_begin(feats.view(B, nb * C, fx.n_frames), y, global_batch=global_batch)
        extract(0, e)                                       # same stream: queued behind the backward pass that read feats
        trainer.step_finish()
        if e:
            e[2].record(main_stream)
        return out

    # Every step = one feature-extraction pass + one CNN fwd/bwd/optimizer pass.  With overlap the extraction
    # of the NEXT batch runs on its own stream under the CNN work of the current one (the features do not
    # depend on the parameters), exactly K of each inside the timed region.
    freed[0].record(main_stream); freed[1].record(main_stream)
    if overlap or pipelined:
        extract(0)
    for i in range(args.warmup):
        if pipelined:
            pipelined_step()
        elif overlap:
            extract((i + 1) % 2)
            train(i % 2)
        else:
            extract(0)
            train(0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base = args.warmup
    for i in range(args.steps):
        if pipelined:
            out = pipelined_step(ev[i])
        elif overlap:
            extract((base + i + 1) % 2, ev[i])
            out = train((base + i) % 2, ev[i])
        else:
            extract(0, ev[i])
            out = train(0, ev[i])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss = float(out["loss"])

    if rank == 0:
        ms = dt / args.steps * 1e3
        t_feat = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        t_train = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if not (overlap or pipelined) else float("nan")
        if fused:
            # dominant kernel: the fused filterbank+STFT extractor; VALU-bound on its algorithmic traffic, so the
            # compute roofline is the honest one: flops = cascade (5 flop/sample/section incl. fix-up) + band DFT
            # algorithmic flops: cascade 9 flop/sample/section (3-op recursion + 2-FMA fix-up) + 1 (gain),
            # band DFT 2 complex partial sums x 64 samples per (frame, bin)
            n_sec, bins = 4, sum(hi - lo + 1 for lo, hi in fx.bins)
            flops = B * C * (nb * T * (n_sec * 9 + 1) + bins * fx.n_frames * 64 * 4)
            roof = {"bound": "mfma", "kernel": "fused_kernel<float> (fp32 VALU; fp32 vector peak == fp32 MFMA peak)",
                    "achieved": round(flops / (t_feat * 1e-3) / 1e12, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(flops / (t_feat * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4), "traffic": None,
                    "ms_per_launch": round(t_feat, 4),
                    "algorithmic_bytes_per_launch": B * C * 4 * (T + nb * fx.n_frames)}
        else:
            t_fb = float(np.mean([e[0].elapsed_time(e[3]) for e in ev]))
            by = (1 + nb) * C * T * 4 * B                         # read x once + write nb filtered copies
            roof = {"bound": "hbm", "kernel": "fb_kernel<float,1>", "achieved": round(by / (t_fb * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (t_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": None, "ms_per_launch": round(t_fb, 4), "algorithmic_bytes_per_launch": by}
        tf = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tf):
            try:
                ent = json.load(open(tf)).get(roof["kernel"].split(" ")[0])
                roof["traffic"] = ent["bytes_per_launch"] if ent and B == 4096 else None
            except Exception:
                pass
        line = {
            "metric": "trials/sec end-to-end (filterbank+CNN fwd+bwd)", "value": round(global_batch * args.steps / dt, 1),
            "unit": "trials/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 activations/grads in the CNN, f32 features + accumulate" if args.bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: 64ch x 2s@256Hz EEG, 9-band Butterworth(4) filterbank -> STFT(64/32) "
                                   "log band power -> Conv4Layers(576,32)+Linear(32,5) fwd+bwd, softmax-CE, AdamW",
                       "trials_per_gpu": B, "global_batch": global_batch, "parallelism": f"dp{world}",
                       "feature_path": "fused" if fused else "filterbank+bandpower kernels",
                       "streams": "features of batch k+1 overlap the CNN step of batch k" if overlap else
                                  "one stream; gradient all-reduce of step k (RCCL stream) under the feature "
                                  "extraction of batch k+1" if pipelined else "one stream"},
            "stages_ms": {"extract_features": round(t_feat, 4),
                          "cnn_fwd_bwd_allreduce_adamw": None if (overlap or pipelined) else round(t_train, 4)},
            "final_loss": round(loss, 5),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(C, T, fs)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
