#!/usr/bin/env python3
"""Headline benchmark: trials/sec end-to-end on synthetic EEG (BASELINE.json metric).

One step = one pass of the hot path over one batch that is already resident in HBM:
  extract_features (Butterworth filterbank -> STFT -> log band power, HIP)
  -> CNN classifier forward -> softmax-CE -> backward (HIP)
  -> [N > 1: one flat-bucket RCCL all-reduce] -> AdamW step (torch, fused).

Workloads (``--config``):
  cfg2 (default; the configuration BASELINE.json's metric is quoted on): 4096 trials per GPU, 64 ch, 2 s @ 256 Hz,
        9 bands, STFT 64/32, Conv4Layers(576, 32) + Linear(32, 5), fp32   (``--bf16``: BASELINE config 3)
  cfg5 (high-resolution stress): 2048 trials per GPU, 128 ch, 4 s @ 1024 Hz, 40 bands, STFT 1024/960,
        EEGNet_Encoder(5120, 32) + Linear(32, 5), fp32

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no WORLD_SIZE in the environment: this process starts the N ranks itself as
                                       CHILD processes through torch.distributed.run, never touches the GPU, and
                                       exits with their code -- rank 0's JSON line is the only stdout)

Before the W warm-up steps every workload runs a time-based GPU pre-roll of the SAME step (>= ``--preroll-ms``,
default 100 ms, disclosed in the line as ``preroll_ms`` / ``preroll_steps``): a fresh box's first milliseconds are
spent at idle clocks and on first-touch page faults, and W = 5 steps of 1.3 ms do not get past them.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC between the ranks' processes (RCCL over xGMI); must be set before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3     # fp32 vector peak (packed v_pk_fma_f32) == fp32-input MFMA peak

CONFIGS = {
    "cfg2": dict(C=64, T=512, fs=256.0, bands="BANDS_9", nperseg=64, noverlap=32, batch=4096, steps=20, warmup=5,
                 workload="cfg2: 64ch x 2s@256Hz EEG, 9-band Butterworth(4) filterbank -> STFT(64/32) log band power "
                          "-> Conv4Layers(576,32)+Linear(32,5) fwd+bwd, softmax-CE, AdamW"),
    "cfg5": dict(C=128, T=4096, fs=1024.0, bands="BANDS_40", nperseg=1024, noverlap=960, batch=2048, steps=5, warmup=2,
                 workload="cfg5 (stress): 128ch x 4s@1024Hz EEG, 40-band Butterworth(4) filterbank -> STFT(1024/960) "
                          "log band power [B,5120,65] -> EEGNet_Encoder(5120,32)+Linear(32,5) fwd+bwd (train-mode "
                          "BatchNorm, dropout 0.25), softmax-CE, AdamW"),
}


def synth_trials(B, C, T, fs, seed):
    """SURVEY.md 8d synthetic EEG: unit white noise + 0.5 sin(2 pi f_y t + phi) on the channels of zone y
    (channels 64..127 of the stress configuration are a second copy of the 64-electrode montage)."""
    import isd_amd
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, C, T), dtype=np.float32)
    y = rng.integers(0, 5, B).astype(np.uint8)
    tone = np.array([6.0, 10.0, 18.0, 26.0, 34.0])
    phase = rng.uniform(0.0, 2.0 * np.pi, B)
    t = np.arange(T) / fs
    zones = isd_amd.zone_index_lists()
    for i in range(B):
        ch = [c + o for o in range(0, C, 64) for c in zones[int(y[i])] if c + o < C]
        X[i, ch] += (0.5 * np.sin(2.0 * np.pi * tone[y[i]] * t + phase[i])).astype(np.float32)
    return X, y


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name, cfg):
    """The oracle (CPU restatement of the same pipeline) timed on this box's host cores, bounded sample."""
    from oracle import cnn as ocnn, dsp as odsp
    # the GPU box grants a 16-CPU share per GPU (cpu_count reports the whole host)
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(threads)
    C, T, fs = cfg["C"], cfg["T"], cfg["fs"]
    bands = getattr(odsp, cfg["bands"])
    n_trials, steps = (128, 16) if cfg_name == "cfg2" else (3, 1)   # cfg5: ~8 s of scipy per trial
    X, y = synth_trials(n_trials, C, T, fs, seed=123)
    nb = len(bands)
    torch.manual_seed(0)
    if cfg_name == "cfg2":
        p = ocnn.init_conv4_params(nb * C, 32, prefix="cnn.", seed=0)
        p["fc.weight"] = torch.randn(5, 32) * 0.1
        p["fc.bias"] = torch.zeros(5)
        logits_of = lambda feats: ocnn.feature_cnn_logits(feats, p)                       # noqa: E731
    else:
        p = {"temporal_conv.0.weight": torch.randn(8, 1, 1, 64) * 0.1, "spatial_conv.0.weight": torch.randn(16, 1, nb * C, 1) * 0.01,
             "separable_conv.0.weight": torch.randn(16, 1, 1, 16) * 0.2, "separable_conv.1.weight": torch.randn(16, 16, 1, 1) * 0.2,
             "projector.2.weight": torch.randn(32, 16) * 0.2, "projector.2.bias": torch.zeros(32),
             "fc.weight": torch.randn(5, 32) * 0.1, "fc.bias": torch.zeros(5)}
        for name, n in (("temporal_conv.1", 8), ("spatial_conv.1", 16), ("separable_conv.2", 16)):
            p[name + ".weight"], p[name + ".bias"] = torch.ones(n), torch.zeros(n)
            p[name + ".running_mean"], p[name + ".running_var"] = torch.zeros(n), torch.ones(n)
        logits_of = lambda feats: torch.nn.functional.linear(                            # noqa: E731
            ocnn.eegnet_encoder(feats.reshape(feats.shape[0], nb * C, -1), p, training=True), p["fc.weight"], p["fc.bias"])
    train = [v for k, v in p.items() if "running" not in k]
    for v in train:
        v.requires_grad_()
    opt = torch.optim.AdamW(train, lr=5e-4)
    yt = torch.from_numpy(y)
    t_feat = t_cnn = 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        feats = torch.from_numpy(odsp.extract_features_scipy(X, fs=fs, bands=bands, nperseg=cfg["nperseg"],
                                                             noverlap=cfg["noverlap"]))
        t1 = time.perf_counter()
        opt.zero_grad()
        ocnn.cross_entropy(logits_of(feats), yt).backward()
        opt.step()
        t2 = time.perf_counter()
        t_feat += t1 - t0
        t_cnn += t2 - t1
    total = t_feat + t_cnn
    return {"value": round(n_trials * steps / total, 2), "unit": "trials/s", "cores": threads, "cpu_model": cpu_model(),
            "kind": "port",
            "sample": f"{steps} steps x {n_trials} trials of the same workload through oracle/ (scipy butter/sosfilt/"
                      f"stft fp64 on 1 thread: {t_feat / steps:.2f} s/step; torch-CPU classifier fwd+bwd+AdamW on "
                      f"{threads} threads: {t_cnn / steps:.2f} s/step)"}


def hip_event_groups_ms(fn, stream, groups, per_group):
    """Per-launch duration of ``fn``: ``groups`` means over ``per_group`` back-to-back launches, each group between ONE
    pair of HIP events recorded on the launch stream (an event pair around every single ~1 ms launch added up to
    0.15 ms of marker handling to it on some boxes: 1.21 ms where rocprofv3 saw 1.06 ms kernels)."""
    fn()
    fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    for e0, e1 in evs:
        e0.record(stream)
        for _ in range(per_group):
            fn()
        e1.record(stream)
    clk = shader_clock_mhz()                                     # probed while the passes are still running
    torch.cuda.synchronize()
    return [float(e0.elapsed_time(e1)) / per_group for e0, e1 in evs], clk


_clock_stream = None


def shader_clock_mhz(spin_us=300):
    """Shader clock right now, under whatever is queued on the device: a one-wave probe kernel on a second stream
    (isd_shader_clock_probe: shader-clock counter against the constant-rate counter over ~spin_us)."""
    global _clock_stream
    from isd_amd import _lib
    L = _lib.lib()
    if _clock_stream is None:
        _clock_stream = torch.cuda.Stream()
    with torch.cuda.stream(_clock_stream):                       # allocated and filled on the probe's own stream
        out = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(L.isd_shader_clock_probe(out.data_ptr(), int(spin_us), _clock_stream.cuda_stream))
    _clock_stream.synchronize()
    t, r = (int(v) for v in out.tolist())
    return round(t / r * L.isd_wall_clock_khz() / 1000.0, 1) if r > 0 else None


def clock_probe_start(spin_us=300):
    """Queue the shader-clock probe on its own stream WITHOUT waiting for it (for use inside a timed region: one
    one-wave launch, no host synchronisation); ``clock_probe_read`` after the region's final synchronize."""
    global _clock_stream
    from isd_amd import _lib
    L = _lib.lib()
    if _clock_stream is None:
        _clock_stream = torch.cuda.Stream()
    with torch.cuda.stream(_clock_stream):
        out = torch.zeros(2, dtype=torch.int64, device="cuda")
    _clock_stream.synchronize()                                  # the fill is done before the region starts

    def fire():
        _lib.check(L.isd_shader_clock_probe(out.data_ptr(), int(spin_us), _clock_stream.cuda_stream))
    return out, fire


def clock_probe_read(out):
    from isd_amd import _lib
    t, r = (int(v) for v in out.tolist())
    return round(t / r * _lib.lib().isd_wall_clock_khz() / 1000.0, 1) if r > 0 else None


def spread(v):
    """min / median / max of a list of per-step milliseconds"""
    v = [float(a) for a in v]
    return {"min": round(min(v), 4), "median": round(float(np.median(v)), 4), "max": round(max(v), 4)} if v else None


def filterbank_hbm_roofline(fx, x, nb, groups=8, per_group=4):
    """North-star evidence: achieved HBM rate of the MATERIALISING filterbank stage (read x once, write nb filtered
    copies -- what the scipy path does), measured after the timed region on (a slice of) the resident batch.
    ``achieved`` / ``frac`` come from the MEDIAN group (the number of record); the best group and the shader clock
    while the passes ran are given beside it."""
    B, C, T = x.shape
    per_trial = (1 + nb) * C * T * 4
    Bs = int(min(B, max(1, (24 << 30) // per_trial)))                # keep the filtered tensor under 24 GiB
    xs = x[:Bs].contiguous()
    y = torch.empty((Bs, nb, C, T), dtype=torch.float32, device=x.device)
    if per_trial * Bs > (4 << 30):
        groups, per_group = 4, 2                                     # long passes (cfg5: 7 ms each)
    times, clk = hip_event_groups_ms(lambda: fx.fb.forward(xs, out=y), torch.cuda.current_stream(), groups, per_group)
    ms, ms_min = float(np.median(times)), float(min(times))
    by = per_trial * Bs
    if T <= 1024:
        launches = {"f32": "fb_kernel<float,%d>", "f64": "fb_kernel<double,%d>",
                    "mixed": "fb_kernel<float,%d> + fb_kernel<double,%d> (per-band precision)"}[fx.fb.precision]
        launches = launches.replace("%d", "1" if T <= 512 else "2")
    else:
        stem = "fb_rows4_kernel" if xs.shape[-1] % 512 == 0 else "fb_long_kernel"      # csrc/fb.hip fb_launch_t
        launches = {"f32": f"{stem}_f32", "f64": f"{stem}_f64",
                    "mixed": f"{stem}_f32 + {stem}_f64 (per-band precision)"}[fx.fb.precision]
    del y
    return {"bound": "hbm", "kernel": launches, "achieved": round(by / (ms * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": None, "ms_per_pass": round(ms, 4), "ms_per_pass_min": round(ms_min, 4),
            "frac_best": round(by / (ms_min * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "passes": f"{len(times)} groups x {per_group} back-to-back passes, one HIP event pair per group; median group",
            "shader_clock_mhz": clk, "trials_per_pass": Bs, "algorithmic_bytes_per_pass": by}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="trials per GPU (default: the configuration's batch)")
    ap.add_argument("--two-kernel", action="store_true", help="materialise the filtered signals (fb + bandpower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-roofline", action="store_true", help="skip the filterbank-stage HBM measurement")
    ap.add_argument("--bf16", action="store_true", help="config 3: bf16 activations/grads in the CNN, fp32 accumulate")
    ap.add_argument("--no-also", action="store_true",
                    help="default invocation only: skip the BASELINE config 3 (bf16) and config 5 (stress) runs that are "
                         "reported under \"also\" beside the cfg2 fp32 line")
    ap.add_argument("--overlap", action="store_true",
                    help="extract the features of the next batch on a second HIP stream while the CNN trains on the "
                         "current one, the CNN stream at high priority (measured, round 3: 1.32 ms against 1.26 ms "
                         "per step on one stream -- the fused extractor already fills every wave slot and the "
                         "CNN's short kernels queue behind its waves -- so the default is one stream)")
    ap.add_argument("--preroll-ms", type=float, default=100.0,
                    help="time-based GPU pre-roll of the same step ahead of the warm-up steps (0: none)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test (tests/test_bench_launch_cpu.py): no HIP, a CPU tensor through the same "
                         "rank bring-up, GradientBucket all-reduce, barriers, MAX-over-ranks timing and one JSON line")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="--stub only: this rank raises before the timed loop")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N > 1 only: wait for the gradient all-reduce before extracting the next batch's features "
                         "(default: the all-reduce of step k runs under the feature extraction of batch k+1)")
    return ap.parse_args()


def run_workload(config, steps, warmup, batch, bf16, two_kernel, overlap_flag, no_pipeline, hbm_roofline, env,
                 preroll_ms=100.0):
    """One workload: a time-based pre-roll of the same step, W untimed warm-up steps, exactly K timed steps between
    barrier + synchronize pairs, the MAX of the ranks' wall times.  Returns the JSON line's fields on rank 0 (None
    elsewhere)."""
    import torch.distributed as dist
    import isd_amd
    from isd_amd.classifier import _EEGNetFeatureModel, _FeatureModel
    rank, world, dev, backend = env["rank"], env["world"], env["dev"], env["backend"]
    cfg = CONFIGS[config]
    B, C, T, fs = batch, cfg["C"], cfg["T"], cfg["fs"]
    bands = getattr(isd_amd, cfg["bands"])
    nb = len(bands)
    Xh, yh = synth_trials(B, C, T, fs, seed=rank)               # rank r uses default_rng(r)
    x = torch.from_numpy(Xh).to(dev)
    y = torch.from_numpy(yh).to(dev)
    del Xh

    torch.manual_seed(42)                                        # reference default seed (train_fast.py:275)
    fx = isd_amd.FeatureExtractor(T, fs, bands, nperseg=cfg["nperseg"], noverlap=cfg["noverlap"])
    if config == "cfg2":
        model = _FeatureModel(nb * C, 32, 5, 4, "bf16" if bf16 else "f32").to(dev)
    else:
        model = _EEGNetFeatureModel(nb * C, 32, 5, kernel_length=64, dropout=0.25).to(dev)
    trainer = isd_amd.Trainer(model, lr=5e-4, weight_decay=1e-2, schedule=None)
    # config 3: the feature map itself is bf16 (the rounding the bf16 first layer applies to an fp32 map anyway)
    feats = torch.empty((B, nb, C, fx.n_frames), dtype=torch.bfloat16 if (bf16 and not two_kernel) else torch.float32,
                        device=dev)
    yfilt = torch.empty((B, nb, C, T), dtype=torch.float32, device=dev) if two_kernel else None
    global_batch = B * world
    fused = not two_kernel

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
    overlap = fused and overlap_flag
    # --overlap: the CNN step runs on a HIGH-priority stream (its kernels are short and latency-bound: they take the
    # wave slots that free up first), the extractor on a normal one
    main_stream = torch.cuda.Stream(priority=-1) if overlap else torch.cuda.current_stream()
    feat_stream = torch.cuda.Stream() if overlap else main_stream
    if overlap:
        main_stream.wait_stream(torch.cuda.current_stream())
        feat_stream.wait_stream(torch.cuda.current_stream())
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
        with torch.cuda.stream(main_stream):
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
    pipelined = world > 1 and not overlap and not no_pipeline

    def pipelined_step(e=None):
        out = trainer.step_begin(feats.view(B, nb * C, fx.n_frames), y, global_batch=global_batch)
        extract(0, e)                                       # same stream: queued behind the backward pass that read feats
        # e[1] (end of the extraction) -> e[4] (the stream got past its wait for the collective) = the part of the
        # all-reduce that did NOT hide under the extraction
        trainer.step_finish(after_wait=e[4] if e else None)
        if e:
            e[2].record(main_stream)
        return out

    # Every step = one feature-extraction pass + one CNN fwd/bwd/optimizer pass.  With overlap the extraction
    # of the NEXT batch runs on its own stream under the CNN work of the current one (the features do not
    # depend on the parameters), exactly K of each inside the timed region.
    freed[0].record(main_stream); freed[1].record(main_stream)
    if overlap or pipelined:
        extract(0)
    it = 0                                                  # steps issued so far (the overlap mode's buffer parity)

    def one_step(e=None):
        nonlocal it
        if pipelined:
            out = pipelined_step(e)
        elif overlap:
            extract((it + 1) % 2, e)
            out = train(it % 2, e)
        else:
            extract(0, e)
            out = train(0, e)
        it += 1
        return out

    def over_ranks(v, op):
        if world == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=op)
        return float(tt.item())

    # Pre-roll: the same step, untimed, until >= preroll_ms of wall time have passed on EVERY rank (the ranks decide
    # on the all-reduced minimum, so they run the same number of steps and collectives).
    pre_steps, pre_ms = 0, 0.0
    if preroll_ms > 0:
        chunk = 1 if config == "cfg5" else 8
        tp = time.perf_counter()
        while pre_ms < preroll_ms and pre_steps < 4096:
            for _ in range(chunk):
                one_step()
            pre_steps += chunk
            torch.cuda.synchronize()
            pre_ms = over_ranks((time.perf_counter() - tp) * 1e3, dist.ReduceOp.MIN)
    for i in range(warmup):
        one_step()
    clk_out, clk_fire = clock_probe_start()
    host_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        h0 = time.perf_counter()
        out = one_step(ev[i])
        if i == steps // 2:
            clk_fire()                                      # shader clock WHILE the timed steps run (no host sync)
        host_ms.append((time.perf_counter() - h0) * 1e3)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = over_ranks(dt, dist.ReduceOp.MAX)
    clk_timed = clock_probe_read(clk_out)
    loss = float(out["loss"])
    if rank != 0:
        return None

    ms = dt / steps * 1e3
    feat_i = [e[0].elapsed_time(e[1]) for e in ev]
    train_i = [e[1].elapsed_time(e[2]) for e in ev] if not (overlap or pipelined) else []
    t_feat = float(np.mean(feat_i))
    t_train = float(np.mean(train_i)) if train_i else float("nan")
    stage_spread = {"extract_features": spread(feat_i), "cnn_fwd_bwd_allreduce_adamw": spread(train_i),
                    "step": spread([ev[i][0].elapsed_time(ev[i + 1][0]) for i in range(steps - 1)]),
                    "host_issue": spread(host_ms)}
    if pipelined:
        # the exposed part of the all-reduce (0 when it hid under the extraction), the forward/backward between the
        # AdamW of step i-1 and the extraction of step i, and AdamW behind the wait
        stage_spread["allreduce_exposed"] = spread([e[1].elapsed_time(e[4]) for e in ev])
        stage_spread["adamw_after_wait"] = spread([e[4].elapsed_time(e[2]) for e in ev])
        stage_spread["cnn_fwd_bwd"] = spread([ev[i][2].elapsed_time(ev[i + 1][0]) for i in range(steps - 1)])
    n_sec, bins = 4, [hi - lo + 1 for lo, hi in fx.bins]
    roof_hbm = None
    if fused and config == "cfg2":
        # dominant kernel: the fused filterbank+STFT extractor, fp32-VALU-bound on its algorithmic traffic, so the
        # compute roofline is the honest one.  Algorithmic flops per (trial, channel): cascade = nb bands x T
        # samples x (4 sections x 9 flop [3-op recursion + two state fix-up FMAs] + 1 gain multiply); band DFT =
        # 64 windowed samples per (frame, in-band bin), one complex MAC by a real sample each (4 flop).
        flops = B * C * (nb * T * (n_sec * 9 + 1) + sum(bins) * fx.n_frames * 64 * 4)
        from isd_amd import _lib as _l
        serial = int(_l.lib().isd_features_fused_last_path()) == 2      # which extractor family the timed steps launched
        kname = ("fused_serial_kernel<3,false> (one row per lane; fp32 vector ALU; no MFMA in this kernel)" if serial
                 else "fused_kernel<float> (fp32 vector ALU; no MFMA in this kernel)")
        roof = {"bound": "valu", "kernel": kname,
                "achieved": round(flops / (t_feat * 1e-3) / 1e12, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / (t_feat * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4), "traffic": None,
                "ms_per_launch": round(t_feat, 4),
                "algorithmic_bytes_per_launch": B * C * 4 * (T + nb * fx.n_frames)}
    elif fused:
        # cfg5: fused_rows4_kernel<float,5> for the 35 bands the AUTO rule keeps in fp32 + fused_rows4_kernel<double,5>
        # for the five lowest 2-Hz bands (4 - 14 Hz at 1024 Hz), one launch each per extraction.  VALU-bound.
        # Algorithmic flops per (trial, channel, band): cascade T x (4 x 9 + 1) + half-block DFT sums of the band's
        # bins and their two Hann neighbours: (bins + 2) x T complex MACs x 4.
        flops = B * C * sum(T * (n_sec * 9 + 1) + (nbin + 2) * T * 4 for nbin in bins)
        kname = {"f32": "fused_rows4_kernel<float,5> (all bands in fp32; one launch per extraction)",
                 "f64": "fused_rows4_kernel<double,5> (priced against the fp32 vector peak)",
                 "mixed": "fused_rows4_kernel<float,5>+<double,5> (one launch each per extraction; priced against "
                          "the fp32 vector peak)"}[fx.fb.precision]
        roof = {"bound": "valu", "kernel": kname,
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
    if fused and world == 1 and hbm_roofline:
        del feats, fbuf                                       # room for the filtered tensor
        roof_hbm = filterbank_hbm_roofline(fx, x, nb)
    tf = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tf):
        try:
            table = json.load(open(tf))
            def measured(ent, trials):
                # PMC bytes of the pass the table was measured on; every trial is independent, so an entry may give
                # bytes per trial instead (measured on a smaller batch of the same shape)
                if not ent:
                    return None
                if "bytes_per_trial" in ent:
                    return int(round(ent["bytes_per_trial"] * trials))
                return ent["bytes_per_launch"] if trials == ent.get("trials_per_launch", 4096) else None
            roof["traffic"] = measured(table.get(roof["kernel"].split(" ")[0]), B)
            if roof_hbm:
                roof_hbm["traffic"] = measured(table.get(roof_hbm["kernel"].split(" ")[0]), roof_hbm["trials_per_pass"])
        except Exception:
            pass
    line = {
        "metric": "trials/sec end-to-end (filterbank+CNN fwd+bwd)", "value": round(global_batch * steps / dt, 1),
        "unit": "trials/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 feature map, activations and activation gradients in the CNN; f32 extraction, parameters and "
                 "accumulation" if bf16 else "f32",
        "data": "synthetic",
        "config": {"workload": cfg["workload"],
                   "trials_per_gpu": B, "global_batch": global_batch, "parallelism": f"dp{world}",
                   "feature_path": "fused" if fused else "filterbank+bandpower kernels",
                   "streams": "features of batch k+1 overlap the CNN step of batch k" if overlap else
                              "one stream; gradient all-reduce of step k (RCCL stream) under the feature "
                              "extraction of batch k+1" if pipelined else "one stream"},
        "stages_ms": {"extract_features": round(t_feat, 4),
                      "cnn_fwd_bwd_allreduce_adamw": None if (overlap or pipelined) else round(t_train, 4)},
        "stages_ms_spread": stage_spread,
        "shader_clock_mhz_timed": clk_timed,
        "preroll_ms": round(pre_ms, 1), "preroll_steps": pre_steps,
        "final_loss": round(loss, 5),
        "roofline": roof,
    }
    if pipelined:
        line["allreduce_exposed_ms"] = stage_spread["allreduce_exposed"]["median"]
    if roof_hbm:
        line["roofline_hbm"] = roof_hbm
    return line


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes through
    torch.distributed.run and exit with their code.  Called before anything in this process touches the GPU (no HIP
    call, no exec): rank 0's JSON line goes straight to the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ISD_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "1")                       # what torchrun would set (with a warning) anyway
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def run_stub(args, env):
    """Launcher self-test: the rank bring-up, GradientBucket's flat all-reduce (start / wait), the barrier-bracketed
    timed region and the MAX-over-ranks reduction on CPU tensors -- everything of the N > 1 path except the kernels."""
    import torch.distributed as dist
    from isd_amd.classifier import GradientBucket
    rank, world = env["rank"], env["world"]
    bucket = GradientBucket()
    grad = torch.empty(157381)                                   # the train_head gradient block's size
    if rank == args.stub_fail_rank:
        raise RuntimeError(f"--stub-fail-rank: rank {rank} fails on purpose")
    for _ in range(args.warmup):
        bucket.all_reduce_wait(bucket.all_reduce_start(grad.fill_(rank + 1.0)))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bucket.all_reduce_wait(bucket.all_reduce_start(grad.fill_(rank + 1.0)))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    want = world * (world + 1) / 2.0
    if float(grad[0]) != want or float(grad[-1]) != want:
        raise RuntimeError(f"stub all-reduce: got {float(grad[0])}, want {want}")
    if rank != 0:
        return None
    return {"metric": "stub steps/sec (launcher self-test; no GPU work)", "value": round(args.steps / dt, 1),
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stub: flat-gradient all-reduce of 157381 floats on CPU tensors",
                       "parallelism": f"dp{world}", "backend": env["backend"],
                       "self_launched": bool(os.environ.get("ISD_BENCH_SELF_LAUNCHED"))}}


def main():
    args = parse_args()
    cfg = CONFIGS[args.config]
    # the headline invocation (any --steps / --warmup): cfg2 fp32, fused, one stream, the configuration's own batch
    default_call = args.config == "cfg2" and args.batch is None and not (args.bf16 or args.two_kernel or args.overlap)
    args.steps = cfg["steps"] if args.steps is None else args.steps
    args.warmup = cfg["warmup"] if args.warmup is None else args.warmup
    args.batch = cfg["batch"] if args.batch is None else args.batch
    if args.config == "cfg5" and (args.bf16 or args.two_kernel):
        raise SystemExit("--bf16 / --two-kernel apply to cfg2")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has not touched the GPU)
        raise SystemExit(self_launch(args.gpus))

    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    # ISD_DIST_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks share a card)
    backend = os.environ.get("ISD_DIST_BACKEND", "gloo" if args.stub else "nccl")
    if args.stub:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend, rank=rank, world_size=world)
        line = run_stub(args, dict(rank=rank, world=world, backend=backend))
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # RCCL's stream at HIGH priority: its few workgroups take the first wave slots the extractor's 65 536
            # waves free instead of queueing behind them (the collective runs under the next batch's extraction)
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=opts)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    env = dict(rank=rank, world=world, dev=dev, backend=backend)

    line = run_workload(args.config, args.steps, args.warmup, args.batch, args.bf16, args.two_kernel, args.overlap,
                        args.no_pipeline, not args.no_hbm_roofline, env, preroll_ms=args.preroll_ms)
    # The driver's plain `python bench.py` also times BASELINE config 3 (the same workload with bf16 activations and
    # gradients) and config 5 (the stress configuration at its own batch of 2048) after the cfg2 fp32 region: `value`
    # stays the cfg2 fp32 number, the other two ride along under "also" (VERDICT r2, item 3).  Each leg has its own
    # pre-roll; the host-only CPU baseline (14 s of scipy, the GPU idle) comes LAST so that no leg starts cold.
    if world == 1 and (default_call or os.environ.get("ISD_BENCH_ALSO")) and not args.no_also:
        also = {}
        keep = ("value", "ms_per_step", "stages_ms", "stages_ms_spread", "shader_clock_mhz_timed", "preroll_ms",
                "preroll_steps", "dtype", "steps", "warmup", "final_loss")
        torch.cuda.empty_cache()
        l3 = run_workload("cfg2", 10, 3, CONFIGS["cfg2"]["batch"], True, False, False, False, False, env,
                          preroll_ms=args.preroll_ms)
        also["cfg3"] = {k: l3[k] for k in keep}
        also["cfg3"]["workload"] = "cfg2's workload with a bf16 feature map and bf16 activations / activation gradients " \
                                   "in the CNN (bf16 MFMA); f32 extraction arithmetic, parameters and accumulation"
        torch.cuda.empty_cache()
        l5 = run_workload("cfg5", 3, 1, CONFIGS["cfg5"]["batch"], False, False, False, False, True, env,
                          preroll_ms=args.preroll_ms)
        also["cfg5"] = {k: l5[k] for k in keep + ("roofline", "roofline_hbm") if k in l5}
        also["cfg5"]["workload"] = CONFIGS["cfg5"]["workload"]
        line["also"] = also
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.config, cfg)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
