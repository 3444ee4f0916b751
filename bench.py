#!/usr/bin/env python3
"""Benchmark of the Rater hot path on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W

N = 1: runs in-process.  N > 1: the driver launches this file under
`python -m torch.distributed.run --nproc-per-node N ...`; every rank owns one GPU
and B independent stateful streams; gradients are averaged with one RCCL all-reduce
per step (weak scaling: global batch = B*N).

A "step" = one training batch of BASELINE.json config 1 ("cfg2"): depth=2 width=512
length=256, V=256, one context variable, B streams of T=256 characters:
forward + backward + gradient all-reduce + clip/Adam, bf16 MFMA operands with f32
accumulation and f32 master weights, dropout on.  Inputs (the 10M-character
synthetic corpus) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      the dominant kernel (the persistent backward scan of one layer) against the
                dense bf16 MFMA peak: algorithmic FLOPs per launch / mean launch time,
                timed with HIP events on the engine's stream in an extra traced step.
                `traffic` (HBM bytes per launch) is NOT measured by this run: it is read from the
                committed rocprofv3 --pmc summary of the same command and stream count named in
                `traffic_source`, and is null when no such file matches
  cpu_baseline  the CPU restatement (oracle/, numpy f32) of the same training step
                with the reference's own batching (1 stream x 256 chars, stateful),
                timed on this box's host cores on a bounded sample
  cpu_baseline_torch  informative second CPU figure: the same step with torch.nn.LSTM (oneDNN) + autograd
  incremental   hypotheses*chars/s of the batched incremental step, split-bf16 precision (the rating
                precision): cfg3 (1024 hypotheses x 512 chars) and the reference's own batch cap
                (128 hypotheses, rating.py:49, 809); wall and GPU-only time per step, algorithmic
                HBM bytes per step; under N > 1 every rank runs it and the values are summed
  rating_window chars/s and ms per stateful 256-char window of the rating forward (1 and 64 streams, split precision)
  end_to_end    chars/s of Rater.train itself (file reading, window generation, vocabulary look-up,
                dropout masks, loss read-backs, one validation pass) over synthetic text files of the
                same topology and stream count, rank 0 only -- what the host side costs
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEPTH, WIDTH, LENGTH, VOC, N_CTX = 2, 512, 256, 256, 1
CORPUS = 10_000_000
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def flops_fwd_per_char(depth=DEPTH, width=WIDTH, voc=VOC, n_ctx=N_CTX):
    """F_fwd = sum_l 2 (D_l + W) 4W + 2 W V"""
    f = 0
    for l in range(depth):
        d = width + 10 * n_ctx if l == 0 else width
        f += 2 * (d + width) * 4 * width
    return f + 2 * width * voc


def flops_cell_per_char(depth=DEPTH, width=WIDTH, n_ctx=N_CTX):
    """the LSTM contractions one cell-step launch stands for (no output projection)"""
    return flops_fwd_per_char(depth, width, 0, n_ctx)


def synthetic_corpus(n=CORPUS, voc=VOC, seed=0):
    rng = np.random.default_rng(seed)
    p = 1.0 / (np.arange(1, voc) + 1.0)
    p /= p.sum()
    ids = rng.choice(np.arange(1, voc), size=n, p=p).astype(np.int32)
    ids[rng.random(n) < 0.01] = 0
    return ids


def cpu_baseline(seconds=15.0):
    """CPU restatement (numpy oracle, f32) of the same training step with the
    reference's batching: stateful, batch 1 x length 256 (rating.py:90-92)."""
    from oracle import lstm_oracle as O
    # a 1 x 256 stateful window is a chain of matrix-vector products: BLAS gains nothing
    # beyond a few threads (128 threads measured 2.5x SLOWER than 8), so cap them
    threads = min(16, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=threads)
    except Exception:
        pass
    cfg = O.ModelConfig(DEPTH, WIDTH, VOC, N_CTX)
    w = O.init_weights(cfg, seed=1, dtype=np.float32)
    opt = O.Adam(cfg, dtype=np.float32)
    rng = np.random.default_rng(0)
    ids = synthetic_corpus(64 * LENGTH + 1, VOC, 0)
    states = O.zero_states(cfg, 1, np.float32)
    ctx = np.full((1, LENGTH, 1), 17)
    n = 0
    t0 = time.time()
    while True:
        k = n % 64
        idx = ids[k * LENGTH:(k + 1) * LENGTH][None].astype(np.int64)
        tgt = ids[k * LENGTH + 1:(k + 1) * LENGTH + 1][None].astype(np.int64)
        masks = O.draw_dropout_masks(cfg, 1, rng)
        _, _, states = O.train_step(cfg, w, opt, idx, ctx, tgt, states, masks)
        n += 1
        el = time.time() - t0
        if el >= seconds or n >= 400:
            break
    return {"value": n * LENGTH / el, "unit": "chars/s", "cores": int(threads), "kind": "port",
            "sample": "%d stateful windows of 1x%d chars (forward+backward+Adam, numpy f32 oracle, %.1f s)" % (n, LENGTH, el)}


def cpu_baseline_torch(seconds=8.0):
    """Second CPU figure beside the oracle's ("optimised CPU stand-in"): the same
    1 x 256 stateful training step written with torch.nn.LSTM / oneDNN + autograd + Adam(clipvalue 1)."""
    import torch
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    W, V, T = WIDTH, VOC, LENGTH
    emb, ctx = torch.nn.Embedding(V, W), torch.nn.Embedding(200, 10)
    layers = [torch.nn.LSTM(W + 10 * N_CTX if l == 0 else W, W, batch_first=True) for l in range(DEPTH)]
    params = list(emb.parameters()) + list(ctx.parameters()) + [p for m in layers for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, eps=1e-7)
    ids = torch.from_numpy(synthetic_corpus(64 * T + 1, V, 0).astype(np.int64))
    c = torch.full((1, T), 17)
    states = [None] * DEPTH
    n, t0 = 0, time.time()
    while True:
        k = n % 64
        x, y = ids[k * T:(k + 1) * T][None], ids[k * T + 1:(k + 1) * T + 1][None]
        h = torch.cat([emb(x), ctx(c)], -1)
        for l, m in enumerate(layers):
            h, st = m(h, states[l])
            states[l] = tuple(s.detach() for s in st)
            if l > 0:
                h = torch.nn.functional.dropout(h, 0.1)
        loss = torch.nn.functional.cross_entropy((h @ emb.weight.t()).view(-1, V), y.view(-1))
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(params, 1.0)
        opt.step()
        n += 1
        el = time.time() - t0
        if el >= seconds or n >= 400:
            break
    return {"value": n * T / el, "unit": "chars/s", "cores": int(threads), "kind": "torch-cpu (oneDNN LSTM + autograd)",
            "sample": "%d stateful windows of 1x%d chars (%.1f s)" % (n, T, el)}


def end_to_end_leg(B, windows_per_file=201):
    """Rater.train (the drop-in API) over B synthetic text files of windows_per_file x 256 characters each (one
    stateful stream per file), cfg2 topology, ONE epoch incl. its validation pass; chars/s = the characters the
    training steps consumed / the wall time of the whole train() call (reading, splitting and encoding the files,
    the vocabulary scan and the validation pass included)."""
    import logging
    import tempfile
    from ocrd_keraslm_amd.lib import Rater
    rng = np.random.default_rng(5)
    p = 1.0 / (np.arange(1, VOC) + 1.0)
    p /= p.sum()
    size = windows_per_file * LENGTH + 1
    n_val = max(1, B // 8)
    with tempfile.TemporaryDirectory() as tmp:
        names = []
        for i in range(B + n_val):
            ids = rng.choice(VOC - 1, size=size, p=p)
            name = os.path.join(tmp, "a_b%d_%d.txt" % (i, 1700 + (i % 29) * 10))
            with open(name, "w", encoding="utf-8") as f:      # (code points U+0100 ..: one character per id)
                f.write((ids.astype('<u4') + 0x100).tobytes().decode('utf-32-le'))
            names.append(name)
        r = Rater(logger=logging.getLogger("bench.e2e"))
        r.width, r.depth, r.length = WIDTH, DEPTH, LENGTH
        r.stateful = True
        r.streams = B
        r.max_epochs = 1
        r.seed = 1
        r.configure()
        files = [open(n, encoding="utf-8") for n in names[:B]]
        val = [open(n, encoding="utf-8") for n in names[B:]]
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            t0 = time.perf_counter()
            r.train(files, val_data=val)
            el = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
            for f in files + val:
                f.close()
        steps = windows_per_file - 1          # (rating.py:342: ceil((size - length) / length) full windows per file)
        chars = steps * B * LENGTH
        t = {k: round(v, 3) for k, v in getattr(r, "timings", {}).items()}
        steady = chars / t["train_steps"] if t.get("train_steps") else None
        return {"value": chars / el, "unit": "chars/s", "seconds": el, "train_steps": steps, "streams": B,
                "phases_s": t, "train_steps_only": steady,
                "note": "Rater.train, one epoch + validation, wall time of the whole call; phases_s splits it into reading + "
                        "splitting the files, mapping them to ids, the training steps, validation, checkpoint; "
                        "train_steps_only = chars / the training-step phase (what further epochs cost)"}


def training_leg(device, depth, width, length, n_ctx, B, steps, warmup, corpus, seed=1):
    """cfg-style stateful training on ONE GPU (no collective): B streams x `length` characters per step, forward + backward +
    clip/Adam, dropout on, inputs resident in HBM.  Returns chars/s, ms per step and the fraction of the bf16 MFMA roof
    over the whole step (3 F_fwd per character)."""
    import torch
    from ocrd_keraslm_amd.lib import hipabi
    from ocrd_keraslm_amd.lib.engine import HipLM
    lm = HipLM(depth, width, VOC, n_ctx, device=device)
    lm.init_weights(seed=seed)
    lm.prepare(hipabi.KL_PREC_BF16)
    lm.ensure_training_buffers()
    T = length
    per = len(corpus) // B
    mine = np.stack([corpus[s * per:(s + 1) * per] for s in range(B)])
    streams = torch.from_numpy(mine).to(device)
    rng = np.random.default_rng(7)
    ctx_ids = torch.from_numpy(rng.integers(0, 200, size=(B, n_ctx)).astype(np.int32)).to(device)
    ctx = ctx_ids[:, None, :].expand(B, T, n_ctx).contiguous()
    n_windows = (per - 1) // T
    gen = torch.Generator(device=device)
    gen.manual_seed(2)
    lm.reset_states(B)

    def step(w):
        w = w % n_windows
        idx = streams[:, w * T:(w + 1) * T].contiguous()
        tgt = streams[:, w * T + 1:(w + 1) * T + 1].contiguous()
        keep = torch.rand((depth, B, width), device=device, generator=gen) >= 0.1
        lm.train_window(idx, ctx, tgt, keep.to(torch.float32) / 0.9)
        lm.adam_step()

    for w in range(warmup):
        step(w)
    lm.read_loss()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lm.read_loss()      # (raises if a persistent scan's hand-off timed out)
    value = steps * B * T / el
    out = {"value": value, "unit": "chars/s", "streams": B, "seq_len": T, "steps": steps, "warmup": warmup,
           "ms_per_step": el / steps * 1e3,
           "mfma_frac": value * 3 * flops_fwd_per_char(depth, width, VOC, n_ctx) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
    return out, lm


def incremental_leg(lm, device, depth, width, n_ctx, N, S, seed=3):
    """N hypotheses x S chained incremental steps through the state pool (split precision), the first fifth untimed."""
    import torch
    lm.ensure_pool(2 * N)
    r3 = np.random.default_rng(seed)
    ids = torch.from_numpy(r3.integers(1, VOC, size=(S, N)).astype(np.int32)).to(device)
    cc = torch.from_numpy(r3.integers(0, 200, size=(N, n_ctx)).astype(np.int32)).to(device)
    a = torch.arange(N, dtype=torch.int32, device=device)
    b = a + N
    lm.pool.zero_()
    warm = S // 5
    for s in range(warm):
        lm.step_slots(ids[s], cc, a, b)
        a, b = b, a
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    e0.record()
    for s in range(warm, S):
        lm.step_slots(ids[s], cc, a, b)
        a, b = b, a
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t1
    gpu_us = e0.elapsed_time(e1) * 1e3 / (S - warm)
    # algorithmic HBM bytes per step: states read + written, probabilities, indices; the bf16
    # hi+lo weights of all layers are read once per step (L2/MALL-resident between steps or not)
    state_bytes = N * (2 * (2 * depth * width * 4) + VOC * 4 + 4 * (1 + n_ctx))
    weight_bytes = sum(((width + 10 * n_ctx if l == 0 else width) + width) * 4 * width * 2 for l in range(depth)) * 2
    return {"value": N * (S - warm) / el, "hypotheses": N, "chars": S, "us_per_step": el / (S - warm) * 1e6, "gpu_us_per_step": gpu_us,
            "algorithmic_bytes_per_step": state_bytes + weight_bytes,
            "hbm_frac": (state_bytes + weight_bytes) / (gpu_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "mfma_frac": N * flops_fwd_per_char(depth, width, VOC, n_ctx) / (gpu_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("KL_BENCH_STREAMS", "3072")),
                    help="stateful streams per GPU (B)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-incremental", action="store_true")
    ap.add_argument("--no-extra-shapes", action="store_true", help="skip the small_batch (cfg2 at 1 and 64 streams) and cfg5 blocks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ocrd_keraslm_amd.lib import hipabi
    from ocrd_keraslm_amd.lib.distributed import GradSync, init_from_env
    from ocrd_keraslm_amd.lib.engine import HipLM

    # rehearsal on a one-GPU box: KL_BENCH_SAME_GPU=1 maps every rank to cuda:0 and uses gloo
    same_gpu = os.environ.get("KL_BENCH_SAME_GPU") == "1"
    if same_gpu:
        os.environ["LOCAL_RANK"] = "0"
    backend = "gloo" if same_gpu else ("nccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None)
    rank, world, local = init_from_env(backend)
    if world != args.gpus and world > 1:
        args.gpus = world
    device = "cuda:%d" % local
    torch.cuda.set_device(local)
    B, T = args.streams, LENGTH

    lm = HipLM(DEPTH, WIDTH, VOC, N_CTX, device=device)
    lm.init_weights(seed=1)                       # identical initial weights on every rank
    lm.prepare(hipabi.KL_PREC_BF16)
    lm.ensure_training_buffers()
    sync = GradSync()

    # synthetic corpus cut into B*world contiguous streams; this rank's share lives in HBM
    corpus = synthetic_corpus()
    n_streams = B * world
    per = CORPUS // n_streams
    mine = np.stack([corpus[(rank * B + s) * per:(rank * B + s + 1) * per] for s in range(B)])
    streams = torch.from_numpy(mine).to(device)
    rng = np.random.default_rng(7)
    ctx_ids = torch.from_numpy(rng.integers(0, 200, size=n_streams)[rank * B:(rank + 1) * B].astype(np.int32)).to(device)
    ctx = ctx_ids[:, None, None].expand(B, T, 1).contiguous()
    n_windows = (per - 1) // T
    gen = torch.Generator(device=device)
    gen.manual_seed(2 + rank)
    lm.reset_states(B)

    def step(w, all_reduce=True):
        w = w % n_windows
        idx = streams[:, w * T:(w + 1) * T].contiguous()
        tgt = streams[:, w * T + 1:(w + 1) * T + 1].contiguous()
        keep = torch.rand((DEPTH, B, WIDTH), device=device, generator=gen) >= 0.1
        masks = keep.to(torch.float32) / 0.9
        lm.train_window(idx, ctx, tgt, masks)
        lm.adam_step(grad_scale=sync.reduce(lm) if all_reduce else 1.0)

    for w in range(args.warmup):
        step(w)
    lm.read_loss()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ce, acc, reg = lm.read_loss()
    chars = args.steps * B * T * world
    value = chars / elapsed

    # ---- roofline leg: one traced step, HIP events around the cell-step launches
    import ctypes as C
    roofline = None
    if rank == 0:
        hipabi.check(lm.lib.kl_trace_enable(lm.handle, 1))
        # (five traced steps, every layer's scan launch bracketed: 10 launches per direction at depth 2 -- the figure below is
        # their AVERAGE, comparable with rocprofv3's per-kernel average over the same command)
        for k in range(5):
            step(args.warmup + args.steps + k, all_reduce=False)    # rank 0 only: no collective in this leg
        torch.cuda.synchronize()
        out = {}
        for kind in (0, 1):
            n, ms, pers, fl = C.c_int(), C.c_float(), C.c_int(), C.c_double()
            hipabi.check(lm.lib.kl_trace_read(lm.handle, kind, C.byref(n), C.byref(ms), C.byref(pers), C.byref(fl)))
            kname = lm.lib.kl_trace_kernel_name(lm.handle, kind).decode()     # the kernel the library actually ran
            out[kname] = (n.value, ms.value, bool(pers.value), fl.value)
        hipabi.check(lm.lib.kl_trace_enable(lm.handle, 0))
        # the recurrence kernel with the longer launch (the timed launches are one layer's scan in the
        # layer-sequential mode, all layers' in the fused mode; both directions use the same mode)
        name = max(out, key=lambda k: out[k][1] / max(out[k][0], 1))
        n, ms, pers, fl = out[name]
        per_launch_s = ms / max(n, 1) / 1e3
        # algorithmic FLOPs of one launch: a persistent scan launch reports the LSTM contractions it
        # carries for its B*T chars (the library knows whether layers are fused or scanned one by one);
        # a step launch carries every layer's cell for B chars (SURVEY.md 8d per-char figure)
        flops_launch = fl if pers else B * flops_cell_per_char()
        achieved = flops_launch / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
        # HBM bytes per launch: NOT measured here (counters need their own rocprofv3 --pmc passes, tools/profile_round.sh);
        # taken from the committed summary of the same command at the same stream count, and labelled as such
        traffic, traffic_source = None, None
        for tag in ("r04", "r03", "r02", "r01"):
            fn = os.path.join("profiles", "%s_pmc_hbm_traffic_B%d.json" % (tag, B))
            try:
                pmc = json.load(open(os.path.join(ROOT, fn)))
                for k, v in pmc["kernels"].items():
                    if k.startswith(name):
                        traffic, traffic_source = v["hbm_bytes_per_launch"], fn + " (committed rocprofv3 --pmc summary, not this run)"
            except Exception:
                pass
            if traffic is not None:
                break
        roofline = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                    "launch_us": per_launch_s * 1e6, "launches_timed": n, "persistent": pers,
                    "flops_per_launch": flops_launch,
                    "other_kernel": {k: {"launches": v[0], "total_ms": v[1], "flops_per_launch": v[3]}
                                     for k, v in out.items() if k != name},
                    "whole_step_frac": value * 3 * flops_fwd_per_char() / world / 1e12 / MFMA_BF16_PEAK_TFLOPS}

    # ---- incremental rescoring: cfg3 (1024 hypotheses x 512 chars) and the reference's batch cap (128 hypotheses),
    # split precision; every rank runs it on its own GPU (independent hypothesis sets, no collective), values summed
    incremental = None
    if not args.no_incremental:
        lm.prepare(hipabi.KL_PREC_SPLIT)
        legs = {}
        for N, S in ((1024, 512), (128, 512), (32, 512)):
            # (a leg lasts 10-20 ms: measured five times -- the first may still see the clocks ramp up --, the MEDIAN reported)
            runs = sorted((incremental_leg(lm, device, DEPTH, WIDTH, N_CTX, N, S, seed=3 + rank) for _ in range(5)),
                          key=lambda r: r["us_per_step"])
            legs[N] = dict(runs[len(runs) // 2], median_of=len(runs), fastest_us_per_step=runs[0]["us_per_step"],
                           slowest_us_per_step=runs[-1]["us_per_step"])
        if world > 1:
            t = torch.tensor([legs[1024]["value"], legs[128]["value"]], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            legs[1024]["value"], legs[128]["value"] = float(t[0].item()), float(t[1].item())
        incremental = {"value": legs[1024]["value"], "unit": "hypotheses*chars/s", "hypotheses": 1024, "chars": 512,
                       "precision": "split-bf16 (3 MFMA passes)", "n_gpus": world, "median_of": 5,
                       "fastest_us_per_step": legs[1024]["fastest_us_per_step"], "slowest_us_per_step": legs[1024]["slowest_us_per_step"],
                       "us_per_step": legs[1024]["us_per_step"], "gpu_us_per_step": legs[1024]["gpu_us_per_step"],
                       "algorithmic_bytes_per_step": legs[1024]["algorithmic_bytes_per_step"],
                       "hbm_frac": legs[1024]["hbm_frac"], "mfma_frac": legs[1024]["mfma_frac"],
                       "n128": {"value": legs[128]["value"], "hypotheses": 128, "us_per_step": legs[128]["us_per_step"],
                                "gpu_us_per_step": legs[128]["gpu_us_per_step"],
                                "algorithmic_bytes_per_step": legs[128]["algorithmic_bytes_per_step"],
                                "hbm_frac": legs[128]["hbm_frac"], "mfma_frac": legs[128]["mfma_frac"],
                                "note": "the reference's batch cap (rating.py:49, 809)"},
                       "n32": {"value": legs[32]["value"], "hypotheses": 32, "us_per_step": legs[32]["us_per_step"],
                               "gpu_us_per_step": legs[32]["gpu_us_per_step"],
                               "note": "a beam of 10 with 3 alternatives per edge (rate_best's default width, rating.py:712); this rank's GPU"}}

    # ---- the step as a beam search sees it (rating.py:809-826): one call per character, the GPU idle in between, wall time
    # from the call to the probabilities in host memory; and Rater.rate_best itself on a synthetic 600-edge page lattice
    beam = None
    if not args.no_incremental and rank == 0 and world == 1:
        try:
            import importlib.util
            def _tool(name):
                spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", name + ".py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return mod
            beam = _tool("probe_step_latency").run((30, 128), DEPTH, WIDTH, VOC, steps=400, skip=100)
            beam["rate_best_ms_per_edge"] = _tool("probe_rate_best").run(600, pages=3, clustering=0)
            beam["rate_best_clustering_ms_per_edge"] = _tool("probe_rate_best").run(600, pages=3, clustering=5)
            beam["note"] = ("median wall microseconds of ONE incremental step with the GPU idle before it: device_entry = index copy + "
                            "kl_step_batch + copy of the probabilities back (round 3's Rater._predict_refs); host_entry = kl_step_batch_host "
                            "(indices in the kernel arguments, delivery into host memory, arrival word) with whole rows / only the target "
                            "characters' probabilities / plus the head vectors history clustering compares")
        except Exception as err:      # (diagnostic block: never costs the metric line)
            beam = {"error": repr(err)}

    # ---- rating windows (rate / rate2 / test): the stateful windowed forward in split precision with probabilities out,
    # at the reference's batching (1 stream x 256 chars, rating.py:490) and at 64 streams; rank 0 reports its own GPU
    rating = None
    if not args.no_incremental:
        lm.prepare(hipabi.KL_PREC_SPLIT)
        rating = {}
        for Br in (1, 64):
            lm.reset_states(Br)
            r4 = np.random.default_rng(11)
            xi = torch.from_numpy(r4.integers(1, VOC, size=(Br, LENGTH)).astype(np.int32)).to(device)
            ci = torch.from_numpy(r4.integers(0, 200, size=(Br, 1, N_CTX)).repeat(LENGTH, axis=1).astype(np.int32)).to(device)
            for _ in range(5):
                lm.forward_window(xi, ci)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            t1 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                lm.forward_window(xi, ci)
            e1.record()
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            gpu_ms = e0.elapsed_time(e1) / reps
            rating["streams_%d" % Br] = {"value": Br * LENGTH * reps / el, "unit": "chars/s", "ms_per_window": el / reps * 1e3,
                                         "gpu_ms_per_window": gpu_ms,
                                         "mfma_frac": Br * LENGTH * flops_fwd_per_char() / (gpu_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        rating["precision"] = "split-bf16 (3 MFMA passes)"
        rating["note"] = "stateful window of 256 chars, probabilities copied to the host as Rater.rate needs them (16.8 MB per window at 64 streams); 1 stream = the reference's batching"

    # ---- the other shapes SURVEY.md 8(d) asks for, one GPU each (N = 1 runs only: no collective in them)
    small_batch = None
    cfg5 = None
    if rank == 0 and world == 1 and not args.no_extra_shapes:
        lm._ws = None                      # (the 3072-stream workspace goes back to the allocator first)
        lm._ws_key = None
        torch.cuda.empty_cache()
        # cfg2 at the reference's own batching (one stateful stream per step, rating.py:90-92) and at 64 streams
        small_batch = {}
        for Bs, st in ((1, 200), (64, 100)):
            try:
                leg, lms = training_leg(device, DEPTH, WIDTH, LENGTH, N_CTX, Bs, st, 10, corpus[:Bs * (CORPUS // 64)])
                del lms
                small_batch["streams_%d" % Bs] = leg
            except Exception as err:
                small_batch["streams_%d" % Bs] = {"error": repr(err)}
        small_batch["note"] = "cfg2 topology, B = 1 is the reference's batching (rating.py:90-92), B = 64 SURVEY.md 8(d)'s second point"
        torch.cuda.empty_cache()
        # cfg5: depth 4, width 1024, length 512, two context variables -- training at 512 streams, incremental at 1024 hypotheses
        try:
            D5, W5, T5, C5 = 4, 1024, 512, 2
            leg, lm5 = training_leg(device, D5, W5, T5, C5, 512, 10, 3, corpus)
            cfg5 = {"training": leg, "workload": "cfg5: depth=4 width=1024 length=512 V=256 2 contexts, stateful training, 512 streams, one GPU"}
            lm5._ws = None
            lm5._ws_key = None
            torch.cuda.empty_cache()
            lm5.prepare(hipabi.KL_PREC_SPLIT)
            cfg5["incremental"] = dict(incremental_leg(lm5, device, D5, W5, C5, 1024, 128), unit="hypotheses*chars/s",
                                       precision="split-bf16 (3 MFMA passes)")
            cfg5["incremental_n128"] = dict(incremental_leg(lm5, device, D5, W5, C5, 128, 128), unit="hypotheses*chars/s")
            # the rating window at the reference's batching: 1 stream x 512 chars, split precision
            lm5.reset_states(1)
            r5 = np.random.default_rng(12)
            xi5 = torch.from_numpy(r5.integers(1, VOC, size=(1, T5)).astype(np.int32)).to(device)
            ci5 = torch.from_numpy(r5.integers(0, 200, size=(1, 1, C5)).repeat(T5, axis=1).astype(np.int32)).to(device)
            for _ in range(3):
                lm5.forward_window(xi5, ci5)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            for _ in range(20):
                lm5.forward_window(xi5, ci5)
            torch.cuda.synchronize()
            ms5 = (time.perf_counter() - t5) / 20 * 1e3
            cfg5["rating_window"] = {"value": T5 / ms5 * 1e3, "unit": "chars/s", "ms_per_window": ms5, "streams": 1, "seq_len": T5,
                                     "precision": "split-bf16 (3 MFMA passes)"}
            del lm5
        except Exception as err:
            cfg5 = dict(cfg5 or {}, error=repr(err))
        torch.cuda.empty_cache()

    # ---- the reference's own model sizes (README.md:252-254: the published model is depth 2, width 128, length 256;
    # its README example width 64): training at 1024 streams and at the reference's batching
    ref_models = None
    other_streams = None
    if rank == 0 and world == 1 and not args.no_extra_shapes:
        ref_models = {}
        for Wm in (128, 64):
            try:
                entry = {}
                for Bs, st in ((1024, 50), (4096, 20), (1, 100)) if Wm == 128 else ((1024, 50), (1, 100)):
                    leg, lmm = training_leg(device, DEPTH, Wm, LENGTH, N_CTX, Bs, st, 5, corpus[:max(Bs, 64) * (CORPUS // 1024)])
                    del lmm
                    e = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "mfma_frac")}
                    # algorithmic HBM bytes of a training step per character and layer (time-major rows; bf16 unless noted): forward
                    # P f32 written + read (32 W), G written (8 W), C f32 (4 W), H (2 W), masked H (2 W); backward G, C, dH f32 read
                    # (8 + 4 + 4 W), dZ written (8 W); weight gradients read dZ twice and H / X once each (16 + 4 W); dX reads dZ (8 W)
                    # and writes dH (4 W): ~104 W bytes per character and layer
                    e["hbm_frac"] = leg["value"] * 104.0 * Wm * DEPTH / 1e9 / HBM_PEAK_GBS
                    entry["streams_%d" % Bs] = e
                ref_models["width_%d" % Wm] = entry
            except Exception as err:
                ref_models["width_%d" % Wm] = {"error": repr(err)}
        ref_models["note"] = ("depth 2, length 256, V=256, 1 context; width 128 = the published model's topology (README.md:252-254), on the "
                              "one-workgroup-per-row-block scans of lstm_scan_w128.hip (all layers in one launch up to 2048 streams); hbm_frac: ~104 W bytes per character and layer")
        torch.cuda.empty_cache()
        # cfg2 at stream counts beside the default: 1000 and 4096 streams (the engine pads / regroups such batches around the
        # counts the persistent scans take, HipLM._stream_groups), 2 context variables at the default count
        other_streams = {}
        for name, nc, Bo in (("streams_1000", N_CTX, 1000), ("streams_4096", N_CTX, 4096), ("contexts_2_streams_%d" % B, 2, B)):
            try:
                leg, lmo = training_leg(device, DEPTH, WIDTH, LENGTH, nc, Bo, 10, 3, corpus)
                del lmo
                other_streams[name] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "mfma_frac")}
            except Exception as err:
                other_streams[name] = {"error": repr(err)}
            torch.cuda.empty_cache()

    # ---- end to end: Rater.train over synthetic files (rank 0, one GPU): what the Python above the ABI costs
    end_to_end = None
    # (N = 1 only: Rater.train would see the process group and start collectives the other ranks are not in)
    if rank == 0 and world == 1 and not args.no_end_to_end:
        try:
            end_to_end = end_to_end_leg(B)
        except Exception as err:      # informative only
            end_to_end = {"error": repr(err)}

    cpu = None
    cpu_torch = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # (reported at N = 1 only)
        cpu = cpu_baseline()
        try:
            cpu_torch = cpu_baseline_torch()
        except Exception as err:      # informative only
            cpu_torch = {"error": repr(err)}

    if rank == 0:
        line = {
            "metric": "chars/sec training (stacked-LSTM forward+backward+Adam)",
            "value": value, "unit": "chars/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "cfg2: depth=2 width=512 length=256 V=256 1 context, stateful training, "
                                   "%d streams/GPU, synthetic 10M-char corpus (SURVEY.md 8d)" % B,
                       "streams_per_gpu": B, "global_batch": B * world, "seq_len": T,
                       "parallelism": "dp%d" % world,
                       # (a throughput run: the timed steps pass over each stream's windows several times, the loss means nothing)
                       "corpus_passes": (args.warmup + args.steps) / max(n_windows, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_torch": cpu_torch, "incremental": incremental, "beam_step_latency": beam,
            "rating_window": rating, "small_batch": small_batch, "cfg5": cfg5, "reference_models": ref_models, "other_stream_counts": other_streams, "end_to_end": end_to_end,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
