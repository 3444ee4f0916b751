#!/usr/bin/env python3
"""Rescoring across the GPUs of one node: whole documents are independent units (pages inside a document depend on
each other through the carried state / traceback, rate.py:150-186, 263-265), so they are dealt round-robin to one
process per GPU and NOTHING is exchanged between the processes ("no collectives").

  python tools/rescore_shard.py --model model.h5 --gpus 8 [--mode rate|generate] [--out DIR] FILE_OR_DIR ...

mode rate      per document: Rater.rate over the whole text (context = ceil(year / 10) from `author_title_year.txt`
               names, rating.py:993-999), state reset between documents; writes <out>/<name>.json with the mean
               log2-probability per character and the perplexity
mode generate  per document: its first line is the prompt; Rater.generate continues it by 64 characters, 3 variants

The parent only starts the workers (each with ONE visible GPU, HIP_VISIBLE_DEVICES) and sums their reports; it never
touches a GPU itself.  A worker that fails makes the parent exit non-zero after the others have finished.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def expand(items):
    paths = []
    for item in items:
        if os.path.isdir(item):
            paths.extend(sorted(os.path.join(item, n) for n in os.listdir(item) if os.path.isfile(os.path.join(item, n))))
        else:
            paths.append(item)
    return paths


def shard(paths, n):
    """document k goes to GPU k % n (round-robin keeps the shards balanced without looking at sizes)"""
    return [paths[i::n] for i in range(n)]


def worker(args):
    from math import log
    from ocrd_keraslm_amd.lib import Rater, windows
    rater = Rater()
    rater.load_config(args.model)
    if args.mode == "generate":
        rater.stateful, rater.incremental = False, True
    rater.configure()
    rater.load_weights(args.model)
    mine = shard(expand(args.data), args.gpus)[args.worker]
    os.makedirs(args.out, exist_ok=True)
    chars = 0
    t0 = time.perf_counter()
    for path in mine:
        with open(path, encoding="utf-8") as f:
            text = windows.normalize(f.read())
        context = windows.context_from_filename(path)
        name = os.path.basename(path)
        if args.mode == "rate":
            if len(text) < 2:
                continue
            rater.model.reset_states(1)                      # documents are independent: no carry-over between them
            probs = rater.rate(text, context)
            bits = -sum(log(max(p, 1e-99), 2) for p in probs[1:]) / max(len(probs) - 1, 1)
            result = {"document": name, "chars": len(text), "bits_per_char": bits, "perplexity": 2.0 ** bits}
            chars += len(text)
        else:
            prompt = text.split("\n", 1)[0][:64] or " "
            variants = rater.generate(prompt, 64, context, 3)
            result = {"document": name, "prompt": prompt, "variants": [prompt[:-1] + v for v in variants]}
            chars += 64 * 3
        with open(os.path.join(args.out, name + ".json"), "w", encoding="utf-8") as f:
            json.dump(result, f, ensure_ascii=False)
    el = time.perf_counter() - t0
    print(json.dumps({"worker": args.worker, "documents": len(mine), "chars": chars, "seconds": el}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", choices=("rate", "generate"), default="rate")
    ap.add_argument("--out", default="rescored")
    ap.add_argument("--worker", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("data", nargs="+")
    args = ap.parse_args()
    if args.worker >= 0:
        return worker(args)
    procs = []
    t0 = time.perf_counter()
    same_gpu = os.environ.get("KL_RESCORE_SAME_GPU") == "1"       # rehearsal of N workers on a one-GPU box
    reports, failed = [], 0

    def collect(p):
        nonlocal failed
        out, _ = p.communicate()
        if p.returncode != 0:
            failed += 1
        else:
            reports.append(json.loads(out.strip().splitlines()[-1]))

    for g in range(args.gpus):
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(g), CUDA_VISIBLE_DEVICES=str(g))
        if same_gpu:
            env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = "0"
        cmd = [sys.executable, os.path.abspath(__file__), "--model", args.model, "--gpus", str(args.gpus), "--mode", args.mode,
               "--out", args.out, "--worker", str(g)] + args.data
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        if same_gpu:
            # one after the other: the persistent scans assume the GPU to themselves (co-resident workgroups), two
            # processes launching them at once on ONE card break the hand-offs (see tests/ddp_hip_worker.py)
            collect(p)
        else:
            procs.append(p)
    for p in procs:
        collect(p)
    el = time.perf_counter() - t0
    chars = sum(r["chars"] for r in reports)
    print(json.dumps({"gpus": args.gpus, "documents": sum(r["documents"] for r in reports), "chars": chars,
                      "seconds": el, "chars_per_s": chars / el if el > 0 else 0.0, "failed_workers": failed,
                      "per_worker": reports}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main() or 0)
