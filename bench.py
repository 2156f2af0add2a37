#!/usr/bin/env python3
"""bench.py -- signal samples/s of the homopolymer-calling forward pass on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  For N > 1 it is launched under ``torch.distributed.run`` (one rank per GPU).

Workload (BASELINE.json configs[1], SURVEY.md 8d "config 2"): synthetic 4096-sample reads
(seeded DAC squiggles, median/MAD normalised exactly like catfish/infer.py:96-105), cut into
118 windows of 35 samples each (pad 34), 256 reads = 30 208 windows per step, fp32, weights of
the reference's bundled checkpoint ckpnt-30000.  A step = device-resident normalised windows
-> device-resident per-sample probabilities through the C ABI (cf_infer).  Reads shard across
ranks with no collective on the data path (weak scaling: every rank runs K steps of its own reads).

``value`` counts UN-PADDED signal samples (256 x 4096 per step and rank).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 4096
READS_PER_STEP = 256
WINDOW = 35
FLOP_PER_SAMPLE = 389504            # SURVEY.md 8d / BASELINE.md 2 (2 x 194 752 MAC)
FLOP_PER_SAMPLE_GRU128 = 2 * 2 * (128 + 64) * 192   # one CIN=128 biGRU layer: 2 dirs x 2 FLOP x 192x192 MAC
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)


def make_reads(n_reads, seed, return_dac=False):
    """Seeded synthetic reads -> normalised float32 windows [n_reads, 118, 35]."""
    rng = np.random.default_rng(seed)
    n_ev = READ_LEN // 4 + 8
    out = np.zeros((n_reads, 118 * WINDOW), dtype=np.float32)
    dacs = np.zeros((n_reads, READ_LEN), dtype=np.int16)
    for i in range(n_reads):
        dwell = rng.geometric(1.0 / 9.0, size=n_ev)
        while dwell.sum() < READ_LEN:
            dwell = np.concatenate([dwell, rng.geometric(1.0 / 9.0, size=n_ev)])
        levels = rng.normal(500.0, 60.0, size=len(dwell))
        sig = np.repeat(levels, dwell)[:READ_LEN] + rng.normal(0.0, 8.0, size=READ_LEN)
        dac = np.clip(np.rint(sig), 0, 2047).astype(np.int16)
        shift = np.median(dac)                       # infer.py:100-105
        scale = np.median(np.abs(dac - shift))
        out[i, :READ_LEN] = ((dac - shift) / scale).astype(np.float32)
        dacs[i] = dac
    return (out.reshape(n_reads, 118, WINDOW), dacs) if return_dac else out.reshape(n_reads, 118, WINDOW)


def load_weights():
    path = os.path.join(ROOT, "tests", "golden", "ckpnt-30000-inference.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def cpu_baseline():
    """CPU oracle timed on the host cores (oracle/cpu_baseline.py); runs BEFORE the GPU is initialised
    because it spawns worker processes."""
    from oracle import cpu_baseline as cb
    return cb.run(os.path.join(ROOT, "tests", "golden", "ckpnt-30000-inference.npz"), read_len=READ_LEN)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pool-reads", type=int, default=2048, help="distinct synthetic reads per rank")
    ap.add_argument("--streams", type=int, default=0, help="scratch slots/internal streams per engine (0 = library default)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="arithmetic of the biGRU matmuls (fp32 = exact fp32 MFMA, the BASELINE configs[1] dtype)")
    ap.add_argument("--no-extra-precisions", action="store_true",
                    help="skip the short informational legs that time the other precisions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="run untimed steps for this long before the W warm-up steps: the GPU clocks down while the CPU "
                         "baseline leg (or process start-up) keeps it idle, and a short W would time the clock ramp")
    ap.add_argument("--no-kernel-events", action="store_true", help="disable per-kernel HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "CATFISH_BENCH_DEVICE" in os.environ:      # rehearsal of the N > 1 path on a box with fewer GPUs than ranks
        local_rank = int(os.environ["CATFISH_BENCH_DEVICE"])
    cpu_res = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline()

    import torch
    import torch.distributed as dist
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        # The data path has no collective; the process group only serves the timing barrier and the MAX over
        # ranks.  RCCL ("nccl") first; if it cannot initialise on this node, fall back to gloo (host barrier).
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            probe = torch.zeros(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            backend = "nccl"
        except Exception as exc:      # pragma: no cover - depends on the node
            sys.stderr.write("bench.py: nccl init failed (%s); using gloo for the timing barrier\n" % exc)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")
            backend = "gloo"

    from catfish_amd.engine import HipEngine
    weights = load_weights()
    eng = HipEngine(weights, device=local_rank, max_windows_per_pass=READS_PER_STEP * 118, n_streams=args.streams,
                    precision=args.precision)

    # every rank owns its own shard of reads (seeded by rank): no data-path collective
    n_pool = max(READS_PER_STEP, (args.pool_reads // READS_PER_STEP) * READS_PER_STEP)
    reads = make_reads(n_pool, seed=1000 + rank)
    n_batches = n_pool // READS_PER_STEP
    dev = torch.device("cuda", local_rank)
    batches = [torch.from_numpy(reads[b * READS_PER_STEP:(b + 1) * READS_PER_STEP].reshape(-1, WINDOW)).to(dev)
               for b in range(n_batches)]
    outs = [torch.empty(READS_PER_STEP * 118 * WINDOW, dtype=torch.float32, device=dev) for _ in range(2)]

    def barrier():
        if world > 1:
            dist.barrier()

    tw = time.perf_counter()
    while (time.perf_counter() - tw) * 1e3 < args.prewarm_ms:          # clock warm-up, outside the W + K steps
        for i in range(4):
            eng.infer_device(batches[i % n_batches], out=outs[i & 1])
        torch.cuda.synchronize()
    for i in range(args.warmup):
        eng.infer_device(batches[i % n_batches], out=outs[i & 1])
    torch.cuda.synchronize()

    if not args.no_kernel_events:
        eng.profile_enable(True, every=4)        # HIP events around every kernel of every 4th step of the timed region
        eng.profile_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.infer_device(batches[i % n_batches], out=outs[i & 1])
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kern = eng.profile_read() if not args.no_kernel_events else {}
    eng.profile_enable(False)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    samples_per_step = READS_PER_STEP * READ_LEN          # un-padded, per rank
    value = world * args.steps * samples_per_step / dt

    result = None
    if rank == 0:
        # parity spot-check of the benchmarked configuration against the oracle (not timed)
        from oracle import catfish_oracle as oracle
        chk = reads[:2]     # the launch keeps the benchmark's size (rocprof averages stay comparable)
        got = eng.infer_device(batches[0], out=outs[0]).cpu().numpy()[:2 * 118 * WINDOW].astype(np.float64)
        want64 = oracle.forward(chk.reshape(-1, WINDOW), weights, np.float64)
        want32 = oracle.forward(chk.reshape(-1, WINDOW), weights, np.float32)
        max_dp = float(np.abs(got - want64).max())
        match = float(np.mean((got >= 0.5) == (want32 >= 0.5)))

        roof = None
        if "gru_layer_mid" in kern:
            ms, n = kern["gru_layer_mid"]
            avg_s = ms / n * 1e-3
            achieved = FLOP_PER_SAMPLE_GRU128 * samples_per_step / avg_s / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as fh:
                    traffic = json.load(fh).get("gru_layer_mid_bytes_per_launch")
            peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
            kname = "gru_layer_kernel<128,false>" if args.precision == "fp32" else \
                "gru_layer_bf16_kernel<128,false,%d>" % (2 if args.precision == "bf16x3" else 1)
            if args.precision != "fp32":
                traffic = None      # profiles/traffic.json was collected for the fp32 kernel
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak, "traffic": traffic,
                    "kernel": kname, "avg_launch_ms": ms / n,
                    "flop_per_launch": FLOP_PER_SAMPLE_GRU128 * samples_per_step}
        result = {
            "metric": "signal samples/s classified",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "prewarm_ms": args.prewarm_ms, "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (split-operand fp32 emulation, f32 accumulate)",
                                           "bf16": "bf16 (f32 accumulate)"}[args.precision], "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 4096-sample reads, 256 reads (30208 windows of 35) "
                                   "per step and GPU, fp32, ckpnt-30000 weights",
                       "reads_per_step": READS_PER_STEP, "read_len": READ_LEN, "windows_per_step": READS_PER_STEP * 118,
                       "parallelism": "reads sharded over %d GPU(s), no collective" % world},
            "roofline": roof,
            "whole_pass": {"achieved_tflops": value / world * FLOP_PER_SAMPLE / 1e12,
                           "frac_of_mfma_peak": value / world * FLOP_PER_SAMPLE / 1e12 /
                           (PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS)},
            "kernels_ms": {k: v[0] / v[1] for k, v in kern.items()},
            "parity": {"max_abs_dp_vs_fp64_oracle": max_dp, "label_match_vs_fp32_oracle": match, "gate": 1e-4},
        }
        result["cpu_baseline"] = cpu_res
    # informational legs: the same workload with the other GRU arithmetics (not the headline value)
    if world == 1 and not args.no_extra_precisions:
        extra = {}
        for prec in ("fp32", "bf16x3", "bf16"):
            if prec == args.precision:
                continue
            e2 = HipEngine(weights, device=local_rank, max_windows_per_pass=READS_PER_STEP * 118, precision=prec)
            tw = time.perf_counter()                 # warm up by time: the GPU clocks down while the CPU oracle ran
            while time.perf_counter() - tw < 0.5:
                for i in range(4):
                    e2.infer_device(batches[i % n_batches], out=outs[i & 1])
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            n2 = max(5, args.steps // 2)
            for i in range(n2):
                e2.infer_device(batches[i % n_batches], out=outs[i & 1])
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            from oracle import catfish_oracle as oracle
            got = e2.infer_device(batches[0], out=outs[0]).cpu().numpy()[:118 * WINDOW].astype(np.float64)
            want = oracle.forward(reads[0], weights, np.float64)
            extra[prec] = {"value": n2 * samples_per_step / d2, "unit": "samples/s", "ms_per_step": d2 / n2 * 1e3,
                           "max_abs_dp_vs_fp64_oracle": float(np.abs(got - want).max())}
            e2.close()
        result["other_precisions"] = extra
        # informational: host-to-host rate of the streaming pipeline (pinned int16 DAC in, spans out, PCIe inclusive)
        from catfish_amd.pipeline import ReadPipeline
        _, dacs = make_reads(READS_PER_STEP, seed=77, return_dac=True)
        pipe = ReadPipeline(eng, max_samples_per_batch=READS_PER_STEP * READ_LEN)
        pb = [list(dacs)] * 12
        for _ in pipe.run(pb[:6], as_lists=False):
            pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in pipe.run(pb, as_lists=False):
            pass
        d3 = time.perf_counter() - t1
        result["host_to_host_pipeline"] = {"value": len(pb) * READS_PER_STEP * READ_LEN / d3, "unit": "samples/s",
                                           "ms_per_batch": d3 / len(pb) * 1e3,
                                           "what": "pinned int16 DAC -> cf_normalize -> cf_infer -> cf_postprocess -> cf_spans -> "
                                                   "host span table, double-buffered; never the headline value"}
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
