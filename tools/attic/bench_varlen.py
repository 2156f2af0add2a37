"""BASELINE config 4: variable-length reads 512..16384 samples, length-bucketed packed batches, 1 GPU.

10k reads (default 2048 for a quick run), lengths ~ LogUniform[512, 16384] (seed 2); every bucket
packs all windows of its reads into one [sum N_i, 35] tensor + offset tables (no padding beyond each
read's own tail).  Timed region: device-resident packed windows -> probabilities -> device labels
(cf_infer + cf_postprocess).  Parity is reported as label match rate / max |dp| against the fp32
oracle on a few reads.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402
from catfish_amd import batching  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16x3", "bf16"])
    ap.add_argument("--max-windows", type=int, default=30208)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(2)
    lens = np.rint(np.exp(rng.uniform(np.log(512), np.log(16384), size=args.reads))).astype(np.int64)
    sigs = [rng.standard_normal(int(n)).astype(np.float32) * 1.3 for n in lens]
    w = bench.load_weights()
    eng = HipEngine(w, device=0, max_windows_per_pass=args.max_windows, precision=args.precision)
    dev = torch.device("cuda", 0)
    buckets = batching.length_buckets(lens, args.max_windows)
    packed = []
    for b in buckets:
        pk = batching.pack_reads([sigs[i] for i in b])
        packed.append((torch.from_numpy(pk.x).to(dev), torch.from_numpy(pk.sample_offsets).to(dev),
                       torch.from_numpy(pk.lengths).to(dev)))

    def run_all():
        for x, offs, ln in packed:
            probs = eng.infer_device(x)
            eng.postprocess_device(probs, offs, ln)
    run_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        run_all()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.repeat
    # parity on 3 reads
    from oracle import catfish_oracle as oracle
    idx = [0, 1, 2]
    res, probs = batching.infer_packed(eng, batching.pack_reads([sigs[i] for i in idx]), return_probs=True)
    match, maxdp = [], 0.0
    for k, i in enumerate(idx):
        x, pad = oracle.pad_and_window(sigs[i])
        want = oracle.forward(x, w, np.float32)[:-pad]
        match.append(np.mean((probs[k] >= 0.5) == (want >= 0.5)))
        maxdp = max(maxdp, float(np.abs(probs[k] - want).max()))
    print(json.dumps({"metric": "signal samples/s classified (config 4: variable-length packed)", "precision": args.precision,
                      "value": float(lens.sum()) / dt, "unit": "samples/s", "reads": args.reads, "buckets": len(buckets),
                      "total_samples": int(lens.sum()), "padding_overhead": float(sum(int(p[0].shape[0]) for p in packed) * 35 / lens.sum() - 1),
                      "label_match_vs_fp32_oracle": float(np.mean(match)), "max_abs_dp_vs_fp32_oracle": maxdp}))
    eng.close()


if __name__ == "__main__":
    main()
