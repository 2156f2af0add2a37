"""Host-to-host rate (PCIe inclusive): pinned int16 DAC reads -> spans, double-buffered (catfish_amd/pipeline.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from catfish_amd.engine import HipEngine  # noqa: E402
from catfish_amd.pipeline import ReadPipeline  # noqa: E402
import bench  # noqa: E402
from oracle import catfish_oracle as oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--reads-per-batch", type=int, default=256)
    ap.add_argument("--depth", type=int, default=2, help="batches in flight (ReadPipeline.depth)")
    args = ap.parse_args()
    w = bench.load_weights()
    dac = oracle.synthetic_dac(512, 4096, seed=11)
    batches = [[dac[(b * args.reads_per_batch + i) % len(dac)] for i in range(args.reads_per_batch)]
               for b in range(args.batches)]
    eng = HipEngine(w, device=0, max_windows_per_pass=args.reads_per_batch * 118, precision=args.precision)
    pipe = ReadPipeline(eng, max_samples_per_batch=args.reads_per_batch * 4096, depth=args.depth)
    list(pipe.run(batches[:3]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_spans = 0
    for res in pipe.run(batches, as_lists=False):
        n_spans += len(res[0])
    dt = time.perf_counter() - t0
    # check one batch against the oracle
    res = next(iter(pipe.run(batches[:1])))
    ok = True
    for i in range(2):
        w_spans, w_len, _ = oracle.infer_read(oracle.normalize_raw_signal(batches[0][i]), w, np.float32)
        ok = ok and res[i] == (w_spans, w_len)
    print(json.dumps({"metric": "host-to-host signal samples/s (int16 DAC in, spans out, PCIe inclusive)",
                      "precision": args.precision, "value": args.batches * args.reads_per_batch * 4096 / dt,
                      "ms_per_batch": dt / args.batches * 1e3, "spans": n_spans,
                      "spans_match_oracle_fp32_on_2_reads": bool(ok)}))
    eng.close()


if __name__ == "__main__":
    main()
