"""Standalone device times of the ingest and post-processing kernels (cf_normalize, cf_postprocess, cf_spans) on one stream, nothing
else running: the round-5 kernels (catfish_amd/csrc/ingest_post.hpp) against round 1's (CATFISH_INGEST_V1=1 behind the debug switch),
for the benchmark batch (256 x 4096 samples), the CLI's batch (1110 x 4096) and BASELINE configs[3]'s mix of lengths.
usage: python tools/exp_ingest_post.py          (sets the knobs itself, one child process per variant)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    import bench
    from catfish_amd.engine import HipEngine
    from catfish_amd.infer import padding_size_for
    eng = HipEngine(bench.load_weights(), device=0, max_windows_per_pass=4096)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    out = {}
    for name, lens in (("256 x 4096", [4096] * 256), ("1110 x 4096", [4096] * 1110),
                       ("600 reads LogUniform[512,16384]", np.rint(np.exp(rng.uniform(np.log(512), np.log(16384), size=600))).astype(int).tolist())):
        dacs = [bench.squiggle_dac(rng, int(n)) for n in lens]
        ln = np.array(lens, dtype=np.int64)
        dac_off = np.concatenate(([0], np.cumsum(ln)))
        n_win = np.array([(int(n) + padding_size_for(int(n))) // 35 for n in lens], dtype=np.int64)
        win_off = np.concatenate(([0], np.cumsum(n_win)))
        d_dac = torch.from_numpy(np.concatenate(dacs)).to(dev)
        d_do, d_wo = torch.from_numpy(dac_off).to(dev), torch.from_numpy(win_off).to(dev)
        x = torch.empty(int(win_off[-1]), 35, dtype=torch.float32, device=dev)
        probs = torch.rand(int(win_off[-1]) * 35, device=dev)
        # correlated scores, so that runs look like the network's (not coin flips): smooth the noise
        probs = torch.sigmoid(4 * torch.nn.functional.avg_pool1d((probs[None, None] - 0.5) * 8, 41, 1, 20)[0, 0])
        d_so, d_ln = torch.from_numpy(win_off * 35).to(dev), torch.from_numpy(ln).to(dev)
        res = {}
        for what, fn in (("normalize", lambda: eng.normalize_device(d_dac, d_do, d_wo, out=x)),
                         ("postprocess", lambda: eng.postprocess_device(probs, d_so, d_ln)),
                         ("spans", None)):
            if fn is None:
                labels = eng.postprocess_device(probs, d_so, d_ln)
                fn = lambda: eng.spans_device(labels)                                                   # noqa: E731
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(50):
                fn()
            t1.record()
            torch.cuda.synchronize()
            res[what + "_us"] = round(t0.elapsed_time(t1) / 50 * 1e3, 1)
        res["samples"] = int(ln.sum())
        res["normalize_GBps_algorithmic"] = round(6.0 * int(ln.sum()) / (res["normalize_us"] * 1e-6) / 1e9, 1)      # 2 B in + 4 B out
        res["postprocess_GBps_algorithmic"] = round(5.0 * int(win_off[-1]) * 35 / (res["postprocess_us"] * 1e-6) / 1e9, 1)   # 4 B in + 1 B out
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for label, env in (("round 5 kernels", {}), ("round 1 kernels (CATFISH_INGEST_V1=1)", {"CATFISH_DEBUG_KNOBS": "1", "CATFISH_INGEST_V1": "1"})):
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, universal_newlines=True)
            print(label)
            for name, r in json.loads(res.stdout.strip().splitlines()[-1]).items():
                print("  %-34s %s" % (name, r))
