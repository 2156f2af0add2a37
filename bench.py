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

``value`` counts UN-PADDED signal samples (256 x 4096 per step and rank).  The line also carries
informational objects that are never the value: ``other_precisions`` (bf16x3 / bf16 on the same
workload, each with its own roofline), ``config4`` (BASELINE configs[3]: 10 000 variable-length reads, packed,
bf16), ``latency`` (118- and 256-window calls), ``config5`` (BASELINE configs[4]: the training step), ``host_to_host_pipeline`` (PCIe-inclusive), ``sharded_gather`` (BASELINE configs[2]'s path:
reads sharded over the ranks, per-rank streaming pipeline, gloo host gather on rank 0) and ``cli_end_to_end`` (the same
shape through the product CLI: files in, merged chunk coordinates out as JSON, the merge tail on every rank).
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
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md, HBM3E peak
ACHIEVABLE_HBM_GBS = 6300.0         # what a device-to-device copy sustains on this chip (same guide)
MID_LAYER_BYTES_PER_SAMPLE = {"fp32": 128 * 4 * 2, "bf16": 128 * 2 * 2}   # a CIN = 128 biGRU layer: input slab + output slab
SHARDED_READS_PER_RANK = 12500      # BASELINE configs[2]: 100 000 reads over 8 GPUs
CONFIG4_READS = 10000               # BASELINE configs[3]
CONFIG4_MAX_WINDOWS = 131072        # windows per packed launch: the regime catfish_amd.cli uses for big jobs (cli.run_pipeline)
CONFIG4_PARITY_READS = 32           # reads of it checked against the fp32 oracle (shortest, longest, 30 spread over the rest)
RANK_NOTES = {}                     # what this rank's informational legs want in its entry of the line's ``ranks`` list


_T_START = time.perf_counter()


def progress(what):
    """One line on stderr per stage (never stdout: that carries the ONE JSON line): a long run shows where it is, and a hung one
    shows where it stopped.  ``CATFISH_BENCH_WATCHDOG_S=N`` additionally dumps every thread's Python stack to stderr every N
    seconds (faulthandler), which names the call a hang sits in."""
    sys.stderr.write("bench.py [%7.1f s] rank %s: %s\n" % (time.perf_counter() - _T_START, os.environ.get("RANK", "0"), what))
    sys.stderr.flush()


def squiggle_dac(rng, length):
    """One seeded synthetic read: int16 DAC squiggle of SURVEY 8d (levels N(500,60^2), dwell Geometric(1/9),
    noise N(0,8^2), clipped to [0,2047])."""
    n_ev = length // 4 + 8
    dwell = rng.geometric(1.0 / 9.0, size=n_ev)
    while dwell.sum() < length:
        dwell = np.concatenate([dwell, rng.geometric(1.0 / 9.0, size=n_ev)])
    levels = rng.normal(500.0, 60.0, size=len(dwell))
    sig = np.repeat(levels, dwell)[:length] + rng.normal(0.0, 8.0, size=length)
    return np.clip(np.rint(sig), 0, 2047).astype(np.int16)


def make_reads(n_reads, seed, return_dac=False):
    """Seeded synthetic reads -> normalised float32 windows [n_reads, 118, 35]."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_reads, 118 * WINDOW), dtype=np.float32)
    dacs = np.zeros((n_reads, READ_LEN), dtype=np.int16)
    for i in range(n_reads):
        dac = squiggle_dac(rng, READ_LEN)
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
    because it spawns worker processes.  Two figures: the 1-GPU box's CPU share (16 workers) as ``value`` and
    every core this process may run on as ``all_cores``."""
    from oracle import cpu_baseline as cb
    wpath = os.path.join(ROOT, "tests", "golden", "ckpnt-30000-inference.npz")
    res = cb.run(wpath, read_len=READ_LEN, budget_s=8.0, workers=16)
    n_aff = cb.affinity_cores()
    if n_aff > res["cores"]:
        full = cb.run(wpath, read_len=READ_LEN, budget_s=6.0, workers=min(n_aff, 256))
        res["all_cores"] = {"value": full["value"], "unit": full["unit"], "cores": full["cores"], "sample": full["sample"]}
    return res


def traffic_record(precision="fp32"):
    """HBM bytes per launch of the dominant kernel from the last FETCH_SIZE / WRITE_SIZE passes (profiles/traffic.json for
    fp32, profiles/traffic_bf16.json for bf16, written by tools/collect_traffic.sh): a STORED measurement -- PMC counters
    cannot be read from inside this process -- so the line says which commit and command it came from."""
    name = {"fp32": "traffic.json", "bf16": "traffic_bf16.json", "bf16x3": "traffic_bf16x3.json"}.get(precision)
    tpath = os.path.join(ROOT, "profiles", name) if name else None
    if tpath is None or not os.path.exists(tpath):
        return None, None
    with open(tpath) as fh:
        t = json.load(fh)
    src = {"file": "profiles/" + name, "kind": "stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run",
           "commit": t.get("commit"), "collected": t.get("collected"), "kernel_avg_ms_at_collection": t.get("kernel_avg_ms")}
    return t, src


def mid_roofline(kern, precision, samples_per_launch, traffic=None, traffic_source=None):
    if "gru_layer_mid" not in kern:
        return None
    ms, n = kern["gru_layer_mid"]
    if traffic is None and traffic_source is None:
        t, traffic_source = traffic_record(precision)
        traffic = t.get("gru_layer_mid_bytes_per_launch") if t else None
    achieved = FLOP_PER_SAMPLE_GRU128 * samples_per_launch / (ms / n * 1e-3) / 1e12
    peak = PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    kname = {"fp32": "gru_layer_kernel<128,false>", "bf16x3": "gru_bf16x3_pipe_kernel<128,false>",
             "bf16": "gru_bf16_pipe_kernel<128,false>"}[precision]
    launch_s = ms / n * 1e-3
    roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": traffic, "traffic_source": traffic_source, "kernel": kname, "avg_launch_ms": ms / n,
            "flop_per_launch": FLOP_PER_SAMPLE_GRU128 * samples_per_launch}
    return nearer_roof(roof, precision, samples_per_launch, launch_s)


def nearer_roof(roof, precision, samples, seconds):
    """Report the dominant kernel against the roof it sits nearer to.  Its own algorithmic bytes are its input and output
    slabs (128 features per sample in, 128 out, 4 B each in fp32 / bf16x3 and 2 B in bf16); the matrix
    view is kept as ``mfma_view`` when HBM is the binding roof, and the measured fabric traffic (when a stored PMC record
    exists) as ``traffic`` plus ``traffic_rate_frac`` = traffic / time / 8 TB/s (about 6.3 TB/s is achievable on a copy)."""
    bytes_per_sample = MID_LAYER_BYTES_PER_SAMPLE["bf16" if precision == "bf16" else "fp32"]
    hbm = bytes_per_sample * samples / seconds / 1e9
    hbm_view = {"achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS,
                "algorithmic_bytes": bytes_per_sample * samples, "achievable_peak": ACHIEVABLE_HBM_GBS}
    if roof.get("traffic"):
        hbm_view["traffic_rate"] = roof["traffic"] / seconds / 1e9
        hbm_view["traffic_rate_frac"] = hbm_view["traffic_rate"] / PEAK_HBM_GBS
    # Which roof binds is decided on the quantities the line reports: algorithmic bytes against HBM, and against the matrix pipe
    # the FLOPs the kernel ISSUES -- bf16x3 spends three bf16 MFMAs per product (a_hi w_hi + a_lo w_hi + a_hi w_lo), so its
    # pipe does 3x the algorithmic FLOPs.  (bf16: the matrix roof is 16x further away than in fp32 while the slabs only halve.)
    issue_factor = 3.0 if precision == "bf16x3" else 1.0
    pipe_frac = roof["frac"] * issue_factor
    if issue_factor != 1.0:
        # the dense peak assumes 2.4 GHz; under bf16 MFMA load on random data the chip holds ~1.5 GHz (tools/exp_x3_stamps.py), and
        # MI355X_MICROARCH.md ('DVFS give-back' item 1) quotes 1,247 TFLOP/s for a tuned 256^2 GEMM on random operands
        roof["matrix_pipe_issued"] = {"achieved": roof["achieved"] * issue_factor, "peak": roof["peak"], "unit": "TFLOP/s",
                                      "frac": pipe_frac, "mfma_per_product": int(issue_factor),
                                      "tuned_gemm_on_random_data_tflops": 1247.0,
                                      "frac_of_that": roof["achieved"] * issue_factor / 1247.0}
    # The HBM side of the comparison is the LARGER of the algorithmic-bytes fraction and -- when a stored PMC record exists -- the
    # measured fabric traffic's fraction (both directions of a layer re-read the input slab: traffic is ~1.5x the slabs).  ``frac``
    # and ``achieved`` stay algorithmic (the contract of the line); ``decided_by`` says which quantities were compared, and a gap
    # below 10 % is called what it is: a near tie, the kernel sits against both roofs.
    hbm_frac = max(hbm_view["frac"], hbm_view.get("traffic_rate_frac", 0.0))
    hbm_what = ("measured HBM traffic (stored PMC record) %.3f of peak" % hbm_view["traffic_rate_frac"]
                if hbm_view.get("traffic_rate_frac", 0.0) > hbm_view["frac"] else "algorithmic HBM bytes %.3f of peak" % hbm_view["frac"])
    pipe_what = ("ISSUED matrix-pipe FLOPs (%d MFMAs per product) %.3f of peak" % (int(issue_factor), pipe_frac)
                 if issue_factor != 1.0 else "algorithmic matrix FLOPs %.3f of peak" % pipe_frac)
    near_tie = abs(hbm_frac - pipe_frac) <= 0.10 * max(hbm_frac, pipe_frac)
    roof["near_tie"] = bool(near_tie)
    roof["decided_by"] = "%s against %s%s" % (hbm_what, pipe_what, "; within 10 %: mfma~hbm, a near tie" if near_tie else "")
    if hbm_frac > pipe_frac:
        mfma_view = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
        roof.update(bound="hbm", achieved=hbm, peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm / PEAK_HBM_GBS, mfma_view=mfma_view,
                    hbm_detail=hbm_view)
    else:
        roof["hbm_view"] = hbm_view
    return roof


def describe_ranks(result, ranks, world, backend, nccl_error, ranks_seen, visible_devices, rccl_probe=None):
    """Who ran what where, into the line (pure bookkeeping, tested on the CPU): ``ranks`` (every rank's record), ``collective``,
    ``distinct_devices``, ``cpu_sets_disjoint``, and -- when ranks SHARED a card (``CATFISH_BENCH_DEVICE``, or fewer GPUs than
    ranks) -- the rehearsal marking: ``rehearsal: true``, ``value: null`` (the number moves to ``rehearsal_value``), ``n_gpus`` = the
    cards actually driven, ``n_ranks`` = N, the same on the two end-to-end legs, ``whole_node_end_to_end: null``.  Nothing in a
    rehearsal line is an N-GPU number, and nothing in it can be read as one."""
    from catfish_amd import placement
    # a card = (host, PCI address, UUID): two ranks drive the same card only when ALL of it agrees (a runtime that reports one UUID
    # for every card must not turn a real N-GPU run into a "rehearsal")
    cards = sorted({(r["host"], r["pci_bus_id"], r["uuid"]) for r in ranks})
    distinct = len(cards)
    result["ranks"] = ranks
    result["collective"] = {"backend": backend, "nccl_init_error": nccl_error, "ranks_seen": ranks_seen,
                            "ranks_seen_is_world": bool(ranks_seen == world), "rccl_probe": rccl_probe,
                            "what": "timing barrier + MAX over ranks on the host group (gloo), whatever RCCL does: the data path has no "
                                    "collective, so the clock does not depend on one; ranks_seen = all_reduce(sum) of 1 per rank over that "
                                    "group; rccl_probe = one all-reduce of a device scalar over a second, RCCL group with a short timeout "
                                    "(did this node's RCCL see all ranks?)" if world > 1 else "single process: no process group"}
    result["distinct_devices"] = distinct
    result["visible_devices_rank0"] = visible_devices
    cpu_sets = [set(placement.parse_cpulist(r["cpus"])) for r in ranks]
    result["cpu_sets_disjoint"] = bool(all(not (cpu_sets[i] & cpu_sets[j]) for i in range(len(ranks)) for j in range(i + 1, len(ranks))))
    result["config"]["parallelism"] = ("reads sharded over %d GPU(s), one rank each, no collective" % world if distinct == world else
                                       "%d ranks on %d device(s): REHEARSAL of the launch path, not a %d-GPU measurement" % (world, distinct, world))
    if distinct < world:
        result["rehearsal"] = True
        result["rehearsal_value"] = result.pop("unverified_value", None) or result["value"]
        result["value"] = None
        result["n_ranks"] = world
        result["n_gpus"] = distinct
        for leg in ("sharded_gather", "cli_end_to_end"):
            if isinstance(result.get(leg), dict) and "n_gpus" in result[leg]:
                result[leg]["n_ranks"], result[leg]["n_gpus"], result[leg]["rehearsal"] = world, distinct, True
    else:
        result["rehearsal"] = False
    # the two host-inclusive rates at the top level, next to ``value`` (which stays the device-resident configs[1] rate the
    # roofline is computed on): what a caller holding host buffers gets from one GPU, and what the whole job delivers from files
    h2h = result.get("host_to_host_pipeline")
    result["host_to_host_value"] = h2h.get("value") if isinstance(h2h, dict) else None
    cli_leg = result.get("cli_end_to_end")
    result["whole_node_end_to_end"] = cli_leg.get("value") if isinstance(cli_leg, dict) else None
    if result["rehearsal"]:
        result["rehearsal_whole_node_end_to_end"] = result["whole_node_end_to_end"]
        result["whole_node_end_to_end"] = None
    return result


RCCL_PROBE_TIMEOUT_S = 90.0


def probe_rccl(dist, torch, local_rank, timeout_s=None):
    """Bring RCCL ("nccl" on ROCm) up over a SECOND process group and all-reduce one device scalar: -> dict(ok, ranks_seen, seconds,
    error, timeout_s).  Bounded: the group is created with ``timeout_s`` and waits block (TORCH_NCCL_BLOCKING_WAIT), so a collective
    that does not complete raises here instead of hanging the run or being torn down by the watchdog; any failure is recorded and
    the group dropped.  The default (gloo) group must exist already; nothing the benchmark times uses the probed group."""
    import datetime
    timeout_s = float(os.environ.get("CATFISH_RCCL_PROBE_TIMEOUT_S", RCCL_PROBE_TIMEOUT_S)) if timeout_s is None else float(timeout_s)
    out = {"ok": False, "ranks_seen": None, "seconds": None, "error": None, "timeout_s": timeout_s}
    if os.environ.get("CATFISH_RCCL_PROBE", "1") == "0":
        out["error"] = "skipped (CATFISH_RCCL_PROBE=0)"
        return out
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
    t0 = time.perf_counter()
    group = None
    try:
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        one = torch.ones(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(one, group=group)
        torch.cuda.synchronize()
        out["ranks_seen"] = int(round(float(one.item())))
        out["ok"] = out["ranks_seen"] == dist.get_world_size()
    except Exception as exc:      # pragma: no cover - depends on the node
        out["error"] = "%s: %s" % (type(exc).__name__, " ".join(str(exc).split())[:400])
        sys.stderr.write("bench.py: RCCL probe failed (%s); the timing barrier is on gloo either way\n" % out["error"])
    out["seconds"] = time.perf_counter() - t0
    # every rank learns whether ALL ranks got through (a probe that worked on some ranks only is not a working RCCL) -- over gloo
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, bool(out["ok"]))
    if not all(flags) and out["ok"]:
        out["ok"], out["error"] = False, "the probe failed on rank(s) %s" % [r for r, f in enumerate(flags) if not f]
    if group is not None:
        try:
            dist.destroy_process_group(group)
        except Exception as exc:      # pragma: no cover
            out["destroy_error"] = "%s: %s" % (type(exc).__name__, " ".join(str(exc).split())[:200])
    return out


def timed_steps(eng, batches, outs, n, torch):
    t0 = time.perf_counter()
    for i in range(n):
        eng.infer_device(batches[i % len(batches)], out=outs[i & 1])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def warm_by_time(eng, batches, outs, seconds, torch):
    tw = time.perf_counter()
    while time.perf_counter() - tw < seconds:
        for i in range(4):
            eng.infer_device(batches[i % len(batches)], out=outs[i & 1])
        torch.cuda.synchronize()


def leg_config4(weights, local_rank, torch, profile_only=False):
    """BASELINE configs[3]: variable-length reads 512..16384 (log-uniform, seed 2), DAC squiggles normalised on the
    device, packed into length-bucketed launches, bf16.  Timed: device-resident packed windows -> probabilities ->
    device labels (cf_infer + cf_postprocess).  Judged by oracle/tolerances.py (label match against the fp32 oracle and a
    loose probability bound, on 32 reads): ``parity.passed``; a failing leg reports no value.
    ``profile_only``: one warm pass and the timed passes, nothing else -- the command tools/collect_traffic_config4.sh puts
    under rocprofv3, so that every launch it counts belongs to a whole pass over the 10 000 reads."""
    from catfish_amd import batching
    from catfish_amd.engine import HipEngine
    from catfish_amd.infer import padding_size_for
    from oracle import catfish_oracle as oracle
    from oracle import tolerances as tol
    rng = np.random.default_rng(2)
    lens = np.rint(np.exp(rng.uniform(np.log(512), np.log(16384), size=CONFIG4_READS))).astype(np.int64)
    dacs = [squiggle_dac(rng, int(n)) for n in lens]
    max_windows = CONFIG4_MAX_WINDOWS
    eng = HipEngine(weights, device=local_rank, max_windows_per_pass=max_windows, precision="bf16")
    dev = torch.device("cuda", local_rank)
    packed = []
    for b in batching.length_buckets(lens, max_windows):
        ln = lens[b]
        dac_off = np.concatenate(([0], np.cumsum(ln)))
        n_win = np.array([(int(n) + padding_size_for(int(n))) // WINDOW for n in ln], dtype=np.int64)
        win_off = np.concatenate(([0], np.cumsum(n_win)))
        x = torch.empty(int(win_off[-1]), WINDOW, dtype=torch.float32, device=dev)
        eng.normalize_device(torch.from_numpy(np.concatenate([dacs[i] for i in b])).to(dev),
                             torch.from_numpy(dac_off).to(dev), torch.from_numpy(win_off).to(dev), out=x)
        packed.append((x, torch.from_numpy(win_off * WINDOW).to(dev), torch.from_numpy(ln).to(dev), int(ln.sum())))
    torch.cuda.synchronize()

    def run_all():
        for x, offs, ln, _ in packed:
            eng.postprocess_device(eng.infer_device(x), offs, ln)
    tw = time.perf_counter()
    while True:
        run_all()
        torch.cuda.synchronize()
        if profile_only or time.perf_counter() - tw >= 0.4:
            break
    eng.profile_enable(True, every=1)
    eng.profile_reset()
    rep = 3
    t0 = time.perf_counter()
    for _ in range(rep):
        run_all()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / rep
    kern = eng.profile_read()
    eng.profile_enable(False)
    if profile_only:
        eng.close()
        return {"passes": rep + 1, "buckets": len(packed), "ms_per_pass": dt * 1e3,
                "kernels_ms_per_pass": {k: v[0] / rep for k, v in kern.items()}}
    # parity on 32 reads (shortest, longest, 30 spread over the rest) against the fp32 oracle, as SURVEY 8d asks for this config
    idx = sorted({int(np.argmin(lens)), int(np.argmax(lens))} |
                 {int(i) for i in np.linspace(0, CONFIG4_READS - 1, CONFIG4_PARITY_READS - 2).astype(int)})
    _, probs = batching.infer_reads_dac(eng, [dacs[i] for i in idx], max_windows=max_windows, return_probs=True)
    n_match = n_tot = n_match_corrected = 0
    maxdp, worst_read, min_read_match = 0.0, None, 1.0
    for p, i in zip(probs, idx):
        xw, _pad = oracle.pad_and_window(oracle.normalize_raw_signal(dacs[i]))
        want = oracle.forward(xw, weights, np.float32)[:len(p)]
        m = (p >= 0.5) == (want >= 0.5)
        n_match += int(m.sum())
        n_tot += len(p)
        n_match_corrected += int(np.sum(np.asarray(oracle.correct_short(oracle.class_from_threshold(p))) ==
                                        np.asarray(oracle.correct_short(oracle.class_from_threshold(want)))))
        min_read_match = min(min_read_match, float(m.mean()))
        dp = float(np.abs(p - want).max())
        if dp > maxdp:
            maxdp, worst_read = dp, {"index": int(i), "length": int(lens[i])}
    eng.close()
    passed = bool(n_match / n_tot >= tol.CONFIG4_MIN_LABEL_MATCH and min_read_match >= tol.CONFIG4_MIN_LABEL_MATCH_PER_READ
                  and maxdp <= tol.CONFIG4_MAX_ABS_DP)
    total = int(lens.sum())
    windows = sum(int(p[0].shape[0]) for p in packed)
    value = total / dt
    roof = None
    if "gru_layer_mid" in kern:
        ms, n = kern["gru_layer_mid"]
        # the CIN = 128 mid layer over ALL buckets of one repetition: algorithmic flops of the un-padded samples
        mid_s = ms / rep * 1e-3
        ach = FLOP_PER_SAMPLE_GRU128 * total / mid_s / 1e12
        # counter traffic of THIS launch shape (131 072-window launches), per pass over the reads: a stored measurement
        tpath = os.path.join(ROOT, "profiles", "traffic_bf16_config4.json")
        t4 = json.load(open(tpath)) if os.path.exists(tpath) else None
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                "traffic": t4.get("gru_layer_mid_bytes_per_pass") if t4 else None,
                "traffic_source": None if t4 is None else {
                    "file": "profiles/traffic_bf16_config4.json", "commit": t4.get("commit"), "collected": t4.get("collected"),
                    "kind": "stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --only-leg config4`, summed over "
                            "the launches of one pass; not measured in this run",
                    "mid_layer_ms_per_pass_at_collection": t4.get("mid_layer_ms_per_pass"),
                    "mfma_busy_frac_at_collection": t4.get("mfma_busy_frac")},
                "kernel": "gru_bf16_pipe_kernel<128,false>", "launches_per_repetition": n // rep,
                "ms_per_repetition": ms / rep}
        roof = nearer_roof(roof, "bf16", total, mid_s)
    return {"workload": "configs[3]: %d reads, lengths LogUniform[512,16384] (seed 2), DAC squiggles, length-bucketed packed "
                        "launches of <= %d windows, bf16 biGRU arithmetic, device-resident" % (CONFIG4_READS, max_windows),
            "value": value if passed else None, "unit": "samples/s", "dtype": "bf16 (f32 accumulate)", "reads": CONFIG4_READS, "buckets": len(packed),
            "total_samples": total, "padding_overhead": windows * WINDOW / total - 1.0, "ms_per_pass": dt * 1e3,
            "whole_pass_frac_of_bf16_peak": value * FLOP_PER_SAMPLE / 1e12 / PEAK_BF16_MFMA_TFLOPS,
            "roofline": roof, "kernels_ms_per_pass": {k: v[0] / rep for k, v in kern.items()},
            "label_match_vs_fp32_oracle": n_match / n_tot, "max_abs_dp_vs_fp32_oracle": maxdp, "parity_sample": "%d samples of %d reads" % (n_tot, len(idx)),
            "parity": {"label_match_vs_fp32_oracle": n_match / n_tot, "label_match_after_correct_short": n_match_corrected / n_tot,
                       "min_label_match": tol.CONFIG4_MIN_LABEL_MATCH,
                       "lowest_label_match_of_a_read": min_read_match, "min_label_match_per_read": tol.CONFIG4_MIN_LABEL_MATCH_PER_READ,
                       "max_abs_dp_vs_fp32_oracle": maxdp, "gate": tol.CONFIG4_MAX_ABS_DP, "worst_read": worst_read,
                       "criterion": "oracle/tolerances.py (SURVEY 8d: label match against the fp32 oracle; the probability bound is secondary)",
                       "passed": passed},
            **({} if passed else {"unverified_value": value})}


def leg_config5(weights, local_rank, torch):
    """BASELINE configs[4] (SURVEY 8d "config 5"): the training step of RNN.train_network (rnn_class.py:201-210) --
    forward, loss, backward, Adam(1e-3), keep_prob 0.8 -- on balanced batches of uniform-label windows
    (ExampleDb.get_training_set's shape, networks/trainingDB/ExampleDb.py:50-83), the whole step on HIP kernels through the
    C ABI (catfish_amd/native_step.py).  ms/step and windows/s at 256 (the reference's batch) and 4096 windows, and the
    10-step loss trajectory with dropout off next to the torch-CPU trainer of the same graph."""
    from catfish_amd.training import Trainer
    out = {"what": "forward + sigmoid cross-entropy + backward + TF-rule Adam(lr 1e-3), dropout keep_prob 0.8 on the biGRU outputs; "
                   "synthetic config-2 windows, labels uniform per window, half positive; never the headline value"}
    pool = make_reads(40, seed=5).reshape(-1, WINDOW)
    rng = np.random.default_rng(0)
    dev = "cuda:%d" % local_rank
    for batch in (256, 4096):
        x = pool[rng.permutation(len(pool))[:batch]]
        y = np.repeat((np.arange(batch) % 2)[:, None], WINDOW, axis=1).astype(np.float32)
        tr = Trainer(weights, 3, 2, "Adam", 1e-3, keep_prob=0.8, device=dev, seed=0)
        # warm up BY TIME, then time ~0.4 s worth of steps: the leg before this one ends in host work (parity checks) and the GPU clocks
        # down meanwhile; five warm-up steps + a 30-step sample (round 5) timed the clock ramp -- batch_4096 read 2.86 ms in one run and
        # 3.73 ms in the next.  The torch-CPU checker of the loss trajectory runs AFTER both timings, not between them.
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.5:
            tr.train_step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_step(x, y)
        torch.cuda.synchronize()
        n = int(max(30, min(1000, 0.4 / ((time.perf_counter() - t0) / 10))))
        halves = []
        for _ in range(2):                                   # two equal halves: their spread says how steady the leg is
            t0 = time.perf_counter()
            for _ in range(n // 2):
                tr.train_step(x, y)
            torch.cuda.synchronize()
            halves.append((time.perf_counter() - t0) / (n // 2))
        dt = sum(halves) / 2
        # forward + backward of the same graph ~ 3x the forward's FLOPs (backward = dX and dW products of every forward product)
        flops = 3.0 * FLOP_PER_SAMPLE * WINDOW * batch
        ach = flops / dt / 1e12
        out["batch_%d" % batch] = {"ms_per_step": dt * 1e3, "windows_per_s": batch / dt, "native_step": bool(tr.step_impl is not None),
                                   "steps_timed": 2 * (n // 2), "ms_per_step_halves": [h * 1e3 for h in halves],
                                   "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None, "flop_per_step": flops,
                                                "what": "whole training step (not one kernel): 3 x forward FLOPs per window (389 504 x 35) / step "
                                                        "time / fp32 MFMA peak"
                                                        + ("; at the reference's batch of 256 windows the step is ~30 dependent launches of 16 "
                                                           "tiles each -- latency-bound, 16 tiles for 256 CUs" if batch == 256 else "")}}
        if batch == 256:
            x256, y256 = x, y
    gpu = Trainer(weights, 3, 2, "Adam", 1e-3, keep_prob=1.0, device=dev, seed=0)
    cpu = Trainer(weights, 3, 2, "Adam", 1e-3, keep_prob=1.0, device="cpu", seed=0)
    lg = [float(gpu.train_step(x256, y256)) for _ in range(10)]
    lc = [float(cpu.train_step(x256, y256)) for _ in range(10)]
    out["loss_10_steps_dropout_off"] = {"device": lg, "torch_cpu": lc, "batch": 256,
                                        "max_abs_diff": float(np.max(np.abs(np.array(lg) - np.array(lc))))}
    return out


def leg_latency(eng, torch, local_rank):
    """Small calls, fp32: one 4096-sample read (118 windows -- the reference's per-read ``model.infer``, infer.py:44) and a
    256-window micro-batch, device-resident in -> out with a synchronise per call, and the 118-window call with
    numpy in / numpy out (cf_infer_host: H2D and D2H inside)."""
    dev = torch.device("cuda", local_rank)
    out = {"what": "ms per call, fp32, one call in flight (synchronised after every call); never the headline value"}
    for n in (118, 256):
        x = torch.randn(n, WINDOW, device=dev)
        y = torch.empty(n * WINDOW, device=dev)
        for _ in range(10):
            eng.infer_device(x, out=y)
        torch.cuda.synchronize()
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.infer_device(x, out=y)
            torch.cuda.synchronize()
        out["%d_windows_ms" % n] = (time.perf_counter() - t0) / reps * 1e3
    xh = np.random.default_rng(0).normal(size=(118, WINDOW, 1))
    for _ in range(5):
        eng.infer_host(xh)
    t0 = time.perf_counter()
    for _ in range(50):
        eng.infer_host(xh)
    out["118_windows_host_numpy_in_out_ms"] = (time.perf_counter() - t0) / 50 * 1e3
    return out


def leg_sharded_gather(eng, weights, rank, world, dist, torch):
    """BASELINE configs[2]'s path, timed from the first submit to the gathered result on rank 0: in-memory reads (seed 1)
    sharded over the ranks in contiguous blocks, each rank streams ITS reads through its engine's ReadPipeline (int16 DAC up,
    normalise + forward + post-processing on device, run lists down), the ranks' span tables gathered on rank 0 over a gloo
    group as ONE flat ``SpanTable`` (arrays; the reference's per-read Python lists are built on demand, ``table.read(i)``).
    Weak scaling: 12 500 reads per rank (= 100 000 on 8)."""
    from catfish_amd import sharding
    n_total = SHARDED_READS_PER_RANK * world
    lengths = [READ_LEN] * n_total
    mine = sharding.shard_contiguous([sharding.windows_of(n) for n in lengths], world)[rank]
    reads = [None] * n_total
    for i in mine:                                        # every rank generates only its own shard
        reads[i] = squiggle_dac(np.random.default_rng([1, i]), READ_LEN)
    group = sharding.host_gather_group() if world > 1 else None
    batch = READS_PER_STEP * READ_LEN
    runner = sharding.EngineBatchRunner(eng, batch)
    # warm the pipeline (pinned buffers, first launches) outside the timed region
    list(runner.run([[reads[i] for i in mine[:READS_PER_STEP]]]))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table = sharding.infer_reads_sharded(eng, reads, lengths=lengths, max_samples_per_batch=batch, batch_runner=runner,
                                         rank=rank, world_size=world, gather_group=group, as_table=True)
    dt = time.perf_counter() - t0                          # rank 0: includes the gather of every rank's results
    RANK_NOTES["sharded_gather_s"] = dt
    t1 = time.perf_counter()
    as_lists = table.expand() if rank == 0 else None      # the reference's result type, built once on rank 0 (timed apart)
    dt_lists = time.perf_counter() - t1
    if world > 1:
        dist.barrier()
    if rank != 0:
        return None
    from oracle import catfish_oracle as oracle
    ok = len(table) == n_total and bool((table.lengths == READ_LEN).all()) and len(as_lists) == n_total
    for i in mine[:2]:                                     # untimed check of two reads against the oracle
        spans, n, _ = oracle.infer_read(oracle.normalize_raw_signal(reads[i]), weights, np.float32)
        ok = ok and table.read(i) == (spans, n) == as_lists[i]
    return {"workload": "configs[2]: %d reads x %d samples (seed 1) sharded over %d rank(s), %d per rank; host gather on rank 0"
                        % (n_total, READ_LEN, world, SHARDED_READS_PER_RANK),
            "value": n_total * READ_LEN / dt, "unit": "samples/s", "seconds": dt, "reads": n_total, "n_gpus": world,
            "results_ok": bool(ok), "spans_found": int(table.start.shape[0]),
            "python_lists_on_rank0_seconds": dt_lists, "value_with_python_lists": n_total * READ_LEN / (dt + dt_lists),
            "what": "pinned int16 DAC -> per-rank ReadPipeline (cf_normalize, cf_infer, cf_postprocess, cf_spans) -> per-rank span "
                    "tables -> gloo gather_object on rank 0 as one flat table; PCIe and the gather included; the reference's per-read "
                    "Python span lists (its result type) are built afterwards on rank 0 and timed apart; never the headline value"}


def leg_cli_end_to_end(weights, rank, world, local_rank, dist, torch):
    """The product entry point on BASELINE configs[2]'s shape, files -> JSON: ``catfish_amd.cli.run_pipeline`` (the body of
    catfish/catfish:23-94, split step included but timed apart) over a directory of 12 500 x 4096-sample int16 reads per rank.  Every rank
    loads, classifies AND merges / centres / complements its own files (catfish/catfish:50-82), formats its part of the two
    JSON documents and writes it at its offset.  Timed: everything but the network load -- output directories, the listing of the
    input directory (catfish/catfish:49-50; it grows with the job), classification, merge tail, documents on disk; page cache
    warm (the files were just written)."""
    import contextlib
    import io
    import shutil
    import tempfile
    from catfish_amd import checkpoint, cli, sharding
    n_total = SHARDED_READS_PER_RANK * world
    box = [tempfile.mkdtemp(prefix="catfish_bench_") if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    root = box[0]
    try:
        if rank == 0:
            os.makedirs(os.path.join(root, "reads"))
            os.makedirs(os.path.join(root, "ResNetRNN", "checkpoints"))
            with open(os.path.join(root, "ResNetRNN", "ResNetRNN.txt"), "w") as fh:
                fh.write("MODEL TYPE: ResNet-RNN\n\nbatch_size: 256\noptimizer_choice: RMSProp\nlearning_rate: 0.001\n"
                         "layer_size: 64\nn_layers: 3\nkeep_prob: 0.8\nlayer_size_res: 32\nn_layers_res: 2\n")
            checkpoint.write_checkpoint(os.path.join(root, "ResNetRNN", "checkpoints", "ckpnt-30000"), weights)
        if world > 1:
            dist.barrier()
        for i in range(rank, n_total, world):                 # every rank writes a share of the directory
            np.save(os.path.join(root, "reads", "read_%06d.npy" % i), squiggle_dac(np.random.default_rng([1, i]), READ_LEN))
        if world > 1:
            dist.barrier()
        timings = {}
        host_group = sharding.host_gather_group() if world > 1 else None
        sink = io.StringIO()
        t0 = time.perf_counter()
        failure = None
        try:
            with contextlib.redirect_stdout(sink):            # the pipeline prints the reference's progress lines
                res = cli.run_pipeline(os.path.join(root, "reads"), os.path.join(root, "out"), chunk_size=1000,
                                       network_path=os.path.join(root, "ResNetRNN"), device=local_rank, timings=timings)
        except Exception as exc:                              # noqa: BLE001 -- raised below, after the barrier every rank reaches
            failure = exc
        total = time.perf_counter() - t0
        per_rank = [None] * world
        mine = {k: timings.get(k) for k in ("listing_s", "model_s", "infer_s", "chunks_s", "write_s", "split_s")}
        mine.update(rank=rank, total_s=total, failed=None if failure is None else "%s: %s" % (type(failure).__name__, failure),
                    split=res.get("split") if failure is None else None)
        if world > 1:
            dist.all_gather_object(per_rank, mine, group=host_group)      # doubles as the barrier every rank reaches
        else:
            per_rank = [mine]
        RANK_NOTES.setdefault("cli_end_to_end", {}).update(mine)
        if failure is not None:
            raise failure
        if rank != 0:
            return None
        # the network load is set-up; the directory listing (names on every rank, sizes of a rank's block, the agreement) is part
        # of the job -- it grows with the number of files -- and is counted.  The split step (every rank writes the int16 slices of its
        # reads) is file writing after the documents: timed apart (``split_s``), reported beside ``value``, not inside it
        dt = total - timings["model_s"] - timings["split_s"]
        post = timings.get("write_s", 0.0)
        with open(os.path.join(root, "out", "TEMP", "hp_positions.json")) as fh:
            hp = json.load(fh)
        with open(os.path.join(root, "out", "TEMP", "nonhp_positions.json")) as fh:
            nonhp = json.load(fh)
        ok = (res["reads"] == n_total == len(nonhp) and res["samples"] == n_total * READ_LEN
              and len(hp) == res["reads_with_hp"] and sum(len(v) for v in hp.values()) == res["hp_chunks"])
        # untimed CONTENT check (catfish/catfish:55-82, 85-92): the documents' entries and the split files of the first, a middle and
        # the last file of rank 0's block against the oracle's classification + the per-read Python rules + numpy slicing
        from oracle import catfish_oracle as oracle
        lo, hi = res["file_range"]
        checked = []
        for k in sorted({lo, (lo + hi) // 2, hi - 1}) if hi > lo else []:
            name = res["files"][k - lo]
            dac = squiggle_dac(np.random.default_rng([1, int(name[len("read_"):-len(".npy")])]), READ_LEN)
            spans, length, _ = oracle.infer_read(oracle.normalize_raw_signal(dac), weights, np.float32)
            merged, non = cli.chunks_of_read([list(sp) for sp in spans], length, 1000)
            same = hp.get(name) == merged and nonhp.get(name) == json.loads(json.dumps(non))
            for j, (s0, s1) in enumerate((merged or []) + (non if merged else [])):
                part = os.path.join(root, "out", "TEMP", "HP" if j < len(merged) else "nonHP", "%s_%d.npy" % (name.split(".")[0], j))
                same = same and os.path.exists(part) and np.array_equal(np.load(part), dac[s0:s1])
            checked.append({"file": name, "hp_chunks": 0 if merged is None else len(merged), "matches_oracle": bool(same)})
            ok = ok and same
        ok = ok and len(checked) > 0
        split_counts = [p.get("split") for p in per_rank]
        return {"workload": "configs[2] through the CLI: %d files x %d samples (seed 1) over %d rank(s), %d per rank; chunk_size "
                            "1000" % (n_total, READ_LEN, world, SHARDED_READS_PER_RANK),
                "value": n_total * READ_LEN / dt if ok else None, "unit": "samples/s", "seconds": dt,
                "model_load_seconds_excluded": timings["model_s"], "listing_s": timings["listing_s"],
                "split_s": max(p["split_s"] for p in per_rank), "split_seconds_excluded": timings["split_s"],
                "split": {"files_hp": sum(c["files_hp"] for c in split_counts), "files_nonhp": sum(c["files_nonhp"] for c in split_counts),
                          "reads": sum(c["reads"] for c in split_counts), "samples": sum(c["samples"] for c in split_counts),
                          "what": "catfish/split_f5.py:8-81 for int16 .npy reads: signal[s0:s1] of every chunk as HP/<stem>_<k>.npy or "
                                  "nonHP/<stem>_<k>.npy, every rank its own reads; slowest rank's seconds; beside value, not inside it "
                                  "(the reference writes gzip-9 HDF5 copies here -- not a comparison)"},
                "content_checked": checked,
                "timed_region": "output directories + listing of the input directory (every rank: names; its block: sizes; the ranks' "
                                "agreement) + classification + merge tail + documents on disk; the network load and the split step "
                                "(split_s) are left out",
                "rank0": {"listing_s": timings.get("listing_s"), "infer_s": timings.get("infer_s"), "chunks_s": timings.get("chunks_s"),
                          "write_s": timings.get("write_s")},
                "per_rank": per_rank,
                "rank0_after_classification_frac": post / dt, "reads": n_total, "reads_with_hp": len(hp),
                "hp_chunks": res["hp_chunks"], "document_bytes": res["bytes"], "n_gpus": world, "results_ok": bool(ok),
                "what": "int16 .npy files -> per-rank loader thread + ReadPipeline (cf_normalize, cf_infer, cf_postprocess, cf_spans) "
                        "-> per-rank cf_chunks_from_spans (merge, center_hp, complement) -> per-rank cf_chunks_json, every rank writes its "
                        "part of the two documents at its offset (no result gather); never the headline value"}
    finally:
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)


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
                    help="skip the informational legs (other precisions, config 4, host-to-host pipeline)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the informational sharded_gather leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reads-per-rank", type=int, default=SHARDED_READS_PER_RANK,
                    help="reads (files) per rank of the sharded_gather / cli_end_to_end legs (BASELINE configs[2]: 12 500 = 100 000 on 8)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="run untimed steps for this long before the W warm-up steps: the GPU clocks down while the CPU "
                         "baseline leg (or process start-up) keeps it idle, and a short W would time the clock ramp")
    ap.add_argument("--no-kernel-events", action="store_true", help="disable per-kernel HIP events")
    ap.add_argument("--only-leg", choices=["config4"], default=None,
                    help="run ONE informational leg in its profiling form and print its JSON (the command tools/collect_traffic_config4.sh "
                         "puts under rocprofv3); no headline line")
    args = ap.parse_args()
    globals()["SHARDED_READS_PER_RANK"] = int(args.reads_per_rank)
    if args.only_leg == "config4":
        import torch
        torch.cuda.set_device(0)
        print(json.dumps(leg_config4(load_weights(), 0, torch, profile_only=True)))
        return

    if os.environ.get("CATFISH_BENCH_WATCHDOG_S"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CATFISH_BENCH_WATCHDOG_S"]), repeat=True, file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launcher_local_rank = local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared_device = None
    if "CATFISH_BENCH_DEVICE" in os.environ:      # rehearsal of the N > 1 path on a box with fewer GPUs than ranks
        shared_device = local_rank = int(os.environ["CATFISH_BENCH_DEVICE"])
    cpu_res = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress("cpu_baseline leg (oracle on the host cores)")
        cpu_res = cpu_baseline()

    # Rank placement BEFORE the first GPU call and the first pinned allocation: this rank (and the loader thread, the library's
    # file pool and the HIP runtime's threads it will start) runs on the CPUs next to its MI355X -- catfish_amd/placement.py.
    # After the CPU baseline, which is entitled to every core of the box.
    from catfish_amd import placement
    import torch                                     # (importing it touches no GPU; nor does counting the devices on this image)
    import torch.distributed as dist
    n_visible = torch.cuda.device_count()
    if shared_device is None and 0 < n_visible <= launcher_local_rank:
        # fewer visible cards than ranks on this node (a launcher that narrows HIP_VISIBLE_DEVICES per rank, or a box with fewer GPUs):
        # ranks wrap around the cards there are.  Whether that is N cards or a rehearsal is decided from the cards' identities below.
        local_rank = launcher_local_rank % n_visible
        sys.stderr.write("bench.py: rank %d: %d visible device(s) for local rank %d -> device %d\n" % (rank, n_visible, launcher_local_rank, local_rank))
    device_of_rank = (lambda r: shared_device) if shared_device is not None else ((lambda r: r % n_visible) if n_visible > 0 else None)
    place = placement.bind(launcher_local_rank, local_world, device_of_rank=device_of_rank)

    progress("bound to CPUs %s (%s)" % (placement.format_cpulist(place.get("cpus") or []), place.get("source")))
    # The rank now owns fewer CPUs than the machine has, but OpenMP sized its pool when it was first loaded (numpy, above): a
    # pool of 256 spinning threads on 128 CPUs turned the torch-CPU CHECKER of the config5 leg from 3 s into 266 s (measured,
    # gpurun_out/round5 first run).  One thread per owned CPU at most; the product path runs no torch CPU op at all.
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(place.get("cpus") or [1]), 64)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    backend, nccl_error, ranks_seen, rccl_probe = None, None, 1, None
    if world > 1:
        # The data path has no collective, so the clock does not depend on RCCL either: the process group that carries the timing
        # barrier and the MAX over ranks is the HOST group (gloo), always.  RCCL is brought up beside it as a PROBE -- one all-reduce
        # of a device scalar over a second group with a short timeout -- so that the line shows whether the node's RCCL saw all N
        # ranks, and a stuck or failing bring-up costs at most RCCL_PROBE_TIMEOUT_S and a note (``collective.rccl_probe``), never the run.
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("CATFISH_DIST_TIMEOUT_S", "1800"))))
        probe = torch.ones(1)
        dist.all_reduce(probe)
        backend, ranks_seen = "gloo", int(round(float(probe.item())))
        rccl_probe = probe_rccl(dist, torch, local_rank)
        nccl_error = rccl_probe.get("error")

    progress("process group %s; creating the engine on device %d" % (backend, local_rank))
    from catfish_amd.engine import HipEngine
    weights = load_weights()
    eng = HipEngine(weights, device=local_rank, max_windows_per_pass=READS_PER_STEP * 118, n_streams=args.streams,
                    precision=args.precision)
    pci_bus_id, uuid = eng.device_identity()
    progress("engine on %s; generating %d reads" % (pci_bus_id, args.pool_reads))
    placement.verify(place, pci_bus_id, launcher_local_rank, local_world)     # the card the runtime gave us against the sysfs guess

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

    progress("warm-up and the timed steps")
    warm_by_time(eng, batches, outs, args.prewarm_ms * 1e-3, torch)      # clock warm-up, outside the W + K steps
    for i in range(args.warmup):
        eng.infer_device(batches[i % n_batches], out=outs[i & 1])
    torch.cuda.synchronize()

    # the K timed steps: no instrumentation inside the clock (per-kernel HIP events are collected in a pass of their own below)
    eng.profile_enable(False)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.infer_device(batches[i % n_batches], out=outs[i & 1])
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    eng.check_error()
    # ... then an equally long UNTIMED pass over the same batches with HIP events around every kernel of every step (on the
    # streams the kernels are launched on): ``kernels_ms`` and ``roofline.avg_launch_ms`` come from this pass, ``value`` from the one above
    kern, dt_events = {}, None
    if not args.no_kernel_events:
        eng.profile_enable(True, every=1)
        eng.profile_reset()
        dt_events = timed_steps(eng, batches, outs, args.steps, torch)
        kern = eng.profile_read()
        eng.profile_enable(False)
        eng.check_error()

    dt_local = dt
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)          # host group (gloo): the clock does not depend on RCCL
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    samples_per_step = READS_PER_STEP * READ_LEN          # un-padded, per rank
    value = world * args.steps * samples_per_step / dt

    result = None
    parity_ok = True
    progress("timed region done: %.3f ms per step; parity gate" % (dt / args.steps * 1e3))
    if rank == 0:
        # parity gate of the benchmarked configuration against the oracle (not timed): a build that fails it reports no value
        from oracle import catfish_oracle as oracle
        # the first, a middle and the last read of the launch (tile 0, the interior, the ragged tail of the last workgroup); the
        # launch keeps the benchmark's size (rocprof averages stay comparable)
        picks = [0, READS_PER_STEP // 2, READS_PER_STEP - 1]
        chk = reads[picks]
        got = eng.infer_device(batches[0], out=outs[0]).cpu().numpy().reshape(READS_PER_STEP, 118 * WINDOW)[picks].reshape(-1).astype(np.float64)
        want64 = oracle.forward(chk.reshape(-1, WINDOW), weights, np.float64)
        want32 = oracle.forward(chk.reshape(-1, WINDOW), weights, np.float32)
        max_dp = float(np.abs(got - want64).max())
        match = float(np.mean((got >= 0.5) == (want32 >= 0.5)))
        # ... and after correct_short (SURVEY 8d asks for both): per read, on its real samples (infer.py:47 trims the padding first)
        g3, w3 = got.reshape(len(picks), -1)[:, :READ_LEN], want32.reshape(len(picks), -1)[:, :READ_LEN]
        match_corrected = float(np.mean([np.mean(np.asarray(oracle.correct_short(oracle.class_from_threshold(g))) ==
                                                 np.asarray(oracle.correct_short(oracle.class_from_threshold(w)))) for g, w in zip(g3, w3)]))
        from oracle import tolerances as tol
        gate = tol.CONFIG4_MAX_ABS_DP if args.precision == "bf16" else tol.GATE_MAX_ABS_DP
        min_match = tol.CONFIG4_MIN_LABEL_MATCH if args.precision == "bf16" else 1.0
        parity_ok = bool(np.isfinite(got).all() and max_dp < gate and match >= min_match)

        roof = mid_roofline(kern, args.precision, samples_per_step)
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        result = {
            "metric": "signal samples/s classified",
            "value": value if parity_ok else None, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "prewarm_ms": args.prewarm_ms, "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (split-operand fp32 emulation, f32 accumulate)",
                      "bf16": "bf16 (f32 accumulate)"}[args.precision], "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 4096-sample reads, 256 reads (30208 windows of 35) "
                                   "per step and GPU, fp32, ckpnt-30000 weights",
                       "reads_per_step": READS_PER_STEP, "read_len": READ_LEN, "windows_per_step": READS_PER_STEP * 118,
                       "parallelism": "reads sharded over %d GPU(s), no collective" % world},
            "roofline": roof,
            "whole_pass": {"achieved_tflops": value / world * FLOP_PER_SAMPLE / 1e12,
                           "frac_of_mfma_peak": value / world * FLOP_PER_SAMPLE / 1e12 / peak},
            "kernels_ms": {k: v[0] / v[1] for k, v in kern.items()},
            "kernel_events": {"pass": "separate untimed pass of %d steps over the same batches right after the timed steps, HIP events "
                                      "around every kernel of every step; the timed steps run with events off" % args.steps,
                              "ms_per_step_with_events": None if dt_events is None else dt_events / args.steps * 1e3,
                              "launches_timed": {k: v[1] for k, v in kern.items()}},
            "parity": {"max_abs_dp_vs_fp64_oracle": max_dp, "label_match_vs_fp32_oracle": match,
                       "label_match_after_correct_short": match_corrected, "gate": gate,
                       "min_label_match": min_match, "passed": parity_ok,
                       "sample": "reads %s of the %d in one launch (first, middle, last)" % (picks, READS_PER_STEP)},
        }
        if not parity_ok:
            result["unverified_value"] = value
        result["cpu_baseline"] = cpu_res
    # informational legs: the same workload with the other GRU arithmetics (not the headline value)
    if world == 1 and not args.no_extra_precisions:
        from oracle import catfish_oracle as oracle
        extra = {}
        for prec in ("fp32", "bf16x3", "bf16"):
            if prec == args.precision:
                continue
            progress("other_precisions leg: %s" % prec)
            e2 = HipEngine(weights, device=local_rank, max_windows_per_pass=READS_PER_STEP * 118, precision=prec)
            warm_by_time(e2, batches, outs, 0.5, torch)      # the GPU clocks down while the CPU oracle ran
            n2 = max(5, args.steps // 2)
            e2.profile_enable(True, every=2)
            e2.profile_reset()
            d2 = timed_steps(e2, batches, outs, n2, torch)
            k2 = e2.profile_read()
            e2.profile_enable(False)
            got = e2.infer_device(batches[0], out=outs[0]).cpu().numpy()[:118 * WINDOW].astype(np.float64)
            want = oracle.forward(reads[0], weights, np.float64)
            v2 = n2 * samples_per_step / d2
            extra[prec] = {"value": v2, "unit": "samples/s", "ms_per_step": d2 / n2 * 1e3,
                           "max_abs_dp_vs_fp64_oracle": float(np.abs(got - want).max()),
                           "label_match_vs_fp64_oracle": float(np.mean((got >= 0.5) == (want >= 0.5))),
                           "roofline": mid_roofline(k2, prec, samples_per_step),
                           "whole_pass_frac_of_mfma_peak": v2 * FLOP_PER_SAMPLE / 1e12 /
                           (PEAK_F32_MFMA_TFLOPS if prec == "fp32" else PEAK_BF16_MFMA_TFLOPS),
                           "kernels_ms": {k: v[0] / v[1] for k, v in k2.items()}}
            e2.close()
        result["other_precisions"] = extra
        progress("config4 leg (10 000 variable-length reads, bf16)")
        result["config4"] = leg_config4(weights, local_rank, torch)
        for name, leg in (("latency", lambda: leg_latency(eng, torch, local_rank)),
                          ("config5", lambda: leg_config5(weights, local_rank, torch))):
            progress("%s leg" % name)
            try:
                result[name] = leg()
            except Exception as exc:      # informational legs: never lose the headline line to them
                result[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        # informational: host-to-host rate of the streaming pipeline (pinned int16 DAC in, spans out, PCIe inclusive)
        progress("host_to_host_pipeline leg")
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
    if not args.no_sharded_leg and args.precision == "fp32":
        progress("sharded_gather leg (%d reads per rank)" % SHARDED_READS_PER_RANK)
        try:
            sg = leg_sharded_gather(eng, weights, rank, world, dist, torch)
        except Exception as exc:      # informational leg: never lose the headline line to it
            sg = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if world > 1:
                sys.stderr.write("bench.py: sharded_gather leg failed on rank %d: %s\n" % (rank, exc))
        if rank == 0:
            result["sharded_gather"] = sg
    eng.close()
    if not args.no_sharded_leg and args.precision == "fp32":
        progress("cli_end_to_end leg (%d files per rank)" % SHARDED_READS_PER_RANK)
        try:
            ce = leg_cli_end_to_end(weights, rank, world, local_rank, dist, torch)
        except Exception as exc:      # informational leg: never lose the headline line to it
            ce = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if world > 1:
                sys.stderr.write("bench.py: cli_end_to_end leg failed on rank %d: %s\n" % (rank, exc))
        if rank == 0:
            result["cli_end_to_end"] = ce
    # Who ran what where: every rank's identity, placement and times, gathered over the HOST group (gloo) and printed by rank 0,
    # so that a scaling record can be judged from the line alone -- did the collective see N ranks, did every rank drive its own
    # card, which rank was the slow one.  Ranks that SHARE a card are a rehearsal of the launch path, not a measurement.
    progress("legs done; gathering the ranks' records")
    import socket
    me = {"rank": rank, "local_rank": launcher_local_rank, "device_index": local_rank, "pci_bus_id": pci_bus_id, "uuid": uuid,
          "device_name": torch.cuda.get_device_name(local_rank), "host": socket.gethostname(), "pid": os.getpid(),
          "dt_s": dt_local, "ms_per_step": dt_local / args.steps * 1e3, "samples_per_s": args.steps * samples_per_step / dt_local,
          "cpus": placement.format_cpulist(place.get("cpus") or []), "placement": placement.summary(place)}
    me.update(RANK_NOTES)
    ranks = [me]
    if world > 1:
        from catfish_amd import sharding
        ranks = [None] * world
        dist.all_gather_object(ranks, me, group=sharding.host_gather_group())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        describe_ranks(result, ranks, world, backend, nccl_error, ranks_seen, torch.cuda.device_count(), rccl_probe)
        print(json.dumps(result))
        if not parity_ok:
            sys.stderr.write("bench.py: PARITY GATE FAILED (max |dp| %.3g, label match %.5f): no value reported\n" % (max_dp, match))
            sys.exit(3)


if __name__ == "__main__":
    main()
