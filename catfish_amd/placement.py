"""Rank placement: bind a rank -- and with it its loader thread, the library's file-reading pool and the first touch of its
pinned staging buffers -- to the host CPUs that sit next to its MI355X.

The reference's per-file loop (catfish/catfish:50-82) shards over one process per GPU; what grows with the job on the host side
is reading files, staging them in pinned memory and the merge tail (SURVEY.md 8e names the host side and PCIe, not xGMI, as the
scaling limit).  On a two-socket 8-GPU node a rank that runs on the far socket pays the inter-socket hop on every staged byte, so
every rank is bound IN PROCESS (``os.sched_setaffinity``: no numactl / taskset wrapper, no re-exec) before it touches the GPU
or allocates pinned memory.  Threads created afterwards (``sharding._prefetched``'s loader, ``cf_load_npy_int16``'s pool, the
HIP runtime's own) inherit the mask.

Where a rank's CPUs come from, first that yields anything inside the process's current mask:

  1. ``pci``    ``/sys/bus/pci/devices/<bdf>/local_cpulist`` of the GPU the rank drives;
  2. ``numa``   ``/sys/bus/pci/devices/<bdf>/numa_node`` -> ``/sys/devices/system/node/node<N>/cpulist``;
  3. ``split``  an even split of the current affinity mask over the ranks of the node by LOCAL_RANK.

The GPU's PCI address is read WITHOUT a HIP call from the KFD topology (``/sys/class/kfd/kfd/topology/nodes/*/properties``:
``domain`` / ``location_id`` of the nodes with ``simd_count > 0``, in node order = HIP's device order, filtered by
``ROCR_VISIBLE_DEVICES`` / ``HIP_VISIBLE_DEVICES`` when those are plain index lists); ``verify`` compares it with what the HIP
runtime reports once the device is open and re-binds when they differ.  Ranks whose GPUs share one locality set (four GPUs of a
socket; or several ranks rehearsing on one card) cut it into disjoint slices, whole physical cores first, in LOCAL_RANK order:
every rank can work that out alone from sysfs, no exchange needed.  ``CATFISH_BIND=0`` turns all of it off.
"""
from __future__ import annotations

import os

SYSFS = "/sys"
MIN_CPUS_PER_RANK = 2      # a slice smaller than this (main thread + loader) is not worth the isolation: share the locality set


def parse_cpulist(text):
    """``"0-3,8,10-11"`` -> [0, 1, 2, 3, 8, 10, 11] (the kernel's list format; empty / malformed pieces are skipped)."""
    out = set()
    for piece in (text or "").replace("\n", "").split(","):
        piece = piece.strip()
        if not piece:
            continue
        lo, _, hi = piece.partition("-")
        try:
            a = int(lo)
            b = int(hi) if hi else a
        except ValueError:
            continue
        out.update(range(a, b + 1))
    return sorted(out)


def format_cpulist(cpus):
    """Inverse of ``parse_cpulist``: [0, 1, 2, 3, 8] -> ``"0-3,8"``."""
    cpus = sorted(set(int(c) for c in cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(runs)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def _visible_filter():
    """The index list of ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES (ROCR filters first, HIP indexes into what is left), or
    None per variable when it is unset or not a plain list of integers (UUID forms: the KFD guess is then left to ``verify``)."""
    out = []
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        raw = os.environ.get(var)
        if raw is None or raw.strip() == "":
            out.append(None)
            continue
        try:
            out.append([int(p) for p in raw.split(",") if p.strip() != ""])
        except ValueError:
            out.append(None)
    return out


def kfd_gpu_bdfs(sysfs=None):
    """PCI addresses (``dddd:bb:dd.f``) of the GPUs in KFD topology order, no HIP call.  [] when the topology is not there."""
    sysfs = SYSFS if sysfs is None else sysfs
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        text = _read(os.path.join(base, str(n), "properties"))
        if not text:
            continue
        props = {}
        for line in text.splitlines():
            key, _, val = line.partition(" ")
            try:
                props[key] = int(val)
            except ValueError:
                pass
        if props.get("simd_count", 0) <= 0:
            continue                                    # a CPU node
        loc = props.get("location_id", 0)
        out.append("%04x:%02x:%02x.%x" % (props.get("domain", 0) & 0xFFFF, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7))
    return out


def device_bdf(device_index, sysfs=None):
    """The PCI address of HIP device ``device_index`` as the KFD topology and the visibility variables predict it, or None."""
    bdfs = kfd_gpu_bdfs(sysfs)
    for flt in _visible_filter():
        if flt is not None:
            bdfs = [bdfs[i] for i in flt if 0 <= i < len(bdfs)]
    return bdfs[device_index] if 0 <= device_index < len(bdfs) else None


def normalize_bdf(text):
    """``"0000:C5:00.0"`` / ``"c5:00.0"`` -> ``"0000:c5:00.0"`` (what sysfs names the directory); None for anything else."""
    if not text:
        return None
    t = text.strip().lower()
    if t.count(":") == 1:
        t = "0000:" + t
    parts = t.replace(".", ":").split(":")
    if len(parts) != 4:
        return None
    try:
        return "%04x:%02x:%02x.%x" % tuple(int(p, 16) for p in parts)
    except ValueError:
        return None


def locality_cpus(bdf, allowed, sysfs=None):
    """-> (source, cpus): the CPUs next to PCI device ``bdf`` that this process may run on; ("none", []) when sysfs does not
    say (containers often hide it) or when none of them is inside ``allowed``."""
    sysfs = SYSFS if sysfs is None else sysfs
    allowed = set(allowed)
    if bdf:
        dev = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        cpus = [c for c in parse_cpulist(_read(os.path.join(dev, "local_cpulist"))) if c in allowed]
        if cpus:
            return "pci", cpus
        node = _read(os.path.join(dev, "numa_node"))
        try:
            node = int(node.strip()) if node is not None else -1
        except ValueError:
            node = -1
        if node >= 0:
            cpus = [c for c in parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")))
                    if c in allowed]
            if cpus:
                return "numa", cpus
    return "none", []


def _core_order(cpus, sysfs=None):
    """``cpus`` ordered so that the hardware threads of one physical core sit next to each other (cores by their lowest CPU
    number): a slice of it takes whole cores, never one rank's SMT sibling of another rank's core.  Plain order when the
    topology files are missing."""
    sysfs = SYSFS if sysfs is None else sysfs
    key = {}
    for c in cpus:
        sib = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology", "thread_siblings_list")))
        key[c] = (min(sib) if sib else c, c)
    return sorted(cpus, key=lambda c: key[c])


def slice_for(position, parts, cpus, sysfs=None):
    """The ``position``-th of ``parts`` near-equal, disjoint slices of ``cpus`` (whole cores first).  The whole set when a
    slice would be smaller than MIN_CPUS_PER_RANK."""
    cpus = _core_order(sorted(set(cpus)), sysfs)
    if parts <= 1 or len(cpus) // parts < MIN_CPUS_PER_RANK:
        return sorted(cpus)
    lo = position * len(cpus) // parts
    hi = (position + 1) * len(cpus) // parts
    return sorted(cpus[lo:hi])


def plan(local_rank, local_world, device_of_rank=None, allowed=None, sysfs=None, bdf_of_device=None):
    """Work out, without touching the GPU, the CPUs of rank ``local_rank`` of ``local_world`` ranks on this node.

    ``device_of_rank``  function local rank -> HIP device index (default: identity -- rank r drives device r);
    ``allowed``         the CPUs the process may use now (default ``os.sched_getaffinity(0)``);
    ``bdf_of_device``   function device index -> PCI address (default ``device_bdf``: the KFD topology).
    -> dict(cpus, source "pci" | "numa" | "split", bdf, device, shared_with = local ranks that have the same locality set)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(set(int(c) for c in allowed))
    device_of_rank = (lambda r: r) if device_of_rank is None else device_of_rank
    bdf_of_device = (lambda d: device_bdf(d, sysfs)) if bdf_of_device is None else bdf_of_device
    local_world = max(1, int(local_world))
    sets = []
    for r in range(local_world):
        bdf = bdf_of_device(device_of_rank(r))
        source, cpus = locality_cpus(bdf, allowed, sysfs)
        sets.append((source, tuple(cpus), bdf))
    source, cpus, bdf = sets[local_rank]
    if source == "none":
        # no locality information for this rank's GPU: an even split of what is left of the mask once the ranks that DO have a
        # locality set took theirs (or of the whole mask when nobody has one)
        taken = set(c for s, cs, _b in sets if s != "none" for c in cs)
        pool = [c for c in allowed if c not in taken] or allowed
        peers = [r for r in range(local_world) if sets[r][0] == "none"]
        return {"cpus": slice_for(peers.index(local_rank), len(peers), pool, sysfs), "source": "split", "bdf": bdf,
                "device": device_of_rank(local_rank), "shared_with": peers}
    peers = [r for r in range(local_world) if sets[r][0] != "none" and sets[r][1] == cpus]
    return {"cpus": slice_for(peers.index(local_rank), len(peers), cpus, sysfs), "source": source, "bdf": bdf,
            "device": device_of_rank(local_rank), "shared_with": peers}


def enabled():
    return os.environ.get("CATFISH_BIND", "1") != "0"


_APPLIED = None      # the placement this process already took: a second bind() must not slice the slice


def current():
    """The placement this process bound itself to (``bind``'s record), or None."""
    return _APPLIED


def bind(local_rank=None, local_world=None, device_of_rank=None, sysfs=None, bdf_of_device=None):
    """Bind the calling thread -- call it from the main thread at start-up -- and with it every thread started afterwards (they inherit
    the mask: the loader helper, the library's file pool, the HIP runtime's own) to the rank's CPUs.  Threads that already exist
    (a BLAS pool created at numpy's import) keep their masks; nothing on the product path runs on them.  Call BEFORE the first GPU
    call and before any pinned allocation.  -> the ``plan`` dict plus ``bound`` (False when ``CATFISH_BIND=0``, when the platform has no
    ``sched_setaffinity`` or when the call was refused), ``mask_before``, and ``error`` when refused.  Never raises: a job that
    cannot be pinned still runs.  Once per process: a later call returns the first one's record (bench.py binds at start-up and
    then calls ``cli.run_pipeline``, which would otherwise cut its slice of the mask into slices again)."""
    global _APPLIED
    if _APPLIED is not None:
        return _APPLIED
    _APPLIED = _bind(local_rank, local_world, device_of_rank, sysfs, bdf_of_device)
    return _APPLIED


def _bind(local_rank, local_world, device_of_rank, sysfs, bdf_of_device):
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "source": "unsupported", "cpus": [], "device": None, "bdf": None}
    before = sorted(os.sched_getaffinity(0))
    if not enabled():
        return {"bound": False, "source": "off (CATFISH_BIND=0)", "cpus": before, "mask_before": before, "device": None, "bdf": None}
    try:
        p = plan(local_rank, local_world, device_of_rank, allowed=before, sysfs=sysfs, bdf_of_device=bdf_of_device)
    except Exception as exc:                                   # noqa: BLE001 -- placement is an optimisation, never a reason to fail
        return {"bound": False, "source": "error", "error": "%s: %s" % (type(exc).__name__, exc), "cpus": before,
                "mask_before": before, "device": None, "bdf": None}
    p["mask_before"] = before
    try:
        os.sched_setaffinity(0, p["cpus"])
        p["bound"] = True
    except OSError as exc:
        p["bound"] = False
        p["error"] = "%s: %s" % (type(exc).__name__, exc)
    return p


def verify(placement, actual_bdf, local_rank=None, local_world=None, device_of_rank=None, sysfs=None):
    """After the device is open: ``actual_bdf`` is what the HIP runtime says the rank's GPU is.  When the KFD guess was another
    device (or there was none) and sysfs knows the real one's neighbourhood, bind again -- later than ideal (pinned buffers
    allocated in between stay where they are), which the record says: ``rebound_after_gpu_init``."""
    actual = normalize_bdf(actual_bdf)
    placement["bdf_from_runtime"] = actual
    if not placement.get("bound") or actual is None or actual == placement.get("bdf"):
        placement["bdf_agrees"] = actual is None or actual == placement.get("bdf")
        return placement
    placement["bdf_agrees"] = False
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    allowed = placement.get("mask_before") or sorted(os.sched_getaffinity(0))
    source, cpus = locality_cpus(actual, allowed, sysfs)
    if source == "none":
        return placement                                           # nothing better known: the split stays
    # the other ranks' real devices are unknown here (no exchange): take their PLANNED devices, as ``plan`` does -- the ranks sharing this
    # locality set are this rank plus those whose planned set equals it, and the set is cut among them (4 of 8 ranks per socket: a quarter
    # of the socket each, not an eighth)
    device_of_rank = (lambda r: r) if device_of_rank is None else device_of_rank
    peers = [r for r in range(max(1, int(local_world)))
             if r == local_rank or tuple(locality_cpus(device_bdf(device_of_rank(r), sysfs), allowed, sysfs)[1]) == tuple(cpus)]
    new = slice_for(peers.index(local_rank), len(peers), cpus, sysfs)
    placement["shared_with"] = peers
    try:
        os.sched_setaffinity(0, new)
        placement.update(cpus=sorted(new), source=source, bdf=actual, rebound_after_gpu_init=True)
    except OSError as exc:
        placement["error"] = "%s: %s" % (type(exc).__name__, exc)
    return placement


def summary(placement):
    """The placement as the short record a bench line / log carries."""
    if placement is None:
        return None
    out = {"bound": bool(placement.get("bound")), "source": placement.get("source"), "cpus": format_cpulist(placement.get("cpus") or []),
           "n_cpus": len(placement.get("cpus") or []), "bdf": placement.get("bdf")}
    for key in ("bdf_from_runtime", "bdf_agrees", "rebound_after_gpu_init", "error", "shared_with"):
        if key in placement:
            out[key] = placement[key]
    return out
