"""Rank placement (catfish_amd/placement.py) against a fake sysfs tree: which CPUs a rank of the sharded per-file loop
(catfish/catfish:50-82, one process per GPU) binds itself to.  No GPU, no real sysfs: every source (PCI local_cpulist, NUMA
node, even split) and the slicing between ranks that share a locality set is driven from files under tmp_path."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from catfish_amd import placement  # noqa: E402


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(text)


def _bdf(i):
    return "0000:%02x:00.0" % (0x05 + 0x10 * i)


def fake_node(root, n_gpus=8, local_cpulist=True, numa=True, n_cpus=64, smt=True):
    """A two-socket node: CPUs 0..n/2-1 are cores, c + n/2 their SMT siblings; socket 0 = cores [0, n/4), socket 1 = [n/4, n/2);
    KFD nodes 0-1 are the CPUs, 2.. the GPUs (first half on socket 0)."""
    root = str(root)
    half, quarter = n_cpus // 2, n_cpus // 4
    sockets = [list(range(0, quarter)) + list(range(half, half + quarter)),
               list(range(quarter, half)) + list(range(half + quarter, n_cpus))]
    for s, cpus in enumerate(sockets):
        _write(os.path.join(root, "devices/system/node/node%d/cpulist" % s), placement.format_cpulist(cpus) + "\n")
        _write(os.path.join(root, "class/kfd/kfd/topology/nodes/%d/properties" % s), "cpu_cores_count %d\nsimd_count 0\n" % len(cpus))
    for c in range(n_cpus):
        sib = sorted({c, (c + half) % n_cpus}) if smt else [c]
        _write(os.path.join(root, "devices/system/cpu/cpu%d/topology/thread_siblings_list" % c), placement.format_cpulist(sib) + "\n")
    for g in range(n_gpus):
        bus = 0x05 + 0x10 * g
        _write(os.path.join(root, "class/kfd/kfd/topology/nodes/%d/properties" % (2 + g)),
               "cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id %d\nunique_id 12345\n" % (bus << 8))
        dev = os.path.join(root, "bus/pci/devices", _bdf(g))
        socket = 0 if g < (n_gpus + 1) // 2 else 1
        _write(os.path.join(dev, "local_cpulist"), (placement.format_cpulist(sockets[socket]) if local_cpulist else "") + "\n")
        _write(os.path.join(dev, "numa_node"), "%d\n" % (socket if numa else -1))
    return root


def test_cpulist_round_trip():
    assert placement.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert placement.parse_cpulist("") == [] and placement.parse_cpulist(None) == [] and placement.parse_cpulist("x,3") == [3]
    assert placement.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert placement.normalize_bdf("0000:C5:00.0") == "0000:c5:00.0" == placement.normalize_bdf("c5:00.0")
    assert placement.normalize_bdf("nonsense") is None and placement.normalize_bdf("") is None


def test_kfd_topology_gives_the_pci_addresses_without_a_hip_call(tmp_path, monkeypatch):
    root = fake_node(tmp_path)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert placement.kfd_gpu_bdfs(root) == [_bdf(g) for g in range(8)]
    assert placement.device_bdf(3, root) == _bdf(3) and placement.device_bdf(8, root) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6,2")                 # HIP device 0 is then the node's GPU 6
    assert placement.device_bdf(0, root) == _bdf(6) and placement.device_bdf(1, root) == _bdf(2) and placement.device_bdf(2, root) is None
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,5,6,7,0,1,2,3")    # ROCr filters first, HIP indexes into what is left
    assert placement.device_bdf(0, root) == _bdf(2)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")          # a UUID form: no filter is applied (verify() settles it later)
    assert placement.device_bdf(0, root) == _bdf(4)
    assert placement.kfd_gpu_bdfs(str(tmp_path / "nowhere")) == []


def test_source_1_pci_local_cpulist_eight_ranks_get_disjoint_whole_cores(tmp_path):
    root = fake_node(tmp_path)
    plans = [placement.plan(r, 8, allowed=range(64), sysfs=root) for r in range(8)]
    assert [p["source"] for p in plans] == ["pci"] * 8
    assert [p["bdf"] for p in plans] == [_bdf(g) for g in range(8)]
    sets = [set(p["cpus"]) for p in plans]
    assert all(len(s) == 8 for s in sets) and len(set().union(*sets)) == 64            # disjoint, the whole node used
    assert set().union(*sets[:4]) == set(range(0, 16)) | set(range(32, 48))             # GPUs 0-3: socket 0 only
    assert sets[0] == {0, 1, 2, 3, 32, 33, 34, 35}                                      # whole cores: a core and its SMT sibling together
    assert plans[5]["shared_with"] == [4, 5, 6, 7]


def test_source_2_numa_node_when_local_cpulist_is_empty(tmp_path):
    root = fake_node(tmp_path, local_cpulist=False)
    p = placement.plan(6, 8, allowed=range(64), sysfs=root)
    assert p["source"] == "numa" and set(p["cpus"]) <= set(range(16, 32)) | set(range(48, 64)) and len(p["cpus"]) == 8
    all_sets = [set(placement.plan(r, 8, allowed=range(64), sysfs=root)["cpus"]) for r in range(8)]
    assert sum(len(s) for s in all_sets) == len(set().union(*all_sets)) == 64


def test_source_3_even_split_of_the_mask(tmp_path):
    # (a) no KFD topology at all (this container); (b) sysfs knows the device but says nothing about its neighbourhood;
    # (c) the neighbourhood lies outside the CPUs this process may use (a cgroup cpuset)
    empty = str(tmp_path / "empty")
    os.makedirs(empty)
    allowed = list(range(100, 116))
    for root, mask in ((empty, allowed), (fake_node(tmp_path / "b", local_cpulist=False, numa=False), allowed),
                       (fake_node(tmp_path / "c"), allowed)):
        plans = [placement.plan(r, 4, allowed=mask, sysfs=root) for r in range(4)]
        assert [p["source"] for p in plans] == ["split"] * 4
        assert [p["cpus"] for p in plans] == [[100, 101, 102, 103], [104, 105, 106, 107], [108, 109, 110, 111], [112, 113, 114, 115]]
    # a mask too small to cut (fewer than MIN_CPUS_PER_RANK each): every rank keeps all of it
    assert placement.plan(2, 8, allowed=range(8), sysfs=empty)["cpus"] == list(range(8))


def test_ranks_rehearsing_on_one_card_cut_its_neighbourhood_into_disjoint_slices(tmp_path):
    root = fake_node(tmp_path)
    plans = [placement.plan(r, 4, device_of_rank=lambda r: 0, allowed=range(64), sysfs=root) for r in range(4)]
    sets = [set(p["cpus"]) for p in plans]
    assert all(p["source"] == "pci" and p["bdf"] == _bdf(0) and p["shared_with"] == [0, 1, 2, 3] for p in plans)
    assert all(len(s) == 8 for s in sets) and set().union(*sets) == set(range(0, 16)) | set(range(32, 48))
    # a cgroup that leaves 6 of the card's CPUs: three cores, sliced 2 + 2 + 2 would split... whole set of 6 over 4 ranks is < 2 each -> shared
    few = [0, 1, 2, 32, 33, 34]
    assert [placement.plan(r, 4, device_of_rank=lambda r: 0, allowed=few, sysfs=root)["cpus"] for r in range(4)] == [few] * 4
    # ... and over 2 ranks: 3 each, core 0 whole on rank 0
    two = [placement.plan(r, 2, device_of_rank=lambda r: 0, allowed=few, sysfs=root)["cpus"] for r in range(2)]
    assert two == [[0, 1, 32], [2, 33, 34]]


def test_mixed_node_some_gpus_known_some_not(tmp_path):
    root = fake_node(tmp_path, n_gpus=1)                      # sysfs knows GPU 0 only (socket 0); ranks 1, 2 drive "devices" 1, 2
    plans = [placement.plan(r, 3, allowed=range(64), sysfs=root) for r in range(3)]
    assert [p["source"] for p in plans] == ["pci", "split", "split"]
    assert set(plans[0]["cpus"]) == set(range(0, 16)) | set(range(32, 48))
    rest = set(range(16, 32)) | set(range(48, 64))            # the ranks without locality share what the others left, whole cores each
    assert set(plans[1]["cpus"]) | set(plans[2]["cpus"]) == rest and not set(plans[1]["cpus"]) & set(plans[2]["cpus"])
    assert plans[1]["cpus"] == list(range(16, 24)) + list(range(48, 56))


_CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
from catfish_amd import placement
before = sorted(os.sched_getaffinity(0))
first = placement.bind(local_rank=int(sys.argv[1]), local_world=int(sys.argv[2]), sysfs=sys.argv[3])
again = placement.bind(local_rank=0, local_world=1, sysfs=sys.argv[3])      # once per process: the first record comes back
import threading
seen = []
t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))
t.start(); t.join()
print(json.dumps({"before": before, "after": sorted(os.sched_getaffinity(0)), "thread": seen[0], "same": again is first,
                  "summary": placement.summary(first), "current": placement.current() is first}))
"""


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no sched_setaffinity on this platform")
def test_bind_really_binds_and_threads_inherit_it(tmp_path):
    import json
    mask = sorted(os.sched_getaffinity(0))
    if len(mask) < 4:
        pytest.skip("needs 4 CPUs")
    empty = str(tmp_path / "empty")
    os.makedirs(empty)
    env = dict(os.environ)
    env.pop("CATFISH_BIND", None)
    outs = []
    for r in range(2):
        res = subprocess.run([sys.executable, "-c", _CHILD % ROOT, str(r), "2", empty], stdout=subprocess.PIPE, env=env, check=True,
                             universal_newlines=True)
        outs.append(json.loads(res.stdout))
    assert outs[0]["before"] == mask
    assert outs[0]["after"] == mask[:len(mask) // 2] and outs[1]["after"] == mask[len(mask) // 2:]
    assert all(o["thread"] == o["after"] and o["same"] and o["current"] for o in outs)
    assert outs[0]["summary"]["bound"] is True and outs[0]["summary"]["source"] == "split"
    assert outs[0]["summary"]["cpus"] == placement.format_cpulist(mask[:len(mask) // 2])
    env["CATFISH_BIND"] = "0"
    res = subprocess.run([sys.executable, "-c", _CHILD % ROOT, "0", "2", empty], stdout=subprocess.PIPE, env=env, check=True,
                         universal_newlines=True)
    off = json.loads(res.stdout)
    assert off["after"] == mask and off["summary"]["bound"] is False


def test_verify_keeps_a_right_guess_and_rebinds_a_wrong_one(tmp_path, monkeypatch):
    root = fake_node(tmp_path)
    calls = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(sorted(cpus)))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    monkeypatch.setattr(placement, "_APPLIED", None)
    p = placement.bind(local_rank=1, local_world=8, sysfs=root)
    assert p["bound"] and p["bdf"] == _bdf(1) and calls == [p["cpus"]]
    placement.verify(p, "0000:15:00.0".upper(), local_rank=1, local_world=8, sysfs=root)        # the runtime agrees (upper-case hex)
    assert p["bdf_agrees"] is True and len(calls) == 1 and "rebound_after_gpu_init" not in p
    placement.verify(p, _bdf(6), local_rank=1, local_world=8, sysfs=root)                       # it is the card on the other socket
    assert p["bdf_agrees"] is False and p["rebound_after_gpu_init"] is True and p["bdf"] == _bdf(6)
    assert set(calls[-1]) <= set(range(16, 32)) | set(range(48, 64)) and calls[-1] == p["cpus"]
    # the real card's locality set is cut among the ranks that SHARE it -- this rank plus the four planned on that socket --,
    # not among all eight ranks of the node: a fifth of the socket (whole cores), not an eighth
    assert p["shared_with"] == [1, 4, 5, 6, 7]
    assert p["cpus"] == placement.slice_for(0, 5, sorted(set(range(16, 32)) | set(range(48, 64))), root)
    assert len(p["cpus"]) > len(placement.slice_for(1, 8, sorted(set(range(16, 32)) | set(range(48, 64))), root)) or len(p["cpus"]) == 32
    s = placement.summary(p)
    assert s["bdf_from_runtime"] == _bdf(6) and s["rebound_after_gpu_init"] is True and s["n_cpus"] == len(p["cpus"])
    monkeypatch.setattr(placement, "_APPLIED", None)
