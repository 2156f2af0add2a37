#!/bin/bash
# GPU box: what sysfs says about the card(s) and the CPUs around them -- the inputs of catfish_amd/placement.py.
echo "== affinity"; python3 -c "import os; print(sorted(os.sched_getaffinity(0)))"; nproc
echo "== kfd topology"
for n in /sys/class/kfd/kfd/topology/nodes/*; do
  echo "-- $n"; grep -E "^(cpu_cores_count|simd_count|domain|location_id|unique_id|gfx_target_version) " $n/properties 2>&1
done
echo "== pci devices of class display/processing accelerator with local_cpulist"
for d in /sys/bus/pci/devices/*; do
  c=$(cat $d/class 2>/dev/null)
  case "$c" in 0x03*|0x12*) echo "$d class $c vendor $(cat $d/vendor 2>/dev/null) numa_node $(cat $d/numa_node 2>&1) local_cpulist $(cat $d/local_cpulist 2>&1)";; esac
done
echo "== numa nodes"; for n in /sys/devices/system/node/node*; do echo "$n $(cat $n/cpulist 2>&1)"; done
echo "== cpu0 siblings"; cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list 2>&1
echo "== visible"; echo "HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES"
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from catfish_amd import placement
print("kfd bdfs:", placement.kfd_gpu_bdfs())
print("plan 1 rank:", placement.summary(placement.plan(0, 1)))
print("plan rank 2 of 4 on device 0:", placement.summary(placement.plan(2, 4, device_of_rank=lambda r: 0)))
from catfish_amd.engine import device_identity
print("runtime identity of device 0:", device_identity(0))
PY
