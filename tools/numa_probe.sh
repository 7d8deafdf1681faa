#!/bin/bash
# Does the step time of a host-bound scene (5k Gaussians / 256^2) follow the host core the process runs on? Topology, then the same
# bench line under different CPU sets (taskset). Runs on the GPU box.
echo "== topology"
nproc; lscpu 2>/dev/null | grep -i "numa\|socket\|model name\|thread" | head -12
for d in /sys/class/drm/card*/device; do [ -e $d/numa_node ] && echo "$d numa_node=$(cat $d/numa_node) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done
echo "affinity of this shell: $(taskset -pc $$ 2>/dev/null | cut -d: -f2)"
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -1
line() { python bench.py --workload 5k-256-sh0 --cpu-budget 0 --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
nodes=$(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)
echo "== free (no pinning)"; for i in 1 2 3 4; do line; done
for n in $(seq 0 $((nodes-1))); do
  cl=$(cat /sys/devices/system/node/node$n/cpulist)
  echo "== taskset -c $cl (node $n)"; for i in 1 2 3; do taskset -c $cl bash -c "$(declare -f line); line"; done
done
first=$(cat /sys/devices/system/node/node0/cpulist | cut -d, -f1 | cut -d- -f1)
echo "== taskset -c $first (ONE core)"; for i in 1 2 3; do taskset -c $first bash -c "$(declare -f line); line"; done
