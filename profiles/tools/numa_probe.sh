lscpu | grep -i "numa\|socket\|model name" | head -12
for d in /sys/class/drm/card*/device; do echo $d $(cat $d/local_cpulist 2>/dev/null) numa=$(cat $d/numa_node 2>/dev/null); done | head -12
python - <<'PY'
import os, torch
print("allowed cpus:", sorted(os.sched_getaffinity(0))[:8], "...", len(os.sched_getaffinity(0)))
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if "pci" in k})
PY
for i in 1 2 3 4 5 6 7 8; do
  LORADS_FORCE_DIST=1 LORADS_PRINT_CPU=1 timeout -k 10 300 python bench.py --no-cpu --no-extra 2>/tmp/e.txt | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', round(d['value'],1), round(d['ms_per_step'],4))"; grep "cpu at end" /tmp/e.txt
done
