cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc; python3 -c "import os;print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; lscpu | grep -E "NUMA|Socket|Thread|Model name" ; cat /proc/meminfo | grep -i huge | head -3; cat /sys/kernel/mm/transparent_hugepage/enabled
python3 - <<'PY'
import time, threading, numpy as np
def spin(n):
    x=0
    for i in range(n): x+=i
# measure parallel speedup of a pure-CPU native loop via numpy (releases GIL): matrix dot small
a=np.random.rand(256,256).astype(np.float32)
def work():
    for _ in range(400): a@a
for nt in (1,4,16,64,128,256):
    th=[threading.Thread(target=work) for _ in range(nt)]
    t=time.time(); [x.start() for x in th]; [x.join() for x in th]; dt=time.time()-t
    print(nt, 'threads', round(dt,3),'s  speedup', round(nt*0.0+ (nt/dt),1))
PY
