"""How many CPUs this process may actually keep busy: the smaller of the affinity mask and the cgroup CPU quota.
A container can see 256 logical CPUs and still be throttled to a fraction of them (cpu.max); running more busy
threads than that stalls every thread of the process for the rest of the scheduler period."""
import math
import os


def effective_cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(math.floor(int(quota) / int(period)))))
        except (OSError, ValueError):
            pass
    try:   # cgroup v1
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    if os.environ.get('SD_CPUS'):   # explicit cap (e.g. to rehearse a rank's share of a multi-GPU node on one GPU)
        n = min(n, max(1, int(os.environ['SD_CPUS'])))
    return max(1, n)


def configure_openmp(threads=None):
    """call before the native libraries are loaded: bounded team size, idle team threads sleep instead of spinning
    (spinning threads burn the CPU quota the working threads need)"""
    n = threads or effective_cpus()
    os.environ.setdefault('OMP_NUM_THREADS', str(n))
    os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
    os.environ.setdefault('GOMP_SPINCOUNT', '0')
    return int(os.environ['OMP_NUM_THREADS'])
