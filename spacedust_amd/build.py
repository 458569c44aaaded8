"""Build libsdgpu.so in-tree: host stages with g++ (same FP flags as the reference's AVX2 build so the
double-precision stages contract identically), HIP kernels with hipcc for gfx950, linked by hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libsdgpu.so')
CLI = os.path.join(HERE, 'sdgpu')   # the multi-call host driver (csrc/cli), linked against libsdgpu.so
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
HOST_FLAGS = ['-std=c++17', '-O3', '-mavx2', '-mfma', '-ffp-contract=fast', '-fopenmp', '-fPIC']
HIP_FLAGS = ['--offload-arch=gfx950', '-std=c++17', '-O3', '-fPIC', '-fopenmp', '-Wno-unused-result'] + os.environ.get('SD_HIP_EXTRA', '').split()   # (SD_HIP_EXTRA: tuning builds, e.g. -DSD_KP_LOW=17)


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False):
    inc = ['-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(CSRC, 'host'), '-I' + os.path.join(CSRC, 'hip')]
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(ROOT, 'include', 'spacedust_gpu.h')]
    for d in ('host', 'hip'):
        headers += [os.path.join(CSRC, d, f) for f in os.listdir(os.path.join(CSRC, d)) if f.endswith(('.h', '.inc'))]
    newest_hdr = max(os.path.getmtime(h) for h in headers)
    objs = []
    jobs = []
    for f in sorted(os.listdir(os.path.join(CSRC, 'host'))):
        if f.endswith('.cpp'):
            src = os.path.join(CSRC, 'host', f)
            obj = os.path.join(objdir, f[:-4] + '.o')
            objs.append(obj)
            if force or _newer(src, obj) or newest_hdr > os.path.getmtime(obj):
                jobs.append(['g++'] + HOST_FLAGS + inc + ['-c', src, '-o', obj])
    for f in sorted(os.listdir(os.path.join(CSRC, 'hip'))):
        if f.endswith('.hip'):
            src = os.path.join(CSRC, 'hip', f)
            obj = os.path.join(objdir, f[:-4] + '.hip.o')
            objs.append(obj)
            if force or _newer(src, obj) or newest_hdr > os.path.getmtime(obj):
                jobs.append([HIPCC] + HIP_FLAGS + inc + ['-c', src, '-o', obj])
    procs = []
    for cmd in jobs:
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('compile failed: ' + ' '.join(cmd))
    if jobs or not os.path.exists(OUT):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-fopenmp'] + objs + ['-o', OUT]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    build_cli(force=force or bool(jobs), verbose=verbose)
    build_tools(force=force, verbose=verbose)
    return OUT


def build_tools(force=False, verbose=False):
    """tools/libvalupeak.so: the VALU issue micro-benchmark behind bench.py's sw_valu calibration (not part of the product)"""
    # tools/launch_latency: dependent launches beside long-lived workgroups; a measurement tool, built when its source is newer
    lsrc = os.path.join(ROOT, 'tools', 'csrc', 'launch_latency.hip')
    lout = os.path.join(ROOT, 'tools', 'launch_latency')
    if os.path.exists(lsrc) and (force or _newer(lsrc, lout)):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O2', '-w', '-o', lout, lsrc])
    # tools/fetch_calib: known byte counts per access pattern for the FETCH_SIZE / WRITE_SIZE calibration (tools/fetch_calib.sh)
    csrc_ = os.path.join(ROOT, 'tools', 'csrc', 'fetch_calib.hip')
    cout = os.path.join(ROOT, 'tools', 'fetch_calib')
    if os.path.exists(csrc_) and (force or _newer(csrc_, cout)):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O2', '-w', '-o', cout, csrc_])
    src = os.path.join(ROOT, 'tools', 'csrc', 'valu_peak.hip')
    out = os.path.join(ROOT, 'tools', 'libvalupeak.so')
    if not os.path.exists(src) or (not force and not _newer(src, out)):
        return out
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-w', '-shared', '-fPIC', '-o', out, src]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out


def build_cli(force=False, verbose=False):
    """spacedust_amd/sdgpu: the C++ host driver speaking the reference's module command lines and DB files; it reaches the
    kernels only through the C ABI of libsdgpu.so (found next to the binary via $ORIGIN)."""
    cli = os.path.join(CSRC, 'cli')
    srcs = sorted(os.path.join(cli, f) for f in os.listdir(cli) if f.endswith('.cpp'))
    deps = srcs + [os.path.join(cli, f) for f in os.listdir(cli) if f.endswith('.h')] + [os.path.join(ROOT, 'include', 'spacedust_gpu.h')]
    if not force and os.path.exists(CLI) and all(os.path.getmtime(d) <= os.path.getmtime(CLI) for d in deps):
        return CLI
    objdir = os.path.join(HERE, 'build')
    objs, procs = [], []
    for src in srcs:
        obj = os.path.join(objdir, 'cli_' + os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        cmd = ['g++', '-std=c++17', '-O2', '-fopenmp', '-I' + os.path.join(ROOT, 'include'), '-I' + cli, '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('compile failed: ' + ' '.join(cmd))
    cmd = ['g++', '-fopenmp'] + objs + ['-L' + HERE, '-lsdgpu', '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,/opt/rocm/lib', '-lpthread', '-o', CLI]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
