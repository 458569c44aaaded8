"""The library's own device-wide primitives (csrc/hip/sd_scan_sort.h: reduce-then-scan prefix sums / running maxima and the stable LSD
radix sort of (key, value) pairs that replaced hipCUB on the hot path) through their self-test entry points of the C ABI, against numpy:
sizes around the tile (4 096) and workgroup boundaries, bit ranges that are not multiples of the digit, equal keys (stability)."""
import ctypes as C

import numpy as np
import pytest

from spacedust_amd import _lib
from spacedust_amd._lib import ptr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 2, 63, 64, 255, 4095, 4096, 4097, 65536, 1000003, 5 * 4096 * 1024 + 17])
def test_radix_sort_pairs_is_stable_and_sorted(gpu, n):
    L = _lib.load()
    L.sd_selftest_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(n)
    for begin, end, spread in ((0, 32, 1 << 32), (0, 16, 1 << 16), (0, 5, 40), (7, 19, 1 << 22), (3, 32, 1 << 32), (0, 32, 3)):
        keys = rng.integers(0, spread, size=n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        ok, ov = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        assert L.sd_selftest_sort_pairs(gpu.h, ptr(keys), ptr(vals), n, begin, end, ptr(ok), ptr(ov)) == 0
        digit = (keys >> np.uint32(begin)) & np.uint32((1 << (end - begin)) - 1 if end - begin < 32 else 0xFFFFFFFF)
        order = np.argsort(digit, kind='stable')
        assert np.array_equal(ov, vals[order]), (n, begin, end)
        assert np.array_equal(ok, keys[order]), (n, begin, end)
        if n > 2000000:
            break   # (one bit range at the largest size)


@pytest.mark.parametrize('n', [1, 15, 16, 4095, 4096, 4097, 8 * 4096 * 1024 + 3])
def test_scans_equal_numpy(gpu, n):
    L = _lib.load()
    L.sd_selftest_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(n + 1)
    x = rng.integers(0, 1 << 31, size=n, dtype=np.uint64).astype(np.uint32)
    ex = np.zeros(n + 1, np.uint64)
    mx = np.zeros(n, np.uint32)
    assert L.sd_selftest_scan(gpu.h, ptr(x), n, ptr(ex), ptr(mx)) == 0
    want = np.zeros(n + 1, np.uint64)
    np.cumsum(x.astype(np.uint64), out=want[1:])
    assert np.array_equal(ex, want)
    assert np.array_equal(mx, np.maximum.accumulate(x))
