"""Pins two restatements the bit-exactness of the device path rests on:
 * glibc sinf/cosf (steering of the BRIEF pattern) against the live libm of this image,
 * libstdc++ std::sort (tie order of quad-tree nodes) against the real std::sort."""
import ctypes as C

import numpy as np


def test_sincos_restatement_matches_libm(emu_lib):
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    emu_lib.rgbl_test_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(0)
    # the reference feeds angle*pi/180 with angle in [0, 360): sample that range densely plus edge values
    deg = np.concatenate([rng.uniform(0, 360, 60000), np.arange(0, 361, 0.25), [1e-6, 1e-3, 44.999, 45.0, 45.001, 359.9999]])
    xs = (deg.astype(np.float32) * np.float32(np.pi / 180)).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    for x in xs:
        emu_lib.rgbl_test_sincosf(float(x), C.byref(s), C.byref(c))
        assert np.float32(s.value).view(np.uint32) == np.float32(libm.sinf(float(x))).view(np.uint32), x
        assert np.float32(c.value).view(np.uint32) == np.float32(libm.cosf(float(x))).view(np.uint32), x


def _sort_both(emu_lib, oracle, key, val):
    k1, v1 = key.copy(), val.copy()
    k2, v2 = key.copy(), val.copy()
    emu_lib.rgbl_test_std_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    oracle.lib().orc_std_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    emu_lib.rgbl_test_std_sort(k1.ctypes.data, v1.ctypes.data, len(k1))
    if len(key) <= 2048:  # the workgroup-parallel form used by the quad-tree kernel must agree as well
        k3, v3 = key.copy(), val.copy()
        emu_lib.rgbl_test_block_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        emu_lib.rgbl_test_block_sort(k3.ctypes.data, v3.ctypes.data, len(k3))
        assert np.array_equal(k3, k1) and np.array_equal(v3, v1), "block sort differs from the serial restatement"
    oracle.lib().orc_std_sort_pairs(k2.ctypes.data, v2.ctypes.data, len(k2))
    assert np.array_equal(k1, k2)
    assert np.array_equal(v1, v2), "tie order differs from std::sort"


def test_introsort_restatement_reproduces_std_sort_tie_order(emu_lib, oracle):
    rng = np.random.default_rng(1)
    for n in list(range(0, 40)) + [63, 64, 65, 100, 257, 1000, 1737, 5000]:
        for distinct in (2, 5, 50, 10**9):
            key = rng.integers(0, distinct, n).astype(np.uint64)
            val = np.arange(n, dtype=np.uint32)
            _sort_both(emu_lib, oracle, key, val)
    # (size, UL.x)-like keys with heavy ties, as the quad-tree produces them
    for n in (120, 434, 1737):
        key = (rng.integers(2, 6, n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 40, n).astype(np.uint64) * np.uint64(31)
        _sort_both(emu_lib, oracle, key, np.arange(n, dtype=np.uint32))


def test_introsort_heapsort_fallback_path(emu_lib, oracle):
    # median-of-three killer sequence (Musser): drives introsort into its depth limit -> heap sort branch
    # (40 .. 200: small sizes at which the depth limit 2 floor(log2 n) is exhausted inside short ranges, so that the heap-sort
    # branch starts from ranges of a few dozen elements as well as from long ones)
    for n in (40, 64, 66, 70, 72, 80, 90, 100, 128, 200, 512, 2048):
        k = n // 2
        a = np.zeros(n, np.uint64)
        for i in range(1, k + 1):
            if i % 2 == 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        _sort_both(emu_lib, oracle, a, np.arange(n, dtype=np.uint32))
    # sorted / reversed / organ-pipe inputs
    for n in (100, 3000):
        base = np.arange(n, dtype=np.uint64) // 3
        for key in (base, base[::-1].copy(), np.concatenate([base[: n // 2], base[: n - n // 2][::-1]])):
            _sort_both(emu_lib, oracle, key.copy(), np.arange(n, dtype=np.uint32))
