"""SURVEY.md 8(b) "Threading": two extractor objects called from two std::threads exactly as Frame.cc:122-125 does, and the
matcher entry points used concurrently from three threads (Tracking / LocalMapping / LoopClosing) on pooled handles while an
extractor runs - tests/shim_threads_test.cpp.  The C++ program holds every concurrent result to the sequential run of the same
call (and every thread's rgbl_last_error() to its own error); this side holds the sequential results to the oracle.
Three builds: the SIMT emulation, the emulation under ThreadSanitizer (host side of the library: thread-local error state,
the matcher pool, handle state; the emulator's fibers are announced to TSan), and `-m gpu` the product on the MI355X."""
import fcntl
import glob
import os
import struct
import subprocess

import numpy as np
import pytest

import parity_checks as pc
import shim_driver
from oracle import oracle_py as O
from orb_slam3_rgbl_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "orb_slam3_rgbl_amd", "shim")
BUILD = os.path.join(ROOT, "tests", "_build")


def build_exe(libdir, libname, exe, extra=()):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "shim_threads_test.cpp"), os.path.join(SHIM, "ORBextractor.cc")]
    deps = srcs + glob.glob(os.path.join(SHIM, "*.h")) + [os.path.join(ROOT, "include", "rgbl_frontend.h"),
                                                           os.path.join(ROOT, "tests", "shim_standins.h"), os.path.join(libdir, "lib%s.so" % libname)]
    with open(exe + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
            return
        tmp = "%s.tmp.%d" % (exe, os.getpid())
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-DRGBL_FORCE_CV_COMPAT", "-I" + SHIM, "-I" + os.path.join(ROOT, "tests")] + list(extra) + srcs +
                              ["-o", tmp, "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-pthread"])
        os.replace(tmp, exe)


def build_tsan_emulator():
    """The kernel sources + C ABI for the SIMT emulator with -fsanitize=thread (tests/emu/hip_emu.h announces its fibers)."""
    lib = os.path.join(BUILD, "librgbl_frontend_emu_thread.so")
    csrc = os.path.join(ROOT, "orb_slam3_rgbl_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("extractor.hip", "depth.hip", "matcher.hip", "records.hip", "gather.hip")] + \
           [os.path.join(ROOT, "tests", "emu", f) for f in ("hip_emu.cpp", "nccl_emu.cpp")]
    deps = srcs + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(ROOT, "tests", "emu", "*.h")) + [os.path.join(ROOT, "include", "rgbl_frontend.h")]
    os.makedirs(BUILD, exist_ok=True)
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in deps):
            return lib
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-DRGBL_EMU", "-fsanitize=thread", "-fno-omit-frame-pointer",
               "-I" + os.path.join(ROOT, "tests", "emu"), "-Wno-unknown-pragmas", "-shared", "-o", lib + ".tmp"]
        for s in srcs:
            cmd += ["-x", "c++", s]
        subprocess.check_call(cmd, cwd=csrc)
        os.replace(lib + ".tmp", lib)
    return lib


def write_inputs(tmp, w, h, n_tri, n1, n2):
    left, right = pc.stereo_pair(55, w, h)
    left.tofile(os.path.join(tmp, "left.raw"))
    right.tofile(os.path.join(tmp, "right.raw"))
    kf1, kf2, K, R, t, _, sf, s2 = pc.make_triangulation_case(n_tri, seed=35)
    T1w = np.eye(4, dtype=np.float32)
    Tw2 = np.eye(4, dtype=np.float32)
    Tw2[:3, 3] = [0.54, -0.01, 0.9]
    T2w = np.linalg.inv(Tw2).astype(np.float32)
    with open(os.path.join(tmp, "tri.bin"), "wb") as f:
        f.write(K.astype(np.float32).tobytes())
        shim_driver.write_kf(f, kf1, sf, s2, T1w, np.linalg.inv(T1w).astype(np.float32))
        shim_driver.write_kf(f, kf2, sf, s2, T2w, Tw2)
    case = pc.make_projection_case(n1, n2, seed=37, motion="forward")
    with open(os.path.join(tmp, "proj.bin"), "wb") as f:
        f.write(struct.pack("<iifii", len(case["valid1"]), len(case["kp2_xy"]), 15.0, 0, 1))
        hdr = np.concatenate([case["grid"], case["Tcw_q"], case["Tcw_t"], case["Tlw_q"], case["Tlw_t"], case["K"],
                              [case["mb"], case["mbf"]], case["scale_factors"]]).astype(np.float32)
        f.write(hdr.tobytes())
        for key, dt in (("valid1", np.uint8), ("world_pos1", np.float32), ("mp_desc1", np.uint8), ("mp_observed1", np.uint8),
                        ("octave1", np.int32), ("angle1", np.float32), ("kp2_xy", np.float32), ("kp2_octave", np.int32),
                        ("kp2_angle", np.float32), ("uright2", np.float32), ("desc2", np.uint8)):
            f.write(np.ascontiguousarray(case[key], dt).tobytes())
    T12 = T1w @ Tw2
    Fm = O.fundamental(K, K, T12[:3, :3].reshape(9), T12[:3, 3])
    C2 = T2w[:3, 3]
    ep = np.array([K[0] * C2[0] / C2[2] + K[2], K[1] * C2[1] / C2[2] + K[3]], np.float32)
    return left, right, (kf1, kf2, Fm, ep, sf, s2), case


def run_and_check(exe, tmp, w, h, iters, n_tri=1200, n1=1500, n2=1700, env=None, nfeat=None, nlevels=8, small=False):
    left, right, tri, case = write_inputs(tmp, w, h, n_tri, n1, n2)
    out = os.path.join(tmp, "threads_out.bin")
    res = subprocess.run([exe, str(w), str(h), os.path.join(tmp, "left.raw"), os.path.join(tmp, "right.raw"), os.path.join(tmp, "tri.bin"),
                          os.path.join(tmp, "proj.bin"), str(iters), out, str(nfeat or (2000 if w >= 1000 else 500)), str(nlevels)],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-6000:]
    assert "threads ok" in res.stdout and " 0 mismatches" in res.stdout
    # the sequential results the threads were held to, against the oracle
    buf = open(out, "rb").read()
    pos = 0

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(buf, dtype, count, pos)
        pos += a.nbytes
        return a
    nfeat = nfeat or (2000 if w >= 1000 else 500)
    descs = []
    for img in (left, right):
        mono, nk = take(np.int32, 2)
        kps, desc = take(O.KP_DTYPE, nk), take(np.uint8, nk * 32).reshape(nk, 32)
        okps, odesc, omono = O.Extractor(nfeat, 1.2, nlevels, 20, 7)(img)
        pc.assert_keypoints_equal(kps, okps, "threaded extractor")
        assert np.array_equal(desc, odesc) and mono == omono
        descs.append(desc)
    nm, npairs = take(np.int32, 2)
    pairs = take(np.int32, 2 * npairs).reshape(npairs, 2)
    kf1, kf2, Fm, ep, sf, s2 = tri
    om12, onm = O.search_triangulation(kf1, kf2, Fm, ep, sf, s2, False, False, False)
    idx1 = np.nonzero(om12 >= 0)[0]
    assert nm == onm == npairs and np.array_equal(pairs[:, 0], idx1) and np.array_equal(pairs[:, 1], om12[idx1]) and nm > (2 if small else 20)
    pn, n2_ = take(np.int32, 2)
    m2 = take(np.int32, n2_)
    om, on = O.search_by_projection(case, 15.0, False, True)
    assert pn == on and np.array_equal(m2, om) and on > (5 if small else 100)
    nbf = take(np.int32, 1)[0]
    bi, bd, sd = take(np.int32, nbf), take(np.int32, nbf), take(np.int32, nbf)
    obi, obd, osd = O.hamming_bf(descs[0], descs[1])
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    return res


def test_threads_under_emulation(emu_lib, tmp_path):
    exe = os.path.join(BUILD, "shim_threads_test_emu")
    build_exe(BUILD, "rgbl_frontend_emu", exe)
    run_and_check(exe, str(tmp_path), 480, 270, 3, n_tri=500, n1=500, n2=600)


@pytest.mark.skipif(not os.environ.get("RGBL_TSAN"), reason="three minutes under ThreadSanitizer: RGBL_TSAN=1 runs it (tools/sanitize_emu.sh does; "
                                                                "the log of the round's run: profiles/r06_tsan_threads.txt)")
def test_threads_under_thread_sanitizer(oracle, tmp_path):
    """TSan clean: no data race in the library's host code between the reference's concurrent callers."""
    lib = build_tsan_emulator()
    exe = os.path.join(BUILD, "shim_threads_test_tsan")
    build_exe(BUILD, "rgbl_frontend_emu_thread", exe, extra=("-fsanitize=thread",))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 second_deadlock_stack=1")
    # the emulator's fibers are slow under TSan (every work-item switch is announced): a 96 x 80 image, one pyramid level, one
    # round - five extractions and two calls of every matcher, ~2 minutes
    res = run_and_check(exe, str(tmp_path), 96, 80, 1, n_tri=60, n1=60, n2=80, env=env, nfeat=60, nlevels=1, small=True)
    assert "ThreadSanitizer" not in res.stderr, res.stderr[-6000:]
    assert os.path.exists(lib)


@pytest.mark.gpu
def test_threads_on_mi355x(gpu_lib, tmp_path):
    exe = os.path.join(BUILD, "shim_threads_test_gpu")
    build_exe(os.path.join(ROOT, "orb_slam3_rgbl_amd"), "rgbl_frontend", exe)
    run_and_check(exe, str(tmp_path), synth.KITTI_W, synth.KITTI_H, 50)
