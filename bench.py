#!/usr/bin/env python3
"""bench.py — RGB-L front-end throughput (extract + depth + match) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic KITTI-resolution RGB-L input that is
already resident in HBM: ORB extraction (pyramid, FAST, quad-tree, orientation, blur, rBRIEF) of B frames,
LiDAR depth (projection, inverse dilation, keypoint gather) of the B scans, Hamming brute-force matching
of every frame against its successor.  Workload = BASELINE.json configs[1] (KITTI-00 RGB-L, nFeatures 2000,
1241x376, 8 levels, FAST 12/7, InverseDilation Diamond-5); `--workload 4k` selects configs[4].

Launch contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver starts it under
torch.distributed.run with one rank per GPU (backend nccl == RCCL) - and a bare `python bench.py --gpus N` (no WORLD_SIZE in
the environment) re-executes itself that way, or exits non-zero when the node has fewer than N GPUs: `n_gpus` of the JSON
line is always what ran (`launch_plan`).  Frames / sequences shard over the ranks with no data-path collective; the only
communication is the gather of the variable-length keypoint/descriptor/depth records to rank 0 (weak scaling), by default
through the library's own C-ABI entry points over RCCL (`--transport abi`: rgbl_gather_*, csrc/gather.hip), or through
torch.distributed (`--transport torch`).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# dmabuf IPC: on this driver RCCL between processes (and any sharing of device memory across processes) needs it, and it has to
# be in the environment BEFORE the HIP runtime is initialised - i.e. before `import torch` - on every path: the self-launch,
# ranks started by the driver's own torchrun, a single process.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, nfeatures, lidar azimuth steps, default batch)
    "kitti": (1241, 376, 2000, 1900, 1024),   # round 5: 1024 frames per step (+1.5 % over 512: longer launches, the same chain; 2048: +1.7 %)
    "4k": (3840, 2160, 8000, 4096, 64),
}
LEVELS, SCALE, INI_TH, MIN_TH = 8, 1.2, 12, 7
PROF_STEPS = 12    # serialised steps of the per-kernel timing leg (profiles/summarize_rocprof.py divides its counter totals by them)


def level_sizes(w, h):
    inv = [np.float32(1.0)]
    sc = np.float32(1.0)
    for _ in range(1, LEVELS):
        sc = np.float32(sc * np.float32(SCALE))
        inv.append(np.float32(1.0) / sc)
    return [(int(np.rint(np.float32(w) * s)), int(np.rint(np.float32(h) * s))) for s in inv]


def launch_plan(gpus, environ, device_count, argv, port=None):
    """What a `bench.py --gpus N` invocation has to do, as data (unit-tested in tests/test_bench_launch.py):
      ("run", world)        - go on in this process as one rank of `world`;
      ("exec", [argv ...])  - N > 1 asked for on a bare command line: re-execute under torch.distributed.run, one rank per GPU;
      ("error", message)    - the request cannot be honoured (fewer GPUs than ranks, or a launcher that started another
                              number of ranks than --gpus says): exit non-zero rather than print a line whose n_gpus is not N."""
    world_env = environ.get("WORLD_SIZE")
    if world_env is not None:
        world = int(world_env)
        if gpus != world:
            return ("error", "--gpus %d but the launcher started WORLD_SIZE=%d ranks: n_gpus must be what runs" % (gpus, world))
        if device_count is not None and int(environ.get("LOCAL_RANK", "0")) >= device_count:
            return ("error", "LOCAL_RANK %s has no GPU (%d visible)" % (environ.get("LOCAL_RANK"), device_count))
        return ("run", world)
    if gpus <= 1:
        return ("run", 1)
    if device_count is not None and device_count < gpus:
        return ("error", "--gpus %d asked for, %d HIP device(s) visible on this node: not launching (one rank per GPU, no oversubscription)" % (gpus, device_count))
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return ("exec", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
                     "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(argv))


def step_bytes_8d(w, h, n_points, k_per_frame):
    """ALGORITHMIC bytes of one frame through the whole step, B_ext + B_dep + B_m (zero fill and scatter of
    the depth maps included, as the contract counts them)."""
    px = [a * b for a, b in level_sizes(w, h)]
    sp = sum(px)
    b_ext = 5 * sp - px[0] - px[-1] + 60 * k_per_frame
    b_dep = 20 * n_points + 12 * w * h + 12 * k_per_frame
    b_m = 32 * 2 * k_per_frame + 8 * k_per_frame
    return b_ext + b_dep + b_m


def algorithmic_bytes(kernel, w, h, n_points, k_per_frame):
    """Compulsory bytes of ONE frame for one kernel."""
    px = [a * b for a, b in level_sizes(w, h)]
    sp = sum(px)
    table = {
        "k_resize_linear": (sp - px[-1]) + (sp - px[0]),          # read levels 0..L-2, write levels 1..L-1
        "k_fast_cells": sp,                                       # every pyramid pixel once
        "k_gauss7": 2 * sp,                                       # read + write
        "k_octree": 4 * 2 * k_per_frame * 5,                      # candidate keys in/out (small, latency bound)
        "k_orient_brief": 60 * k_per_frame,                       # 32 B descriptor + 28 B keypoint out
        "k_project_index": 16 * n_points,
        "k_project_write": 16 * n_points + 4 * n_points,
        "k_inverse_dilate": 8 * w * h,                            # read raw + write processed
        "k_gather_depth": 12 * k_per_frame,
        "k_hamming_bf": 32 * 2 * k_per_frame + 8 * k_per_frame,
        "k_hamming_mfma": 32 * 2 * k_per_frame + 8 * k_per_frame,
        "k_hamming_fp4": 32 * 2 * k_per_frame + 8 * k_per_frame,
    }
    return table.get(kernel)


TRAFFIC_FILES = {"kitti": ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"), "4k": ("r06_pmc_traffic_4k.json", "r05_pmc_traffic_4k.json", "r04_pmc_traffic_4k.json", "r03_pmc_traffic_4k.json")}
CLOCK_HZ, SIMDS = 2.4e9, 1024     # MI355X: 256 CUs x 4 SIMDs, one VALU instruction of a wave64 per 4 cycles and SIMD


def pmc_profile(workload):
    """The committed counter passes of this same command (`--serial`, so that dispatches per step are what this run launches):
    profiles/rNN_pmc_traffic*.json holds, per kernel, the FETCH_SIZE and WRITE_SIZE totals of one step in bytes (converted with
    the factors tools/micro/hbm_calib.hip measured on this GPU) and, from round 3 on, the SQ_INSTS_VALU total.  Returns
    (dict, file name) or (None, None)."""
    for name in TRAFFIC_FILES.get(workload, ()):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                return json.load(open(path)), "profiles/" + name
            except ValueError:
                pass
    return None, None


def pmc_traffic(prof, kernel, frames_per_launch, launches_per_step):
    """HBM-side bytes per LAUNCH of `kernel` from the committed passes; None when they do not know the kernel."""
    try:
        k = prof["kernels"][kernel]
        per_step = (k["fetch_bytes_per_step"] + k["write_bytes_per_step"]) * frames_per_launch / prof["frames_per_step"]
        return per_step / max(launches_per_step, 1)
    except (KeyError, TypeError, ValueError):
        return None


def measured_copy_ceiling(torch, dev, gib=1, reps=6):
    """SURVEY 8(d) "% of a measured device-copy ceiling": a device-to-device copy of `gib` GiB timed in THIS run (HIP events on
    the current stream), bytes read + bytes written per second.  /opt/skills/guides/MI355X_MICROARCH.md measured 6.29 TB/s."""
    n = gib << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.fill_(1)
    dst.copy_(src)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src)
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    del src, dst
    return 2.0 * n / (best * 1e-3) / 1e9


def roofline_of(kernels, prof_steps, workload, w, h, n_points, k_mean, B, step_s=None, copy_ceiling=None):
    """The roofline object of the dominant kernel (largest share of the serialised GPU time) from per-kernel HIP-event times
    {name: (ms summed over prof_steps steps, launches)}: algorithmic bytes over its own time against the 8 TB/s
    of HBM3E - for every kernel of the step as well (`per_kernel_hbm`) -, the HBM-side traffic of the committed counter passes,
    and, where those passes hold SQ_INSTS_VALU, the share of the VALU issue slots the kernel used: the pixel kernels of this
    path are bound by vector issue, not by bytes, and `bound` says which of the two is closer to its ceiling."""
    total_ms = sum(v[0] for v in kernels.values())
    dom = max(kernels, key=lambda k: kernels[k][0])
    ms_sum, launches = kernels[dom]
    per_step_ms = ms_sum / prof_steps                      # all launches of that kernel in one step
    per_launch_ms = ms_sum / max(launches, 1)
    ab = algorithmic_bytes(dom, w, h, n_points, k_mean)
    achieved = (ab * B) / (per_step_ms * 1e-3) / 1e9 if ab else None
    prof, source = pmc_profile(workload)
    traffic = pmc_traffic(prof, dom, B, launches / prof_steps)
    r = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
         "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic,
         "traffic_source": (source + " (counter passes of this command, replayed; not counted in this run)") if traffic else None,
         "traffic_over_algorithmic": (traffic / (ab * B * prof_steps / max(launches, 1))) if (traffic and ab) else None,
         "avg_launch_ms": per_launch_ms, "launches_per_step": launches / prof_steps,
         "algorithmic_bytes_per_frame": ab, "frames_per_launch": B,
         "kernel_share_of_gpu_time": ms_sum / total_ms if total_ms else None,
         "kernels_ms_per_step": {k: v[0] / prof_steps for k, v in sorted(kernels.items())}}
    if copy_ceiling:
        r["peak_measured"] = round(copy_ceiling, 1)
        r["peak_measured_what"] = "device-to-device copy of 1 GiB timed in this run (read + written bytes per second); the guide's figure is 6290 GB/s"
        r["frac_of_measured"] = (achieved / copy_ceiling) if achieved else None
    try:
        valu = prof["kernels"][dom]["valu_insts_per_step"] * B / prof["frames_per_step"]
        issue = valu * 4.0 / (per_step_ms * 1e-3 * CLOCK_HZ * SIMDS)
        r["valu_issue_frac"] = issue
        r["valu_issue_frac_what"] = ("SQ_INSTS_VALU x 4 cycles / (kernel time x 2.4 GHz x 1024 SIMDs): the 4 cycles are what tools/micro/valu_rates "
                                     "measures for the integer instructions these kernels are made of (0.20 - 0.25 wave-instructions per cycle and SIMD; "
                                     "v_add 0.34), not the guide's 2-cycle issue of a wave64 operation - an upper bound for a mixed stream")
        r["valu_insts_per_pixel"] = valu * 64.0 / (ab * B) if dom == "k_fast_cells" else None
        r["valu_source"] = source + " (SQ_INSTS_VALU of this command, replayed) over this run's kernel time"
        if r["frac"] is not None and issue > r["frac"]:
            r["bound"] = "valu"
    except (KeyError, TypeError):
        pass
    try:
        # the counter's own view: SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD issues vector work
        quad = prof["kernels"][dom]["valu_active_quadcycles_per_step"] * B / prof["frames_per_step"]
        r["valu_busy"] = quad * 4.0 / (per_step_ms * 1e-3 * CLOCK_HZ * SIMDS)
        r["valu_busy_source"] = source + " (SQ_ACTIVE_INST_VALU x 4 cycles of this command, replayed) / (this run's kernel time x 2.4 GHz x 1024 SIMDs)"
        if r["frac"] is not None and r["valu_busy"] > r["frac"]:
            r["bound"] = "valu"
    except (KeyError, TypeError):
        pass
    per_kernel = {}
    step_bytes = 0.0
    for k, v in sorted(kernels.items()):
        kb = algorithmic_bytes(k, w, h, n_points, k_mean)
        if kb and v[0] > 0:
            gbs = kb * B / (v[0] / prof_steps * 1e-3) / 1e9
            per_kernel[k] = {"GB/s": round(gbs, 1), "frac": round(gbs / 8000.0, 4)}
            if copy_ceiling:
                per_kernel[k]["frac_of_measured"] = round(gbs / copy_ceiling, 4)
            step_bytes += kb * B
    r["per_kernel_hbm"] = per_kernel
    if step_s:
        # the contract's figure: SURVEY 8(d) B_ext + B_dep + B_m per frame (the depth maps' zero fill and scatter included)
        r["step_algorithmic_GB/s"] = round(step_bytes_8d(w, h, n_points, k_mean) * B / step_s / 1e9, 1)
        r["step_algorithmic_bytes_per_frame"] = step_bytes_8d(w, h, n_points, k_mean)
        # the same over what the step's kernels actually have to move in this implementation (no zero fill: generation-tagged
        # index maps; no raw map): the sum of the per-kernel figures above
        r["step_kernel_sum_GB/s"] = round(step_bytes / step_s / 1e9, 1)
        r["step_algorithmic_frac"] = round(r["step_algorithmic_GB/s"] / 8000.0, 4)
        if copy_ceiling:
            r["step_algorithmic_frac_of_measured"] = round(r["step_algorithmic_GB/s"] / copy_ceiling, 4)
    return r


def load_reference_build():
    """oracle/_ref: the reference's own ORBextractor.cc and DepthModule.cc, compiled unmodified against the OpenCV stand-in
    (oracle/Makefile `ref`; prebuilt files travel to the GPU box).  None when they are not there."""
    ext = os.path.join(ROOT, "oracle", "_ref", "libref_orbextractor.so")
    dep = os.path.join(ROOT, "oracle", "_ref", "libref_depthmodule.so")
    yaml = os.path.join(ROOT, "tests", "golden", "KITTI00-02.yaml")
    if not (os.path.exists(ext) and os.path.exists(dep) and os.path.exists(yaml)):
        return None
    try:
        le, ld = C.CDLL(ext), C.CDLL(dep)
        le.ref_extract.restype = C.c_int
        le.ref_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        ld.ref_depth_create.restype = C.c_void_p
        ld.ref_depth_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        ld.ref_depth_compute.restype = C.c_int
        ld.ref_depth_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        a, b = C.c_int(0), C.c_int(0)
        pm = np.zeros((3, 4), np.float32)
        devnull = os.open(os.devnull, os.O_WRONLY)   # the reference constructor prints its matrices
        saved = os.dup(1)
        os.dup2(devnull, 1)
        try:
            hd = ld.ref_depth_create(yaml.encode(), 5, C.byref(a), C.byref(b), pm.ctypes.data)
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        if not (hd and a.value and b.value):
            return None
        return le, ld, hd
    except OSError:
        return None


def cpu_baseline(seq_frames, scans, proj, w, h, nfeatures, budget_s=20.0, use_reference=True):
    """CPU baseline on a bounded sample of the same workload, single thread.  Extraction and depth run the reference's own
    source files (oracle/_ref) when that build is present and the workload is the KITTI one its settings file describes;
    otherwise the oracle (CPU restatement).  Matching (all-pairs Hamming) is the oracle in both cases.
    Returns (frames/s, frames done, [ms per frame of extract, depth, match], kind)."""
    from oracle import oracle_py as O
    ref = load_reference_build() if use_reference else None
    ex = O.Extractor(nfeatures, SCALE, LEVELS, INI_TH, MIN_TH)
    P = O.make_depth_params(proj)
    cap = 4 * nfeatures + 4096
    t0 = time.perf_counter()
    n = 0
    prev = None
    stage = [0.0, 0.0, 0.0]
    for i in range(len(seq_frames)):
        img = seq_frames[i]
        cloud = scans[i % len(scans)]
        a = time.perf_counter()
        if ref:
            kps = np.zeros(cap, O.KP_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            nk = C.c_int(0)
            ref[0].ref_extract(img.ctypes.data, w, h, img.strides[0], nfeatures, SCALE, LEVELS, INI_TH, MIN_TH, 0, 0,
                               kps.ctypes.data, desc.ctypes.data, cap, C.byref(nk))
            kps, desc = kps[:nk.value], desc[:nk.value]
        else:
            kps, desc, _ = ex(img)
        b = time.perf_counter()
        kp_xy = np.ascontiguousarray(np.stack([kps["x"], kps["y"]], 1), np.float32)
        if ref:
            kx = np.ascontiguousarray(kps["x"], np.float32)
            dd, ur = np.zeros(len(kx), np.float32), np.zeros(len(kx), np.float32)
            cl = np.ascontiguousarray(cloud, np.float32)
            ref[1].ref_depth_compute(ref[2], cl.ctypes.data, cl.shape[1], cl.strides[0] // 4, w, h, kp_xy.ctypes.data, kx.ctypes.data,
                                     len(kx), dd.ctypes.data, ur.ctypes.data, None, None)
        else:
            O.depth(P, cloud, w, h, kp_xy, kps["x"], want_maps=False)
        c = time.perf_counter()
        if prev is not None:
            O.hamming_bf(prev, desc)
        d = time.perf_counter()
        prev = desc
        stage[0] += b - a
        stage[1] += c - b
        stage[2] += d - c
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return n / dt, n, [s / n * 1e3 for s in stage], ("reference" if ref else "port")


def cpu_worker(spec):
    """`python bench.py --cpu-worker CORE,WORKLOAD,BUDGET_S`: one process of the all-cores CPU baseline (leg (ii)): pins itself
    to a core, synthesises its own frames + scans of the workload and loops the single-thread baseline over them for the
    budget.  No torch, no HIP.  Prints one JSON line."""
    core, workload, budget = spec.split(",")
    core, budget = int(core), float(budget)
    try:
        os.sched_setaffinity(0, {core})
    except (AttributeError, OSError):
        pass
    from orb_slam3_rgbl_amd import synth
    from oracle import oracle_py as O
    w, h, nfeatures, n_az, _ = WORKLOADS[workload]
    K = synth.KITTI_K
    if workload != "kitti":
        K = synth.KITTI_K.copy()
        K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
        K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    proj = O.projection_matrix(K, synth.KITTI_TR)
    seq = synth.Sequence(1000 + core, w, h, n_frames=2)
    frames = np.stack([seq.frame(i) for i in range(2)])
    scans = [synth.lidar_scan(5000 + core, n_az=n_az)]
    fps, n, stage_ms, kind = cpu_baseline(list(frames) * 4096, scans, proj, w, h, nfeatures, budget, use_reference=workload == "kitti")
    print(json.dumps({"core": core, "frames": n, "fps": fps, "kind": kind}))


def cpu_baseline_all_cores(workload, budget_s=12.0):
    """SURVEY 8(d) CPU baseline leg (ii): every host core runs the single-thread baseline (leg (i)'s code) on its own
    independent frames at the same time - one process per core, pinned, started from fresh interpreters so that nothing of
    the GPU runtime is shared.  Returns (aggregate frames/s = sum of the per-core rates, cores used, frames done)."""
    import subprocess
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%s,%g" % (c, workload, budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT) for c in cores]
    fps = frames = used = 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=budget_s * 6 + 120)
            d = json.loads(out.strip().splitlines()[-1])
            fps += d["fps"]; frames += d["frames"]; used += 1
        except Exception:
            pr.kill()
    return fps, used, frames


def time_steps(step, sync, warmup, steps):
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return (time.perf_counter() - t0) / steps


def gather_legs_one_rank(lib, torch, dist, dev, d_imgs, d_cloud, w, h, nfeatures, proj, n_points, B, steps):
    """the gather inside timed steps on the hardware with ONE rank - every leg a fresh pipeline on the
    headline's resident inputs, same number of steps, one after the other in this process:
      none            no gather (the headline's configuration at N = 1)
      abi_step/final  the library's rgbl_gather_* with a one-rank RCCL communicator, loopback on: ncclAllGather of the counts
                      and a grouped ncclSend / ncclRecv of the rank's own records, queued on the low-priority stream of the scan
      torch_step/final  torch.distributed with a one-rank nccl process group: all_gather + (no peer: device copy); the
                      collectives run on ProcessGroupNCCL's internal stream - the 'fifth stream' of DESIGN 9
    No scaling claim follows from any of this: one rank has no peer, the bytes never leave the GPU."""
    import socket

    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline, make_comm
    made_pg = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        made_pg = True
    comm = make_comm(lib, dist, dev.index)
    ver = C.c_int(0)
    lib.rgbl_comm_info(comm, None, None, None, C.byref(ver))
    out = {"steps": steps, "frames_per_step": B, "rccl_version": ver.value, "unit": "frames/s"}
    try:
        for name, gather, transport, cm in (("none", "none", "abi", None), ("abi_step", "step", "abi", comm), ("abi_final", "final", "abi", comm),
                                            ("torch_step", "step", "torch", None), ("torch_final", "final", "torch", None)):
            pipe = FrontEndPipeline(lib, torch, dev, w, h, nfeatures, proj, n_points, B, levels=LEVELS, scale=SCALE, ini_th=INI_TH,
                                    min_th=MIN_TH, world=1, rank=0, gather=gather, log_steps=steps + 3, transport=transport, comm=cm,
                                    loopback=cm is not None)
            pipe.set_inputs(d_imgs, d_cloud)
            for _ in range(3):
                pipe.step()
            pipe.finish()
            pipe.sync()
            if gather == "final":   # the slots are per step of the run: start the timed run from slot 0 again
                pipe.step_no = 0
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe.step()
            pipe.finish()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / steps
            pipe.sync()
            out[name] = round(B / dt, 1)
            pipe.close()
            if os.environ.get("RGBL_BENCH_VERBOSE"):
                sys.stderr.write("gather leg %s: %.1f frames/s\n" % (name, out[name]))
        out["cost_of_abi_step"] = round(1.0 - out["abi_step"] / out["none"], 4)
        out["cost_of_torch_step"] = round(1.0 - out["torch_step"] / out["none"], 4)
        out["what"] = ("one rank, collectives inside the timed steps: rgbl_gather_* over RCCL on the scan's low-priority stream (abi) "
                       "against torch.distributed / ProcessGroupNCCL on its internal stream (torch); no peer, no scaling claim")
    finally:
        lib.rgbl_comm_destroy(comm)
        if made_pg:
            dist.destroy_process_group()
    return out


def spot_check(O, orc, P, frames, cloud, out_set, spot_frames, w, h, n_next=None):
    """Parity spot check of what a timed pipeline produced: the given frames of the step's output set (keypoints, descriptors,
    depth, uRight, matches against the following frame) against the CPU oracle, bit for bit.  Returns the JSON line's string."""
    h_n = out_set.n.cpu().numpy()
    ok = True
    n_next = n_next or len(frames)
    for fi in spot_frames:
        n = int(h_n[fi])
        kps = out_set.kp[fi, :n].cpu().numpy().view(np.uint32)
        okps, odesc, _ = orc(frames[fi])
        ok &= n == len(okps) and np.array_equal(kps, okps.view(np.uint32).reshape(n, 7))
        ok &= np.array_equal(out_set.desc[fi, :n].cpu().numpy(), odesc)
        od, our, _, _ = O.depth(P, cloud[fi], w, h, np.stack([okps["x"], okps["y"]], 1), okps["x"], want_maps=False)
        ok &= np.array_equal(out_set.depth[fi, :n].cpu().numpy().view(np.uint32), od.view(np.uint32))
        ok &= np.array_equal(out_set.uright[fi, :n].cpu().numpy().view(np.uint32), our.view(np.uint32))
        nxt = (fi + 1) % n_next     # chunks: the last owned frame meets the halo frame
        ndesc = orc(frames[nxt])[1]
        obi, obd, osd = O.hamming_bf(odesc, ndesc)
        ok &= np.array_equal(out_set.bi[fi, :n].cpu().numpy(), obi) and np.array_equal(out_set.bd[fi, :n].cpu().numpy(), obd)
        ok &= np.array_equal(out_set.sd[fi, :n].cpu().numpy(), osd)
    return ("bit-exact vs CPU oracle on frames %s (keypoints, descriptors, depth, uRight, matches)" %
            ", ".join(str(v) for v in spot_frames)) if ok else "MISMATCH vs CPU oracle"


def serial_kernel_leg(pipe, n_steps):
    """The per-kernel timing leg: `n_steps` serialised steps (one stream for everything, HIP-event brackets around every
    launch) queued back to back - no host synchronisation between them: a GPU that runs dry makes the brackets of a step's
    first launches measure the host's launch latency - and every launch's own duration read back afterwards
    (rgbl_*_profile_samples).  Returns ({kernel: (ms summed over the steps, launches)} built from the per-step MEDIAN, so that
    one slow step does not move it, and {kernel: {"median", "min", "max"} in ms per step})."""
    pipe.serialise()
    pipe.gather = "none"
    pipe.profile(True)
    for _ in range(n_steps):
        pipe.step()
    pipe.sync()
    samples = pipe.profile_samples()
    pipe.profile(False)
    kernels, stats = {}, {}
    for k in sorted(samples):
        v = samples[k]
        per = len(v) // n_steps               # launches per step (7 for the resize, 1 otherwise)
        if per == 0:
            continue
        by_step = [sum(v[i * per:(i + 1) * per]) for i in range(n_steps)]
        ms = sorted(by_step)
        med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
        kernels[k] = (med * n_steps, per * n_steps)
        stats[k] = {"median": round(med, 4), "min": round(ms[0], 4), "max": round(ms[-1], 4), "max_at_step": by_step.index(ms[-1])}
    return kernels, stats


def natural_frames(n, w, h):
    """KITTI-size frames cut from a canvas tiled with the two committed photographs (tests/golden/natural_camera.png and
    natural_brick.png, 512 x 512 grey, CC0 scikit-image samples; alternate tiles mirrored so that no seam adds an edge),
    frame t = the window shifted by (3 t, t) px - the motion of synth.Sequence.  None when PIL or the files are missing."""
    try:
        from PIL import Image
        imgs = [np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "natural_%s.png" % k)).convert("L"), np.uint8) for k in ("camera", "brick")]
    except Exception:
        return None
    cw, ch = w + 3 * n + 8, h + n + 8
    rows = []
    for ty in range((ch + 511) // 512):
        row = []
        for tx in range((cw + 511) // 512):
            t = imgs[(tx // 2 + ty // 2) % 2]
            t = t[:, ::-1] if tx % 2 else t
            t = t[::-1] if ty % 2 else t
            row.append(t)
        rows.append(np.concatenate(row, 1))
    canvas = np.concatenate(rows, 0)[:ch, :cw]
    return np.stack([np.ascontiguousarray(canvas[t:t + h, 3 * t:3 * t + w]) for t in range(n)])


def prescreen_stats(img, t=INI_TH):
    """Share of a level-0 image's pixels that pass k_fast_cells' necessary-condition test (each of the 4 opposite pairs at the
    even ring positions has a darker / brighter member) and that really carry a 9-arc (numpy; one threshold)."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    hh, ww = img.shape
    I = img.astype(np.int16)
    c = I[3:hh - 3, 3:ww - 3]
    R = [I[3 + dy:hh - 3 + dy, 3 + dx:ww - 3 + dx] for dx, dy in ring]
    out = []
    for F in ([r < c - t for r in R], [r > c + t for r in R]):
        p4 = np.ones_like(F[0])
        for k in (0, 2, 4, 6):
            p4 &= F[k] | F[k + 8]
        arc = np.zeros_like(F[0])
        for s0 in range(16):
            a = np.ones_like(F[0])
            for j in range(9):
                a &= F[(s0 + j) % 16]
            arc |= a
        out.append((p4, arc))
    return {"prescreen_survivors": round(float((out[0][0] | out[1][0]).mean()), 4), "pixels_with_a_9_arc": round(float((out[0][1] | out[1][1]).mean()), 4)}


def natural_texture_leg(lib, dev, torch, synth_frames, cloud_np, proj, n_points, synth_rate):
    """extra.natural_texture: the headline step on frames tiled from photographs instead of the SURVEY 8(d) generator - how
    far the synthetic corner density is from natural imagery, and what it does to the rate.  Never `value`."""
    from orb_slam3_rgbl_amd import frontend as F
    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline
    from oracle import oracle_py as O
    w, h, nf, _, _ = WORKLOADS["kitti"]
    B = 256
    frames_np = natural_frames(B, w, h)
    if frames_np is None:
        return {"skipped": "PIL or tests/golden/natural_*.png not available"}
    cl = np.ascontiguousarray(cloud_np[:B]) if len(cloud_np) >= B else np.stack([cloud_np[i % len(cloud_np)] for i in range(B)])
    pipe = FrontEndPipeline(lib, torch, dev, w, h, nf, proj, n_points, B, levels=LEVELS, scale=SCALE, ini_th=INI_TH, min_th=MIN_TH, gather="none")
    d_f, d_c = torch.from_numpy(frames_np).to(dev), torch.from_numpy(cl).to(dev)
    pipe.set_inputs(d_f, d_c)
    dt = time_steps(pipe.step, pipe.sync, 3, 20)
    k_mean = float(pipe.last().n.float().mean().item())
    spot = spot_check(O, O.Extractor(nf, SCALE, LEVELS, INI_TH, MIN_TH), O.make_depth_params(proj), frames_np, cl, pipe.last(), [0, B - 1], w, h)
    # the same B on the synthetic frames, same call, for the ratio
    pipe.set_inputs(torch.from_numpy(np.ascontiguousarray(synth_frames[:B])).to(dev), d_c)
    dt_s = time_steps(pipe.step, pipe.sync, 3, 20)
    pipe.close()
    # FAST candidates of level 0 (what cv::FAST returns over all cells, before the quad-tree) on a few frames, single-frame handle
    ex = F.ORBextractor(nf, SCALE, LEVELS, INI_TH, MIN_TH, w, h, device=dev.index, lib=lib)
    cand = {}
    for name, fr in (("natural", frames_np), ("synthetic", synth_frames)):
        c = []
        for fi in (0, B // 2, B - 1):
            ex(np.ascontiguousarray(fr[fi]))
            c.append(len(ex.level_candidates(0)))
        cand[name] = float(np.mean(c))
    ex.close()
    return {"frames_per_s": B / dt, "ms_per_step": dt * 1e3, "frames_per_step": B, "keypoints_per_frame": k_mean,
            "synthetic_frames_per_s_same_call": B / dt_s, "natural_over_synthetic_rate": round(dt_s / dt, 4),
            "level0_fast_candidates_per_frame": cand,
            "level0_pixel_shares_at_iniThFAST": {"natural": prescreen_stats(frames_np[B // 2]), "synthetic": prescreen_stats(synth_frames[min(B // 2, len(synth_frames) - 1)])},
            "parity_spot_check": spot, "headline_frames_per_s": synth_rate,
            "what": "extract + depth + match of 256 KITTI-size frames cut from a canvas tiled with tests/golden/natural_camera.png / natural_brick.png "
                    "(mirrored tiles, window shifted by (3 t, t) px), the synthetic scans; beside it the same 256-frame step on the first 256 "
                    "synthetic frames of the headline.  The headline stays configs[1] on the SURVEY 8(d) generator"}


def extra_workloads(lib, dev, torch, copy_ceiling=None):
    """Driver-visible figures for the other GPU configurations of BASELINE.json, measured in the same run on the same
    device (short runs; `value` stays the KITTI RGB-L configuration): configs[4] (4K frames + 262 144-point scans,
    nFeatures 8000), configs[2] (KITTI stereo: two extractions + Frame::ComputeStereoMatches), and the latency of the
    host-pointer (drop-in) entry points, one frame per call, PCIe and synchronisation included."""
    from orb_slam3_rgbl_amd import _lib as L
    from orb_slam3_rgbl_amd import frontend as F
    from orb_slam3_rgbl_amd import synth
    out = {}

    def p(t):
        return C.c_void_p(t.data_ptr())

    def sync():
        torch.cuda.synchronize(dev)

    # ---- configs[4]: 4K (the step of pipeline.py on 4K frames; then its serialised per-kernel leg for the roofline object)
    try:
        from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline
        w, h, nf, n_az, B = WORKLOADS["4k"]
        # 64 DISTINCT frames and scans (round 4 repeated 4 frames and one scan 16 x / 64 x)
        seq = synth.Sequence(7, w, h, n_frames=B)
        frames_np = np.stack([seq.frame(i) for i in range(B)])
        frames = torch.from_numpy(frames_np).to(dev)
        cloud_np = np.stack([synth.lidar_scan(7000 + i, n_az=n_az) for i in range(B)])
        n_points = cloud_np.shape[2]
        cloud = torch.from_numpy(cloud_np).to(dev)
        K = synth.KITTI_K.copy()
        K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
        K[0, 2], K[1, 2] = w / 2.0, h / 2.0
        proj = F.projection_matrix(K, synth.KITTI_TR, lib)
        pipe = FrontEndPipeline(lib, torch, dev, w, h, nf, proj, n_points, B, levels=LEVELS, scale=SCALE, ini_th=INI_TH, min_th=MIN_TH,
                                gather="none")
        pipe.set_inputs(frames, cloud)
        dt = time_steps(pipe.step, pipe.sync, 2, 6)
        k_mean = float(pipe.last().n.float().mean().item())
        # parity of exactly what was timed: the batched path (k_compact_cells, XCD grid, k_octree<1024, 2048> on the count pyramid)
        # on the first and the last frame of the last timed step, against the CPU oracle - the headline's check, same wording
        from oracle import oracle_py as O
        spot = spot_check(O, O.Extractor(nf, SCALE, LEVELS, INI_TH, MIN_TH), O.make_depth_params(proj), frames_np, cloud_np, pipe.last(),
                          [0, B - 1], w, h)
        kernels, kstats = serial_kernel_leg(pipe, 4)
        out["cfg5_4k"] = {"frames_per_s": B / dt, "ms_per_step": dt * 1e3, "frames_per_step": B, "image": [w, h], "nfeatures": nf,
                          "lidar_points": n_points, "keypoints_per_frame": k_mean, "distinct_frames_and_scans": B,
                          "parity_spot_check": spot,
                          "what": "extract + depth + match, inputs resident in HBM, 1 GPU",
                          "roofline": roofline_of(kernels, 4, "4k", w, h, n_points, k_mean, B, dt, copy_ceiling)}
        out["cfg5_4k"]["roofline"]["kernels_ms_per_step_stats"] = kstats
        pipe.close()
        del frames, cloud, pipe, frames_np, cloud_np
    except Exception as e:  # never lose the main line over an extra figure
        out["cfg5_4k"] = {"error": repr(e)}

    # ---- configs[2]: KITTI stereo front end
    try:
        w, h, B = synth.KITTI_W, synth.KITTI_H, 64
        sseq = synth.Sequence(40, w + 128, h, n_frames=B)
        lefts, rights = [], []
        for i in range(B):  # 64 distinct rectified pairs: the right view is the scene shifted by a row-dependent disparity of 4 .. 40 px
            full = sseq.frame(i)
            lefts.append(np.ascontiguousarray(full[:, 64:64 + w]))
            r = np.empty((h, w), np.uint8)
            for y in range(h):
                d = int(round(4 + 36.0 * y / h))
                r[y] = full[y, 64 + d:64 + d + w]
            rights.append(r)
        dl = torch.from_numpy(np.stack(lefts)).to(dev)
        dr = torch.from_numpy(np.stack(rights)).to(dev)
        exl = F.ORBextractor(2000, SCALE, LEVELS, 20, 7, w, h, max_batch=B, device=dev.index, lib=lib)
        exr = F.ORBextractor(2000, SCALE, LEVELS, 20, 7, w, h, max_batch=B, device=dev.index, lib=lib)
        cap = exl.max_keypoints

        def outs():
            return (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                    torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
        kl, ddl, nl, ml = outs()
        kr, ddr, nr, mr = outs()
        ur = torch.zeros((B, cap), dtype=torch.float32, device=dev)
        dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)

        def step():
            L.check(lib, lib.rgbl_extract_batch_device(exl.h, p(dl), B, w, h, w, w * h, 0, 0, p(kl), p(ddl), cap, p(nl), p(ml)))
            L.check(lib, lib.rgbl_extract_batch_device(exr.h, p(dr), B, w, h, w, w * h, 0, 0, p(kr), p(ddr), cap, p(nr), p(mr)))
            L.check(lib, lib.rgbl_stereo_matches_batch_device(exl.h, exr.h, B, p(kl), p(ddl), p(nl), p(kr), p(ddr), p(nr), cap,
                                                              0.54, 386.1448, p(ur), p(dp)))
        dt = time_steps(step, sync, 2, 8)
        # parity of what was timed: both extractions and Frame::ComputeStereoMatches of the first and the last pair vs the CPU oracle
        from oracle import oracle_py as O
        ok = True
        for fi in (0, B - 1):
            ol, orr = O.Extractor(2000, SCALE, LEVELS, 20, 7), O.Extractor(2000, SCALE, LEVELS, 20, 7)
            kL, dL, _ = ol(lefts[fi])
            kR, dR, _ = orr(rights[fi])
            n_l, n_r = int(nl[fi].item()), int(nr[fi].item())
            ok &= n_l == len(kL) and n_r == len(kR)
            ok &= np.array_equal(kl[fi, :n_l].cpu().numpy().view(np.uint32), kL.view(np.uint32).reshape(len(kL), 7))
            ok &= np.array_equal(kr[fi, :n_r].cpu().numpy().view(np.uint32), kR.view(np.uint32).reshape(len(kR), 7))
            ok &= np.array_equal(ddl[fi, :n_l].cpu().numpy(), dL) and np.array_equal(ddr[fi, :n_r].cpu().numpy(), dR)
            our, odp = O.stereo_matches(ol, orr, kL, dL, kR, dR, 0.54, 386.1448)
            ok &= np.array_equal(ur[fi, :n_l].cpu().numpy().view(np.uint32), np.asarray(our, np.float32).view(np.uint32))
            ok &= np.array_equal(dp[fi, :n_l].cpu().numpy().view(np.uint32), np.asarray(odp, np.float32).view(np.uint32))
        out["cfg3_stereo"] = {"stereo_frames_per_s": B / dt, "ms_per_step": dt * 1e3, "pairs_per_step": B, "image": [w, h],
                              "distinct_pairs": B,
                              "parity_spot_check": ("bit-exact vs CPU oracle on pairs 0, %d (keypoints and descriptors of both views, uRight, depth)" % (B - 1))
                                                   if ok else "MISMATCH vs CPU oracle",
                              "nfeatures": 2000, "fast_thresholds": [20, 7], "stereo_matches_per_frame": float((dp > 0).float().sum().item() / B),
                              "what": "left + right extraction + Frame::ComputeStereoMatches on the resident pyramids, 1 GPU"}
        exl.close(); exr.close()
    except Exception as e:
        out["cfg3_stereo"] = {"error": repr(e)}

    # ---- drop-in (host-pointer) entry points, one RGB-L frame per call
    try:
        w, h = synth.KITTI_W, synth.KITTI_H
        seq = synth.Sequence(3, w, h, n_frames=4)
        imgs = [seq.frame(i) for i in range(4)]
        scan = synth.lidar_scan(3000)
        ex = F.ORBextractor(2000, SCALE, LEVELS, INI_TH, MIN_TH, w, h, device=dev.index, lib=lib)
        dm = F.DepthModule(F.projection_matrix(synth.KITTI_K, synth.KITTI_TR, lib), w, h, max_points=scan.shape[1],
                           max_keypoints=ex.max_keypoints, device=dev.index, lib=lib)
        mt = F.ORBmatcher(0.6, False, device=dev.index, lib=lib)
        prev = None
        best = None
        for rep in range(4):
            t = [0.0, 0.0, 0.0]
            cnt = 0
            for img in imgs * 3:
                a = time.perf_counter(); kps, desc, _ = ex(img)
                b = time.perf_counter(); dm.CalculateDepthFromPcd(kps, kps, scan, w, h, want_maps=False)
                c = time.perf_counter()
                if prev is not None:
                    mt.BruteForce(prev, desc)
                d = time.perf_counter()
                prev = desc
                t[0] += b - a; t[1] += c - b; t[2] += d - c; cnt += 1
            cur = [v / cnt * 1e3 for v in t]
            if best is None or sum(cur) < sum(best):
                best = cur
        # the same frame with the optional latency hooks: upload + maps of the scan next to the extraction
        scan32 = np.ascontiguousarray(scan, np.float32)
        best_o = None
        for rep in range(4):
            a = time.perf_counter()
            cnt = 0
            for img in imgs * 3:
                ex.Begin(img)
                dm.PrefetchPointcloud(scan32, w, h)
                kps, desc, _ = ex(img)
                dm.CalculateDepthFromPcd(kps, kps, scan32, w, h, want_maps=False)
                mt.BruteForce(prev, desc)
                prev = desc
                cnt += 1
            cur = (time.perf_counter() - a) / cnt * 1e3
            best_o = cur if best_o is None else min(best_o, cur)
        out["host_api_single_frame"] = {"ms_per_frame": {"extract": best[0], "depth": best[1], "match": best[2], "total": sum(best),
                                                         "total_with_begin_prefetch": best_o},
                                        "frames_per_s": 1e3 / sum(best),
                                        "what": "rgbl_extract + rgbl_depth_compute + rgbl_hamming_bf with host pointers: H2D, kernels, D2H, "
                                                "synchronous return - what a drop-in System::TrackRGBL sees; never `value`.  total_with_begin_prefetch: the same three "
                                                "calls behind rgbl_extract_begin + rgbl_depth_prefetch (two extra lines in Frame's RGB-L constructor, "
                                                "INTEGRATION.md): scan upload, projection and up-sampling run next to the extraction"}
        ex.close(); dm.close(); mt.close()
        # the same frame loop from C++, through the drop-in classes (tools/shim_latency.cpp, built and run as a child process):
        # what ORB_SLAM3 itself would see - the figures above carry ~40 us of ctypes / numpy per frame
        try:
            import re
            import subprocess
            res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shim_latency.py")], capture_output=True, text=True, timeout=180)
            m = re.findall(r"(C\+\+ drop-in classes[^:]*): ([0-9.]+) ms per frame \(extract ([0-9.]+), depth ([0-9.]+), match ([0-9.]+)\)", res.stdout)
            if len(m) == 2:
                out["host_api_single_frame"]["cpp_drop_in_classes"] = {
                    "ms_per_frame": {"extract": float(m[0][2]), "depth": float(m[0][3]), "match": float(m[0][4]), "total": float(m[0][1]),
                                     "total_with_begin_prefetch": float(m[1][1])},
                    "what": "tools/shim_latency.cpp: ORB_SLAM3::ORBextractor / DepthModule (shim/) + rgbl_hamming_bf from C++, one frame per call"}
        except Exception as e:
            out["host_api_single_frame"]["cpp_drop_in_classes"] = {"error": repr(e)}
    except Exception as e:
        out["host_api_single_frame"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 0.2 s timed: the boxes of the pool differ by more than a 20-step run resolves
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (default: workload specific)")
    ap.add_argument("--workload", default="kitti", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the 4K / stereo / single-frame figures (extra keys of the JSON line)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--serial", action="store_true", help="one stream for all handles (clean per-kernel timings)")
    ap.add_argument("--gather", default=None, choices=["step", "final", "none"],
                    help="gather of the keypoint / descriptor / depth records to rank 0: stream every step's records while the "
                         "next step computes (default for N > 1), exchange all of them once at the end, or not at all (default "
                         "for N = 1; step / final then run the same choreography without a peer)")
    ap.add_argument("--transport", default="abi", choices=["abi", "torch"],
                    help="what carries the gather: the library's own C-ABI entry points over RCCL (rgbl_gather_*, csrc/gather.hip: "
                         "ncclAllGather + grouped ncclSend / ncclRecv on the pipeline's low-priority stream; default) or "
                         "torch.distributed (ProcessGroupNCCL, which runs its collectives on an internal stream of its own)")
    ap.add_argument("--pg", action="store_true",
                    help="N = 1: create the one-rank process group / RCCL communicator anyway and gather (default --gather step), so "
                         "that the collectives - and with --transport torch ProcessGroupNCCL's internal stream - are inside the timed steps")
    ap.add_argument("--shard", default="sequences", choices=["sequences", "chunks"],
                    help="sequences: one independent sequence per rank (BASELINE configs[3], the default).  chunks: ONE long sequence "
                         "of world x batch frames cut into contiguous chunks (sharding.frame_chunk), every rank also extracts the "
                         "frame behind its chunk (a one-frame halo, recomputed, never communicated) so that its last frame is "
                         "matched against its true successor - SURVEY 8(e) 'single long sequence'")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)

    import torch
    import torch.distributed as dist

    # ---- who runs: this process as one rank, or N ranks under torch.distributed.run (a bare `--gpus N` launches itself)
    plan = launch_plan(args.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0, sys.argv)
    if plan[0] == "error":
        raise SystemExit("bench.py: " + plan[1])
    if plan[0] == "exec":
        sys.stdout.flush()
        os.execv(plan[1][0], plan[1])
    world = plan[1]

    from orb_slam3_rgbl_amd import _lib as L
    from orb_slam3_rgbl_amd import frontend as F
    from orb_slam3_rgbl_amd import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w, h, nfeatures, n_az, default_batch = WORKLOADS[args.workload]
    B = args.batch or default_batch
    lib = L.load()
    gather = args.gather or ("step" if (world > 1 or args.pg) else "none")
    comm = None
    if gather != "none" and args.transport == "abi" and dist.is_initialized():
        from orb_slam3_rgbl_amd.pipeline import make_comm
        comm = make_comm(lib, dist, local_rank)   # rgbl_comm_create: the framework only hands the unique id around

    # ---- synthetic input, one independent sequence per rank (BASELINE configs[3]: sequences shard over GPUs)
    # constant_density: 1200 shapes per frame area whatever the batch (the scene grows with the sequence length)
    halo = 1 if args.shard == "chunks" else 0
    if halo:
        # one sequence for all ranks; rank r owns the frames [r B, (r + 1) B) and recomputes the first frame of the next chunk
        # (the last rank wraps around to frame 0, so that every rank does the same amount of work)
        total = world * B
        seq = synth.Sequence(0, w, h, n_frames=total, constant_density=not os.environ.get("RGBL_BENCH_SPARSE"))
        first, last, _ = sharding.frame_chunk(total, world, rank)
        assert last - first == B
        frames = np.stack([seq.frame((first + i) % total) for i in range(B + 1)])
    else:
        seq = synth.Sequence(sharding.sequences_of_rank(world, world, rank)[0], w, h, n_frames=B, constant_density=not os.environ.get("RGBL_BENCH_SPARSE"))
        frames = np.stack([seq.frame(i) for i in range(B)])
    n_scans = min(B, 8)
    scans = [synth.lidar_scan(rank * 1000 + i, n_az=n_az) for i in range(n_scans)]
    n_points = scans[0].shape[1]
    cloud = np.stack([scans[i % n_scans] for i in range(B + halo)])  # [B (+ halo), 4, N]
    if args.workload == "kitti":
        K = synth.KITTI_K
    else:
        K = synth.KITTI_K.copy()
        K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
        K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)

    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline
    # the step (extract -> depth -> match on resident inputs) and the gather of its records live in
    # orb_slam3_rgbl_amd/pipeline.py - the same code tests/test_distributed.py drives with two gloo ranks
    pipe = FrontEndPipeline(lib, torch, dev, w, h, nfeatures, proj, n_points, B, levels=LEVELS, scale=SCALE, ini_th=INI_TH,
                            min_th=MIN_TH, world=world, rank=rank, gather=gather, serial=args.serial,
                            log_steps=args.steps + args.warmup, transport=args.transport, comm=comm,
                            loopback=(comm is not None and world == 1), halo=halo)
    ex, dm, mt, cap = pipe.ex, pipe.dm, pipe.mt, pipe.cap
    d_imgs = torch.from_numpy(frames).to(dev)
    d_cloud = torch.from_numpy(cloud).to(dev)
    pipe.set_inputs(d_imgs, d_cloud)
    step, sync_all = pipe.step, pipe.sync

    for _ in range(args.warmup):
        step()
    sync_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.finish()   # the records still on their way to rank 0 belong to the job
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    sync_all()
    multi = None
    if world > 1 or dist.is_initialized():
        # what an N > 1 line needs so that "RCCL saw N ranks" is on record, not taken on trust: the communicator's own view of the
        # world, every rank's own rate (its clock around the same barriers), and what the root actually received per step
        own = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(every, own)
        per_rank = [float(v.item()) for v in every]
        elapsed = max(per_rank)
        multi = {"per_rank_frames_per_s": [round(B * args.steps / v, 1) for v in per_rank]}
        cw, cr, cv = C.c_int(0), C.c_int(0), C.c_int(0)
        if comm is not None:
            lib.rgbl_comm_info(comm, C.byref(cw), C.byref(cr), None, C.byref(cv))
            multi["rccl"] = {"world": cw.value, "rank_of_this_process": cr.value, "version": cv.value,
                             "source": "rgbl_comm_info of the communicator the gather ran on (ncclCommInitRank through the C ABI)"}
        else:
            multi["rccl"] = {"world": dist.get_world_size(), "version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                             "source": "torch.distributed process group (backend nccl = RCCL)"}
        if rank == 0 and gather != "none" and pipe.received:
            got = pipe.received[-1]     # the last exchange: per rank (counts [B], records)
            per_rank_bytes = [int(np.minimum(np.maximum(c, 0), cap).sum()) * 68 for c, _ in got]
            multi["gathered_bytes_per_step"] = int(sum(per_rank_bytes))
            multi["gathered_bytes_per_step_per_rank"] = per_rank_bytes
            multi["root_ingest_GB/s"] = round(sum(per_rank_bytes[1:]) / (elapsed / args.steps) / 1e9, 2)
            multi["root_ingest_what"] = ("bytes of the ranks 1 .. N - 1 arriving at rank 0 per step over the step time (one xGMI link per "
                                         "peer, exact-size ncclRecv each); the root's own records are a device copy")

    last = pipe.last()
    d_n, d_kp, d_desc, d_depth, d_uright, d_bi, d_bd, d_sd = (last.n, last.kp, last.desc, last.depth, last.uright, last.bi,
                                                               last.bd, last.sd)
    k_mean = float(d_n.float().mean().item())

    # ---- parity spot check of what the timed pipeline produced (four frames against the CPU oracle)
    spot = None
    if rank == 0:
        from oracle import oracle_py as O
        orc = O.Extractor(nfeatures, SCALE, LEVELS, INI_TH, MIN_TH)
        spot = spot_check(O, orc, O.make_depth_params(proj), frames, cloud, last, sorted({0, B // 3, 2 * B // 3, B - 1}), w, h)

    # ---- opt-in variant, reported beside the headline only: no dense ProcessedDepthMap (rgbl_depth_set_sparse), the keypoints'
    # depths come out of the index maps directly.  Same inputs; depth / uRight must not change by a bit.
    sparse_leg = None
    if world == 1 and not args.no_extras and not args.serial:
        ref_depth, ref_ur = d_depth.clone(), d_uright.clone()
        for ln in pipe.lanes:
            ln.dm.SetSparseUpsampling(True)
        n_sparse = max(10, args.steps // 2)
        dt = time_steps(step, lambda: (pipe.finish(), torch.cuda.synchronize(dev)), 3, n_sparse)
        sl = pipe.last()
        same = bool(torch.equal(sl.depth.view(torch.int32), ref_depth.view(torch.int32)) and
                    torch.equal(sl.uright.view(torch.int32), ref_ur.view(torch.int32)))
        for ln in pipe.lanes:
            ln.dm.SetSparseUpsampling(False)
        step()
        sync_all()
        sparse_leg = {"value": B / dt, "unit": "frames/s", "steps": n_sparse, "depth_uright_equal_to_dense_run": same,
                      "what": "the headline step with rgbl_depth_set_sparse(1): k_inverse_dilate is not launched, the gather evaluates "
                              "the dilation at the keypoints' pixels (the dense map has no reader in RGB-L tracking: Tracking.cc:1584 "
                              "copies it for FrameDrawer.cc:375, which is commented out). Not the headline: SURVEY 8(d) counts the "
                              "dense map's bytes"}

    # ---- rank 0 at N == 1: roofline of the dominant kernel (HIP events on the launch stream) + CPU baseline
    roofline = None
    cpu = None
    kernels = {}
    copy_ceiling = None
    if rank == 0:
        # per-kernel timing leg: one stream for everything, so that the HIP-event brackets around each launch are not
        # stretched by other kernels running concurrently
        prof_steps = PROF_STEPS
        kernels, kernel_stats = serial_kernel_leg(pipe, prof_steps)
        try:
            copy_ceiling = measured_copy_ceiling(torch, dev)
        except Exception as e:   # the line stands without it
            copy_ceiling = None
            sys.stderr.write("copy ceiling not measured: %r\n" % (e,))
        roofline = roofline_of(kernels, prof_steps, args.workload, w, h, n_points, k_mean, B, elapsed / args.steps, copy_ceiling)
        roofline["kernels_ms_per_step_stats"] = kernel_stats
        roofline["kernels_ms_per_step_what"] = ("median over %d serialised steps (HIP events on the launch stream, every step read back "
                                                "on its own), min / max beside it" % prof_steps)
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = min(B, 256)
            fps, n_done, stage_ms, kind = cpu_baseline(frames[:n_cpu], scans, proj, w, h, nfeatures, args.cpu_budget,
                                                       use_reference=args.workload == "kitti")
            what = ("oracle/_ref: the reference's own ORBextractor.cc and DepthModule.cc compiled unmodified against the OpenCV "
                    "stand-in (its cv:: primitives are scalar restatements, not OpenCV's SIMD code), matching = oracle all-pairs "
                    "Hamming" if kind == "reference" else "oracle/ (CPU restatement of the reference path)")
            fps_all = cores_all = n_all = None
            try:
                fps_all, cores_all, n_all = cpu_baseline_all_cores(args.workload, min(args.cpu_budget, 12.0))
            except Exception as e:  # the single-thread figure stands on its own
                fps_all = None
                sys.stderr.write("all-cores CPU baseline failed: %r\n" % (e,))
            cpu = {"value": fps, "unit": "frames/s", "cores": 1, "kind": kind,
                   "sample": "%d synthetic %dx%d frames + scans, extract+depth+match, %s, g++ -O2, 1 thread = the reference's "
                             "one-extractor-thread-per-image model" % (n_done, w, h, what),
                   "ms_per_frame": {"extract": stage_ms[0], "depth": stage_ms[1], "match": stage_ms[2]},
                   "host_cores_available": os.cpu_count(),
                   # SURVEY 8(d) leg (ii): one independent frame per core on all host cores at once (same code per core)
                   "all_cores": {"value": fps_all, "unit": "frames/s", "cores": cores_all, "frames": n_all,
                                 "sample": "one pinned process per core, each looping leg (i)'s code over its own synthetic frames + scan for "
                                           "~%d s at the same time; value = sum of the per-core rates" % min(args.cpu_budget, 12.0)}
                                if fps_all else None}

    if rank == 0:
        total_frames = world * B * args.steps
        out = {
            "metric": "RGB-L front-end frames/sec (extract+depth+match)",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "KITTI-00 RGB-L cfg2 (synthetic stand-in)" if args.workload == "kitti"
                       else "4K synthetic cfg5", "image": [w, h], "nfeatures": nfeatures, "levels": LEVELS,
                       "lidar_points": n_points, "frames_per_gpu_per_step": B, "keypoints_per_frame": k_mean,
                       "match": "Hamming brute force, frame i vs i+1", "upsampling": "InverseDilation Diamond 5",
                       "inputs": "resident in HBM",
                       "frames": "synth.Sequence, %s" % ("shape count of ONE frame for the whole scene (RGBL_BENCH_SPARSE: the pre-correction input)" if os.environ.get("RGBL_BENCH_SPARSE") else "constant corner density: 1200 shapes per frame area, ~8.6 k FAST candidates on level 0"),
                       "parallelism": ("one sequence of %d frames in contiguous chunks + one-frame halo, %d rank(s)" % (world * B, world)) if halo
                                      else "frames/sequences sharded, %d rank(s)" % world,
                       "gather_transport": None if gather == "none" else
                                           ("C ABI rgbl_gather_* over RCCL (ncclAllGather + grouped ncclSend/ncclRecv) on the pipeline's low-priority stream"
                                            if pipe.transport == "abi" and comm is not None else
                                            "C ABI rgbl_gather_* without a communicator (one rank: device copies)" if pipe.transport == "abi" else
                                            "torch.distributed (ProcessGroupNCCL)" if dist.is_initialized() else "no backend (one rank: device copies)"),
                       "gather": {"none": "none",
                                  "step": "step: every step's records (68 B per keypoint, packed on the device) go to rank 0 while the "
                                          "next step computes - counts by all-gather, records by exact-size send / recv, one xGMI link "
                                          "per peer; the volume (~70 MB per rank and step) is a tenth of what the links carry, exchanged "
                                          "once at the end (--gather final) it would be a serial tail of about a third of the compute time",
                                  "final": "final: the records of all steps stay packed in HBM and are exchanged once, inside the timed region"}[gather]},
            "parity_spot_check": spot,
            "multi_gpu": multi,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if world == 1 and not args.no_extras:
            out["extra"] = extra_workloads(lib, dev, torch, copy_ceiling)
            out["extra"]["sparse_upsampling"] = sparse_leg
            # BASELINE configs[2] (SearchForTriangulation) and the matcher calls of one tracked frame: per-call latency through the
            # host-pointer C ABI with the reference's own ORBmatcher.cc / DBoW2 / ComputeStereoMatches timed beside each (bench_calls.py)
            try:
                out["extra"]["natural_texture"] = natural_texture_leg(lib, dev, torch, frames, cloud, proj, n_points, total_frames / elapsed)
            except Exception as e:
                out["extra"]["natural_texture"] = {"error": repr(e)}
            import bench_calls
            for name, leg in (("cfg3_triangulation", bench_calls.triangulation_leg), ("tracking_calls", bench_calls.tracking_leg),
                              ("mapping_calls", bench_calls.mapping_leg)):
                try:
                    out["extra"][name] = leg(lib)
                except Exception as e:   # the headline stands on its own
                    out["extra"][name] = {"error": repr(e)}
            try:
                out["extra"]["gather_rccl_1rank"] = gather_legs_one_rank(lib, torch, dist, dev, d_imgs, d_cloud, w, h, nfeatures, proj, n_points, B,
                                                                         max(10, args.steps // 2))
            except Exception as e:   # the headline stands on its own
                out["extra"]["gather_rccl_1rank"] = {"error": repr(e)}
        line = json.dumps(out)
    # The JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio when the first communicator
    # is created, which a pipe only sees when the buffer is flushed - at exit, behind the line, unless it is flushed here.
    # Every rank flushes, then a barrier, then rank 0 prints.
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if world > 1:
        dist.barrier()
    if rank == 0:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if comm is not None:
        pipe.close()
        lib.rgbl_comm_destroy(comm)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
