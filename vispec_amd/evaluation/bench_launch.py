"""Launcher and host-side helpers of bench.py (round 5: split out of the bench script): request plan per (rank, lane, step), one host thread
per lane, self-launch of the N ranks of `python bench.py --gpus N`, NUMA pinning of a rank's lane threads, host usage of a rank.
Nothing here touches the hot path; the reference's counterpart is the Ray fan-out of evaluation/gen_spec_answer_coco_caption.py:54-83, 409-412."""
import os
import sys

import torch


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def request_plan(n_requests, rank, world, lanes, cohort, n_steps):
    """plan[lane][step] = ids of the requests that lane of this rank runs in that step (a lane takes them `cohort` at a time on one
    weight pass).  n_requests > 0: BASELINE config 4's fixed batch, request i -> replica i mod world (parallel.shard_requests), then
    lane by lane ("strong" scaling: the batch is fixed); 0: every (rank, lane) runs `cohort` requests of its own per step ("weak")."""
    from vispec_amd import parallel
    if n_requests > 0:
        mine = parallel.shard_requests(n_requests, rank, world)
        # fill cohorts before opening lanes: 8 requests on a rank are 2 lanes x cohorts of 4 (one weight pass per four requests), not
        # 4 lanes x pairs; the lanes that stay without requests do nothing
        used = max(1, min(lanes, -(-len(mine) // max(1, cohort))))
        return [[[i + s * n_requests for i in (mine[lane::used] if lane < used else [])] for s in range(n_steps)] for lane in range(lanes)], "strong"
    return [[[((rank * lanes + lane) + s * world * lanes) * cohort + j for j in range(cohort)] for s in range(n_steps)]
            for lane in range(lanes)], "weak"


def run_lanes(fns):
    """Run one callable per lane concurrently (one host thread + one HIP stream per lane); returns their results."""
    import threading
    out = [None] * len(fns)
    err = []

    def work(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:  # surface worker failures in the main thread
            err.append(e)

    if len(fns) == 1:
        work(0)
    else:
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
        [t.start() for t in ths]
        [t.join() for t in ths]
    if err:
        raise err[0]
    return out


def pin_to_gpu_numa_node(local):
    """One rank per GPU, `lanes` host threads per rank (each issues hipGraph launches and waits on events): keep them on the cores of the NUMA
    node the GPU hangs off, so that eight ranks on a two-socket node do not launch across the socket link (host-side launch
    contention is the one scaling risk of a replicas-only design).  PCI address from the device properties -> /sys/bus/pci/devices/<addr>/numa_node
    -> that node's cpulist -> os.sched_setaffinity (threads started later inherit it).  Anything missing (no NUMA information, a container that
    hides /sys, VISPEC_BENCH_AFFINITY=0): no pinning, and the returned string says why.  -> description for the bench line."""
    if os.environ.get("VISPEC_BENCH_AFFINITY", "1") == "0":
        return "off (VISPEC_BENCH_AFFINITY=0)"
    try:
        pr = torch.cuda.get_device_properties(local)
        addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return f"none (GPU {addr}: numa_node = -1, single-node host or no NUMA information)"
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"none (NUMA node {node} of GPU {addr} has no CPU this process may run on)"
        os.sched_setaffinity(0, cpus)
        return f"NUMA node {node} of GPU {addr}: {len(cpus)} CPUs"
    except Exception as e:
        return f"none ({type(e).__name__}: {e})"[:160]


def host_usage():
    """(process CPU seconds user + system, peak resident set size in GB) of this rank."""
    import resource
    t = os.times()
    return t.user + t.system, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def self_launch(n, script):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL rendezvous on 127.0.0.1) —
    the same environment `python -m torch.distributed.run --nproc-per-node N` would set.  Rank 0's stdout carries the JSON line.
    `script` = the bench script to start in every rank."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("VISPEC_FORCE_DEVICE"):
        log(f"error: --gpus {n} requested but {have} GPU(s) are visible; refusing to run a smaller job under that label")
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")  # what torch.distributed.run sets for nproc > 1: N ranks x lanes must not each spin up a 256-thread pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [pr.wait() for pr in procs]
    sys.exit(max(abs(rc) for rc in rcs))


_hip_rt = None


def _hip_runtime():
    """The HIP runtime torch itself uses (same file -> same loaded instance)."""
    global _hip_rt
    if _hip_rt is None:
        import ctypes
        import glob
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so*")) + ["libamdhip64.so"]
        for c in cands:
            try:
                _hip_rt = ctypes.CDLL(c)
                break
            except OSError:
                continue
        if _hip_rt is None:
            raise RuntimeError("libamdhip64.so not found")
    return _hip_rt


def xcd_mask_words(lane, lanes, n_cus=256, n_xcd=8, layout=None):
    """CU mask of lane `lane` of `lanes`: the CUs of its share of the XCDs (8 // lanes XCDs each; lanes that do not divide 8 share the last
    XCDs round-robin).  How the bits of a HIP CU mask map to XCDs is the driver's business: `layout` "striped" = bit i is CU (i // n_xcd) of
    XCD (i % n_xcd), "blocked" = bit i is CU (i % 32) of XCD (i // 32); $VISPEC_CU_MASK_LAYOUT overrides the default.  tools/cu_mask_probe.py
    prints the XCC ids the masked workgroups report, i.e. which of the two this driver uses."""
    layout = layout or os.environ.get("VISPEC_CU_MASK_LAYOUT", "blocked")
    per = max(1, n_xcd // lanes)
    mine = {(lane * per + j) % n_xcd for j in range(per)}
    per_xcd = n_cus // n_xcd
    words = [0] * ((n_cus + 31) // 32)
    for i in range(n_cus):
        xcd = i % n_xcd if layout == "striped" else i // per_xcd
        if xcd in mine:
            words[i // 32] |= 1 << (i % 32)
    return words, sorted(mine)


def masked_stream(device, lane, lanes):
    """A HIP stream whose kernels only run on lane `lane`'s XCDs (hipExtStreamCreateWithCUMask), as a torch stream object."""
    import ctypes
    rt = _hip_runtime()
    words, _ = xcd_mask_words(lane, lanes)
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(words)), arr)
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(h.value, device=device)
