#!/usr/bin/env python3
"""Headline benchmark: poses/sec of one "pose unit" = LC-loss forward+backward + one weighted-PnP solve,
B=256 poses x N=64 correspondences per GPU (BASELINE.json metric, configs[1]), fp32 I/O, synthetic inputs resident in HBM.

    python bench.py --gpus N --steps 200 --warmup 20

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a fresh `python -m torch.distributed.run`
child, before this process has touched the GPU) and relays the child's line and return code; launched by torchrun (WORLD_SIZE
set) it is one of the ranks, and a WORLD_SIZE that differs from --gpus is an error.

One process per GPU; the batch is sharded by pose (every pose is independent: SURVEY.md 8e), so there is no data-path
collective -- only the timing barrier.  Scaling is weak: each rank runs its own B=256 batch.
A step = ONE launch of the fused kernel (lc_pose_unit2_f32: the loss workgroups -- loss + d/d pts2d, d/d inv_std, d/d pts3d for
the `.mean()` cotangent -- and the LM-solve workgroups share a grid).  Two launch forms are timed with the same protocol and both
ride at the top level of the line: `value` (the K steps of a region are K kernel nodes of one hipGraph launch) and
`value_stream_order` (the same K launches issued one by one from Python).  Rank 0 prints ONE JSON line; it also carries the
dense configs' hot-path shapes (`dense`: glmo B=32 N=1024, zlmo B=32 N=1849), the keypoint head and the CPU baseline.
"""
from __future__ import annotations

import argparse
import ctypes
import datetime
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # before libgomp starts (cpu_baseline: an idle OpenMP team must not spin on the cores the other leg uses)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "poses/sec (cov-loss fwd+bwd + weighted PnP), B=256 N=64, 1/2/4/8 MI355X"  # BASELINE.json:metric
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: fp64 vector (non-MFMA) peak; SURVEY.md 8(d) prices the LM against it
SIMD_CYCLES_PER_S = 1024 * 2.4e9  # 256 CUs x 4 SIMDs x 2.4 GHz (MI355X_MICROARCH.md chip-level parameters)

# the dense configs' per-GPU hot-path shapes (BASELINE.json configs[2]-[4]): reference configs/glmo.yaml (B=32, 64x64 output,
# dense_sample 2 -> N = 32*32), zlmo (128x128 output, dense_sample 3 -> N = 43*43); losses.py:336-386, test.py:67-136
DENSE_WORKLOADS = {"glmo_dense": (32, 1024), "zlmo_dense": (32, 1849)}


def algorithmic_bytes(N: int, want_pts3d: bool):
    """SURVEY.md 8(d) per-pose figures."""
    loss = (36 + 28 + 96 + N * (12 + 8 + 8)) + (4 + N * (8 + 8)) + (N * 12 if want_pts3d else 0)
    pnp = (36 + 28 + N * (12 + 8 + 8)) + 36
    return loss, pnp


def algorithmic_flops(N: int, lm_evaluations: float):
    """loss fwd+bwd 115 kflop and 23 kflop per LM evaluation (Jacobian + normal equations + 6x6 solve) at N = 64;
    both are per-point work up to a few hundred flops of 6x6 algebra, so they are scaled by N / 64 for the dense shapes."""
    return 115e3 * N / 64.0, 23e3 * N / 64.0 * lm_evaluations


def static_counters(kernel: str, B: int, N: int):
    """Per-launch PMC counters of `kernel` on the default workload, from the newest committed rocprofv3 --pmc passes
    (profiles/<round>/pmc_traffic.json, written by scripts/profile_round.sh; PMC cannot be collected from inside a run).
    Returns (entry, source) -- source names the file, the round and the git SHA the passes were taken at -- or (None, None).
    The passes are taken on the bench's own shapes: the metric's (256, 64) and the dense blocks' (32, 1024) / (32, 1849), whose
    kernels carry the shape in their key (workgroup count of the tiled loss, points-per-thread template argument of the wide solve)."""
    if (B, N) not in ((256, 64), (32, 1024), (32, 1849)):
        return None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            key = kernel
            if not d.get(key) and kernel.endswith(">"):  # a template whose argument list has grown since: the one recorded instantiation that extends the key
                hits = [k for k in d if k.startswith(kernel[:-1] + ",")]
                key = hits[0] if len(hits) == 1 else kernel
            if d.get(key):
                meta = d.get("_meta", {})
                return d[key], {"file": os.path.relpath(f, ROOT), "git_sha": meta.get("git_sha", "unrecorded (round 1)"),
                                   "command": meta.get("command")}
        except Exception:
            pass
    return None, None


def dense_counter_keys(B: int, N: int, tiled: bool):
    """The keys under which profiles/<round>/pmc_traffic.json (scripts/summarize_prof.py) files the two kernels of a dense block: the
    tiled loss by its workgroup count (the slicing rule of lc_loss.hip: the smallest of 4 / 8 / 16 tiles per workgroup that leaves >= 3
    slices and <= 256 workgroups), the wide solve by its points-per-thread template argument.  None where no pass is keyed that way."""
    T = (N + 63) // 64
    groups = next((B * ((T + ts - 1) // ts) for ts in (4, 8, 16) if (T + ts - 1) // ts >= 3 and B * ((T + ts - 1) // ts) <= 256), None)
    loss_key = f"lc_cov_loss_tiled_kernel[{groups} workgroups]" if (tiled and groups) else None
    pnp_key = f"lc_pnp_lm_wide_kernel<false,false,{4 if N <= 1024 else 8}>" if 256 < N <= 2048 else None
    return loss_key, pnp_key


def rccl_version():
    try:
        return ".".join(map(str, torch.cuda.nccl.version()))
    except Exception as e:  # noqa: BLE001  (never let a version query take the measurement down)
        return f"unavailable ({type(e).__name__})"


def host_cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def valu_bound(sq: dict, B: int):
    """VALU-issue bound from SQ counters of one launch of B poses: SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD spends issuing
    VALU work; poses/s if all 1024 SIMDs issued such work every cycle."""
    cyc = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / B
    return cyc, SIMD_CYCLES_PER_S / cyc


def cpu_baseline(B, N, seed, budget_s=18.0):
    """The oracle (CPU restatement of the reference path) timed on this host: torch closed-form LC loss fwd+bwd, then the C/OpenMP LM
    solve, over B-sized batches.  ONE protocol for every thread count of the sweep (1, 4 = the reference's num_workers test.py:62,127,
    8, 16, 32, all cores): >= 10 warm-up calls, then a SUSTAINED loop of >= budget_s / (number of counts) seconds (>= 3 s by default);
    `value` is the best sustained rate, `poses_per_s_by_threads` holds every count's sustained rate -- the same loop, so the two cannot
    disagree (round 3 picked the count from best-of-three single calls and then ran the loop: 30 k vs 10 k poses/s in one line).  The
    two legs run one after the other and get the same budget: torch.set_num_threads(nt) for the loss, num_threads=nt for the solve
    (both are libgomp teams of this process; OMP_WAIT_POLICY=passive, set before torch is imported, keeps an idle team from spinning
    on the cores the other leg wants)."""
    from lc_amd import synth
    from oracle import lc_loss_oracle, pnp_oracle

    avail = os.cpu_count() or 1
    b = synth.make_batch(B, N, seed=seed)
    L = torch.diag_embed(b["inv_std"]).numpy()
    npb = {k: v.numpy() for k, v in b.items()}
    go = torch.full((B,), 1.0 / B)

    def one(nt):
        lc_loss_oracle.loss_and_grads(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"], grad_out=go)
        pnp_oracle.solve_batched(npb["start"], npb["K"], npb["pts2d"], npb["pts3d"], L, num_threads=nt)

    counts = sorted({min(avail, c) for c in (1, 4, 8, 16, 32, avail)})
    per_count_s = max(budget_s / len(counts), 0.05)
    by_threads, detail = {}, {}
    for nt in counts:
        torch.set_num_threads(nt)
        for _ in range(10):  # thread teams, page-in, allocator (the reference's first call costs seconds: SURVEY.md 8d)
            one(nt)
        calls = []
        t_start = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            one(nt)
            t1 = time.perf_counter()
            calls.append(t1 - t0)
            if t1 - t_start > per_count_s or len(calls) >= 5000:
                break
        dt = time.perf_counter() - t_start
        by_threads[str(nt)] = B * len(calls) / dt
        q = sorted(calls)
        detail[str(nt)] = {"calls": len(calls), "seconds": dt,
                           "ms_per_call_p10_p50_p90": [q[int(0.1 * (len(q) - 1))] * 1e3, q[len(q) // 2] * 1e3, q[int(0.9 * (len(q) - 1))] * 1e3]}
    cores = int(max(by_threads, key=by_threads.get))
    d = detail[str(cores)]
    return dict(value=by_threads[str(cores)], unit="poses/s", cores=cores, kind="port", host_cores_available=avail, host_cpu=host_cpu_model(),
                poses_per_s_by_threads=by_threads, sustained=detail, omp_wait_policy=os.environ.get("OMP_WAIT_POLICY"),
                reference_anchor="un-restated reference, survey container (8-core Xeon, BASELINE.md section 2): lib.cov_mixed.Loss_cov_mixed "
                                 "fwd+bwd alone 36.9 ms per 256 poses = 6.9 k poses/s; the Ceres solve cannot be built or timed anywhere here",
                sample=f"{d['calls']} batches of B={B} N={N} in {d['seconds']:.1f} s sustained after 10 warm-up calls (oracle: torch-CPU closed-form loss "
                       f"fwd+bwd + C/OpenMP LM, {cores} threads = the best SUSTAINED rate of {counts}; every count ran the same loop)")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--regions", type=int, default=51, help="timed regions of exactly --steps steps each (barrier + synchronize on both "
                    "sides of every region); the MEDIAN region is reported (one 3 ms region moves +-8 %% with launch jitter)")
    ap.add_argument("--steady-batch", type=int, default=65536, help="B of the in-run steady-state measurement (0: skip)")
    ap.add_argument("--batch", type=int, default=256, help="poses per GPU per step")
    ap.add_argument("--npts", type=int, default=64)
    ap.add_argument("--slots", type=int, default=4, help="--launch streams: independent batches in flight (own buffers, own stream)")
    ap.add_argument("--launch", default="graph_region", choices=["fused", "eager", "eager2", "graph", "graph2", "graph_fused", "graph_region", "streams"],
                    help="graph_region (default): a step is ONE fused launch (lc_pose_unit2_f32: loss and PnP workgroups share a grid) and "
                         "the K steps of a timed region are K kernel nodes of one hipGraph launch -- `value`; the stream-order form of the "
                         "same launches is timed too -- `value_stream_order`; fused: only the stream-order form; eager: two launches per "
                         "step on one stream; eager2: LM solve forked onto a second stream; graph / graph2 / graph_fused: one step replayed "
                         "as a hipGraph")
    ap.add_argument("--workload", default="all", choices=["all", "metric", *DENSE_WORKLOADS],
                    help="which extra blocks ride on the line next to the headline (which is always the metric's B=256 N=64): the dense "
                         "configs' hot-path shapes `dense.glmo_dense` (B=32 N=1024) and `dense.zlmo_dense` (B=32 N=1849); metric: none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-head", action="store_true", help="skip the keypoint-head measurement attached as out['head']")
    ap.add_argument("--cpu-budget", type=float, default=18.0, help="seconds of sustained CPU-baseline loops, split evenly over the thread counts of the sweep")
    return ap.parse_args(argv)


def launch_plan(gpus: int, env_world, visible_gpus: int, share_gpu: bool):
    """What `--gpus N` means for this process (pure; tests/test_host_logic.py):
    ("self_launch", N)  start N ranks as a child `torch.distributed.run` (N > 1, no WORLD_SIZE yet);
    ("rank", world)     run as one rank of `world`;
    ("error", message)  refuse."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    if env_world is None:
        if gpus == 1:
            return "rank", 1
        if not share_gpu and visible_gpus < gpus:
            return "error", f"--gpus {gpus} but only {visible_gpus} GPU(s) are visible (one process per GPU; LC_BENCH_SHARE_GPU=1 is the one-GPU test mode)"
        return "self_launch", gpus
    world = int(env_world)
    if world != gpus:
        return "error", f"--gpus {gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {gpus}, or drop the launcher and let bench.py start the ranks"
    return "rank", world


def self_launch(n: int, argv):
    """N ranks of this script as a fresh child process (never an exec: this process may not touch the GPU before, and does not)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode  # stdout / stderr are inherited: rank 0's JSON line is this process's line


def init_collectives(world: int, dev, share_gpu: bool):
    """The timing collectives of an N-rank run.  gloo first (cheap, always there: it carries the agreement flags and the
    per-rank region times); then an RCCL group for the barrier, adopted only if EVERY rank brought it up and passed a probe
    all-reduce (agreed by a MIN over gloo) -- otherwise every rank uses gloo, together."""
    import torch.distributed as dist

    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
    info = {"dist": dist, "barrier_group": None, "backend": "gloo", "rccl_error": None}
    if share_gpu:
        return info
    ok, err, grp = 1, None, None
    try:  # RCCL over xGMI; one tiny all-reduce proves the communicator works before anything is timed
        grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe, group=grp)
        torch.cuda.synchronize(dev)
        if int(probe.item()) != world:
            raise RuntimeError(f"probe all-reduce returned {probe.item()} for world {world}")
    except Exception as e:  # noqa: BLE001 -- the data path has no collective: the backend only carries the timing barrier
        ok, err = 0, f"{type(e).__name__}: {e}"
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        info.update(barrier_group=grp, backend="nccl")
    else:
        info["rccl_error"] = err or "another rank could not bring RCCL up"
        print(f"bench.py rank {dist.get_rank()}: RCCL unavailable on at least one rank ({info['rccl_error']}); every rank uses gloo "
              f"for the timing barrier", file=sys.stderr)
    return info


def main():
    args = parse_args()
    share_gpu = os.environ.get("LC_BENCH_SHARE_GPU", "0") == "1"
    # torch.cuda.device_count() does not initialise the GPU on this image; nothing else may touch it before a self-launch
    kind, val = launch_plan(args.gpus, os.environ.get("WORLD_SIZE"), torch.cuda.device_count(), share_gpu)
    if kind == "error":
        print(f"bench.py: {val}", file=sys.stderr)
        raise SystemExit(2)
    if kind == "self_launch":
        raise SystemExit(self_launch(val, sys.argv[1:]))
    world = val
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    # LC_BENCH_SHARE_GPU=1 (test only): every rank uses GPU 0 and the timing collectives run over gloo, so that the N > 1
    # control flow (barrier, max-over-ranks, aggregation) can be exercised on a one-GPU box; RCCL refuses two ranks per device.
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # LC_BENCH_FORCE_COLLECTIVES=1 (test only): bring the timing collectives up even for one rank, so that the RCCL code path of an
    # N-GPU run (group creation bound to the device, probe all-reduce, barrier on the RCCL group) is exercised on a one-GPU box
    force_coll = os.environ.get("LC_BENCH_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ
    coll = init_collectives(world, dev, share_gpu) if (world > 1 or force_coll) else None
    dist = coll["dist"] if coll else None
    if os.environ.get("LC_BENCH_FAIL_RANK") == str(rank):  # test only: a rank that dies before the first barrier (the launcher must take the others down)
        print(f"bench.py rank {rank}: LC_BENCH_FAIL_RANK set, exiting with code 3 before the first barrier", file=sys.stderr)
        sys.exit(3)

    def barrier():
        if coll is None:
            return
        if coll["barrier_group"] is not None:
            dist.barrier(group=coll["barrier_group"], device_ids=[dev_index])
        else:
            dist.barrier()

    def all_agree(flag: bool) -> bool:
        """True iff `flag` holds on every rank (MIN over gloo)."""
        if coll is None:
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    from lc_amd import _lib, synth

    lib = _lib.load()
    B, N = args.batch, args.npts
    P = _lib.ptr

    class Unit:
        """Device buffers of one batch of Bq poses x Nq correspondences and the launches over them."""

        def __init__(self, Bq, Nq, seed, rows=None):
            """rows = (lo, hi): this unit holds that slice of the seeded batch of Bq poses (a rank's shard of a GLOBAL batch)."""
            full = synth.make_batch(Bq, Nq, seed=seed)
            if rows is not None:
                full = {k: v[rows[0]:rows[1]].contiguous() for k, v in full.items()}
                Bq = rows[1] - rows[0]
            self.B, self.N = Bq, Nq
            self.b = b = {k: v.to(dev) for k, v in full.items()}
            self.go = torch.full((Bq,), 1.0 / Bq, device=dev)
            self.sqrt_diag = b["inv_std"].contiguous()  # icov = inv_std^2 -> sqrt factor = inv_std (cer_solver.py:37-38)
            self.loss = torch.empty(Bq, device=dev)
            self.d_u, self.d_s, self.d_x = torch.empty_like(b["pts2d"]), torch.empty_like(b["inv_std"]), torch.empty_like(b["pts3d"])
            self.states = torch.empty_like(b["start"])
            self.tr = torch.empty(Bq, device=dev)
            self.ret = torch.empty(Bq, device=dev, dtype=torch.int32)
            self.iters = torch.zeros(Bq, device=dev, dtype=torch.int32)
            # dense shapes: the workspace through which the tiles of a sample meet (zeroed once, left zeroed by every launch)
            nws = int(lib.lc_cov_loss_workspace_bytes(Bq, Nq))
            self.ws = torch.zeros(nws, dtype=torch.uint8, device=dev) if nws else None

        def launch_loss(self, stream=None):
            b = self.b
            rc = lib.lc_cov_loss3_fwd_bwd_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None,
                                              P(b["bbox_3d"]), P(self.go), self.B, self.N, 32.0, 3.0, 4.0, 0, P(self.loss), P(self.d_u),
                                              P(self.d_s), P(self.d_x), None, P(self.ws), 0 if self.ws is None else self.ws.numel(),
                                              stream or _lib.stream_ptr(dev))
            assert rc == 0

        def launch_pnp(self, stream=None, iters=False):
            # start poses are read-only input, states is output: every step solves from the same perturbed pose
            b = self.b
            rc = lib.lc_pnp_lm3_f32(P(b["K"]), P(b["pts3d"]), P(b["pts2d"]), None, P(self.sqrt_diag), None, None, P(b["start"]), P(self.states),
                                    P(self.tr), P(self.ret), P(self.iters) if iters else None, self.B, self.N, 50, 1e-6, 0, 0, None, 0,
                                    stream or _lib.stream_ptr(dev))
            assert rc == 0

        def launch_fused(self, stream=None):
            # one grid for both halves of the pose unit (lc_amd/csrc/lc_fused.hip; dense shapes: lc_fused_dense.hip)
            b = self.b
            rc = lib.lc_pose_unit2_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]),
                                       P(self.go), self.B, self.N, 32.0, 3.0, 4.0, P(self.loss), P(self.d_u), P(self.d_s), P(self.d_x),
                                       P(self.sqrt_diag), P(b["start"]), P(self.states), P(self.tr), P(self.ret), None, 50, 1e-6,
                                       P(self.ws), 0 if self.ws is None else self.ws.numel(), stream or _lib.stream_ptr(dev))
            assert rc == 0

        def check(self):
            assert int(self.ret.sum().item()) == 0 and bool(torch.isfinite(self.loss).all())

        def mean_lm_iterations(self):
            """Trust-region iterations per pose of this batch (the stand-alone solve is bit-identical to the fused grid's half:
            tests/test_gpu_fused.py), for SURVEY 8(d)'s flop figure."""
            self.launch_pnp(iters=True)
            torch.cuda.synchronize(dev)
            return float(self.iters.float().mean().item()), int(self.iters.max().item())

    main_unit = Unit(B, N, seed=rank)
    side = torch.cuda.Stream(dev)

    def step_serial():
        main_unit.launch_loss()
        main_unit.launch_pnp()

    def step_forked():
        # the two kernels are independent (the loss linearises at the GT pose, the solve starts from `start`):
        # fork the LM solve onto a second HIP stream, join at the end of the step
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            main_unit.launch_pnp()
        main_unit.launch_loss()
        main.wait_stream(side)

    step_fused = main_unit.launch_fused

    # --launch streams (NOT the default, reported separately in profiles/r03/NOTES.md): `slots` independent B-sized batches in flight, each
    # with its own inputs, outputs and stream -- the serving-side picture (independent requests), where a 512-workgroup step
    # leaves most of the chip idle.  A step is still one fused launch over one batch; steps of different slots overlap.
    slot_state = []
    if args.launch == "streams":
        for s_i in range(args.slots):
            slot_state.append((Unit(B, N, seed=1000 * (rank + 1) + s_i), torch.cuda.Stream(dev)))
        torch.cuda.synchronize(dev)
    slot_i = [0]

    def step_streams():
        u, stream = slot_state[slot_i[0] % len(slot_state)]
        slot_i[0] += 1
        u.launch_fused(ctypes.c_void_p(stream.cuda_stream))

    if args.launch in ("fused", "graph_region") and N > 64:
        args.launch = "eager"
    step_eager = {"eager": step_serial, "graph": step_serial, "fused": step_fused, "graph_fused": step_fused, "graph_region": step_fused,
                  "streams": step_streams}.get(args.launch, step_forked)

    def capture(fn, n):
        """`n` calls of fn as one hipGraph, or None (with the reason on stderr) if the runtime cannot capture them."""
        warm = torch.cuda.Stream(dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):
            fn()
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        from lc_amd.inference import quiet_capture  # no Python GC while the stream is capturing

        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: calls of other threads of the process (e.g. the RCCL watchdog of a multi-rank run) neither fail nor invalidate the capture
            with quiet_capture(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(n):
                    fn()
            g.replay()
            torch.cuda.synchronize(dev)
            return g
        except Exception as e:  # noqa: BLE001
            print(f"bench.py rank {rank}: hipGraph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.synchronize(dev)
            return None

    graph = None
    if args.launch.startswith("graph"):
        graph = capture(step_eager, args.steps if args.launch == "graph_region" else 1)
        # every rank times the SAME launch form: if one rank could not capture, all of them issue the launches one by one
        if not all_agree(graph is not None):
            if args.launch != "graph_region":
                raise SystemExit(f"bench.py rank {rank}: --launch {args.launch} needs a hipGraph capture on every rank")
            graph, args.launch = None, "fused"

    def fence():
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)

    from lc_amd import dist as lcd

    def timed_regions(step, steps_per_call):
        """The protocol: W warm-up steps, then R regions of exactly K steps, each bracketed by barrier + synchronize; a region
        lasts as long as its slowest rank (MAX over ranks), the MEDIAN region is reported."""
        for _ in range(max(1, args.warmup // steps_per_call)):
            step()
        region_s = []
        for _ in range(max(1, args.regions)):
            fence()  # synchronize + barrier + synchronize: every rank starts the region together
            t0 = time.perf_counter()
            for _ in range(args.steps // steps_per_call):
                step()
            torch.cuda.synchronize(dev)  # this rank's K steps are done ...
            region_s.append(time.perf_counter() - t0)
            barrier()  # ... and the region ends when the slowest rank is (MAX over ranks below); the collective's own
            # latency (tens of us over RCCL) is not part of the K steps
        return lcd.aggregate_regions(region_s, args.steps, device="cpu")  # per-rank region times travel over gloo

    if graph is not None:
        agg = timed_regions(graph.replay, args.steps if args.launch == "graph_region" else 1)
    else:
        agg = timed_regions(step_eager, 1)
    # the stream-order form of the same K launches, same protocol (what a caller pays who issues the launches one by one)
    agg_so = timed_regions(step_fused, 1) if args.launch == "graph_region" else (agg if args.launch == "fused" else None)
    elapsed = agg["median_region_s"]
    # SURVEY.md 8(e) writes the partition as "32/GPU at B=256, 8 GPUs": beside the weak headline (B poses per rank) the STRONG split of ONE
    # global batch of B poses over the ranks (lc_amd.dist.shard_range), same launch form, same region protocol.  A 13 us launch is bound by
    # per-wave latency, so the strong line is expected to stay flat -- a fact to print beside the weak curve, not to omit.
    strong = None
    if world > 1 and args.launch in ("fused", "graph_region") and B >= world:
        lo, hi = lcd.shard_range(B, rank, world)
        s_unit = Unit(B, N, seed=12345, rows=(lo, hi))  # every rank builds the same global batch and keeps its shard
        s_graph = capture(s_unit.launch_fused, args.steps) if args.launch == "graph_region" else None
        use_graph = all_agree(s_graph is not None)
        s_agg = timed_regions(s_graph.replay, args.steps) if use_graph else timed_regions(s_unit.launch_fused, 1)
        s_unit.check()
        strong = {"scaling": "strong", "value": B * args.steps / s_agg["median_region_s"], "unit": "poses/s", "global_batch": B,
                  "per_rank_batch": hi - lo, "ms_per_step": s_agg["ms_per_step"], "ranks_seen": s_agg["ranks_seen"],
                  "per_rank_ms_per_step": s_agg["per_rank_ms_per_step"], "launch": "graph_region" if use_graph else "fused",
                  "note": f"one global batch of {B} poses split contiguously over {world} rank(s) (lc_amd.dist.shard_range: {hi - lo} on this rank), no "
                          "data-path collective; same regions / barrier / MAX-over-ranks / median protocol as the weak headline.  The launch is "
                          "latency-bound (one wave per pose, the slowest pose's LM iterations): fewer poses per rank do not shorten it"}
    if args.launch == "streams":
        for u, _ in slot_state:
            u.check()
    else:
        main_unit.check()

    # per-kernel launch duration with events on the launch stream (torch's current stream == the kernels' stream)
    def kernel_ms(fn, reps=200, windows=5):
        """Average launch duration: HIP events around `reps` back-to-back launches on the launch stream, MEDIAN of `windows` such
        windows (the shared pool shows occasional ~60 ms stalls; one of them inside a 3 ms window would wreck a single reading)."""
        fn()
        torch.cuda.synchronize(dev)
        out = []
        for _ in range(windows):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            out.append(e0.elapsed_time(e1) / reps)
        return sorted(out)[len(out) // 2]

    def kernel_percentiles_us(fn, reps=200):
        """p10 / median / p90 of individually event-timed launches (SURVEY.md 8d timing protocol)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        fn()
        torch.cuda.synchronize(dev)
        for e0, e1 in evs:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize(dev)
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
        return [ts[reps // 10], ts[reps // 2], ts[(9 * reps) // 10]]

    def steady_state(Bs):
        """One large batch: every SIMD holds several pose waves, the per-wave latency chain is hidden."""
        u = Unit(Bs, N, seed=977)
        ms = kernel_ms(u.launch_fused, reps=6, windows=5)
        assert int(u.ret.sum().item()) == 0
        return Bs, Bs / (ms * 1e-3), ms

    def dense_block(name, Bd, Nd):
        """One dense config's per-GPU hot-path shape: LC-loss fwd+bwd (tiled form) + one weighted-PnP solve per sample over the same N
        correspondences (four wavefronts per pose).  A step = one pose-unit launch over the batch (lc_pose_unit2_f32: the two kernels'
        workgroups in one grid), with the two stand-alone launches back to back timed beside it (`two_launches`)."""
        u = Unit(Bd, Nd, seed=4242)

        def step():
            u.launch_loss()
            u.launch_pnp()
        t_loss, t_pnp = kernel_ms(u.launch_loss, reps=50), kernel_ms(u.launch_pnp, reps=50)

        def wall_clock(step_fn):
            """the headline's protocol in small: 11 regions of 20 steps (one graph replay where it captures), synchronize around
            each, median; also the event-timed step inside the graph and in stream order"""
            g = capture(step_fn, 20)
            t_graph = kernel_ms(g.replay, reps=5) / 20 if g is not None else None
            t_so = kernel_ms(step_fn, reps=50)
            regs = []
            for _ in range(11):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                if g is not None:
                    g.replay()
                else:
                    for _ in range(20):
                        step_fn()
                torch.cuda.synchronize(dev)
                regs.append((time.perf_counter() - t0) / 20)
            return sorted(regs)[len(regs) // 2], t_graph, t_so, g is not None
        wall, t_step, t_step_so, graphed = wall_clock(step)
        u.check()
        # the same unit as ONE launch (tiled loss workgroups + four-wave solve workgroups in one grid: lc_pose_unit2_f32)
        fused = None
        if u.ws is not None and Nd <= 2048:
            u.loss.fill_(float("nan"))
            u.states.fill_(float("nan"))
            wall_f, t_f, t_f_so, graphed_f = wall_clock(u.launch_fused)
            u.check()
            fused = {"value": Bd / wall_f, "unit": "poses/s", "ms_per_step": wall_f * 1e3,
                     "launch": "graph_region (20 steps per replay)" if graphed_f else "stream order",
                     "step_us_events": {"graph": None if t_f is None else t_f * 1e3, "stream_order": t_f_so * 1e3},
                     "what": "lc_pose_unit_dense_kernel: the loss's and the solve's workgroups in one grid, results bit for bit the two launches'"}
        it_mean, it_max = u.mean_lm_iterations()
        by_l, by_p = algorithmic_bytes(Nd, True)
        fl_l, fl_p = algorithmic_flops(Nd, it_mean + 1.0)

        def roof(t_ms, by, fl, counter_key=None):
            gbs, tf = by * Bd / (t_ms * 1e-3) / 1e9, fl * Bd / (t_ms * 1e-3) / 1e12
            out = {"kernel_us": t_ms * 1e3, "algorithmic_bytes_per_sample": by, "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                   "algorithmic_kflop_per_sample": fl / 1e3, "tflops": tf, "fp64_vector_frac": tf / FP64_VECTOR_PEAK_TFLOPS}
            ctr, src = static_counters(counter_key, Bd, Nd) if counter_key else (None, None)
            if ctr is not None:  # committed counter pass of this very shape: HBM bytes per launch and the issue / wait shares
                out["traffic"] = ctr.get("bytes_per_launch")
                out["algorithmic_bytes_per_launch"] = by * Bd
                sq = ctr.get("sq", {})
                if sq.get("SQ_WAVE_CYCLES"):
                    out["valu_active_share_of_wave_cycles"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
                    out["waiting_share_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
                out["counters_from"] = src
            return out
        loss_key, pnp_key = dense_counter_keys(Bd, Nd, u.ws is not None)
        two = {"value": Bd / wall, "unit": "poses/s", "ms_per_step": wall * 1e3,
               "launch": "graph_region (20 steps per replay)" if graphed else "stream order",
               "step_us_events": {"graph": None if t_step is None else t_step * 1e3, "stream_order": t_step_so * 1e3},
               "what": "the loss launch and the solve launch back to back"}
        top = fused if fused is not None else two  # like the headline: a step is ONE launch over the batch where that form exists
        return {"workload": f"{name}: B={Bd} samples x N={Nd} correspondences per GPU (fp32 I/O, fp64 inside)",
                "value": top["value"], "unit": "poses/s", "ms_per_step": top["ms_per_step"], "launch": top["launch"],
                "step_us_events": top["step_us_events"],
                "step": "one launch (lc_pose_unit2_f32)" if fused is not None else "two launches",
                "two_launches": two,
                "lm_iterations": {"mean": it_mean, "max": it_max},
                "lc_cov_loss_kernel": dict(roof(t_loss, by_l, fl_l, loss_key), form="tiled: the sample's 64-point tiles dealt to 256-thread workgroups (4, 8 or 16 tiles each), one hand-off" if u.ws is not None else "one workgroup per sample"),
                "lc_pnp_lm_wide_kernel": roof(t_pnp, by_p, fl_p, pnp_key),
                "bound": "neither HBM nor MFMA: VALU issue / per-workgroup latency (SURVEY.md 8d); both fractions are quoted"}

    def test_time_block(name, objects=64):
        """SURVEY.md 8f rows f1 + f2 + f3 + a24 chained -- the reference's test.py:67-136 for one batch of detections AT THE REFERENCE'S OWN KNOBS
        (lc_amd.synth.TEST_TIME_CONFIGS; gsplmo = the sparse head's path, test.py:47-64 at configs/gsplmo.yaml's 16 keypoints): zlmo = configs/zlmo.yaml:30-37 (128x128 maps, dense_sample 1 -> 16 384 candidates per object,
        quantile_in_mask 0.2, rel_reproj_err, solvers [weighted_filtered], 21 code planes + model_transform), glmo = configs/glmo.yaml:28-32
        (64x64 maps, stride 2, quantile 0.3, solvers [weighted]).  [code decode,] dense front end + point selection (one launch), P3P RANSAC
        over ALL selected points (three launches, the inlier re-selection inside the last), inlier refinement + weighted solve; no host
        synchronisation, replayed as ONE hipGraph.  Timed like the headline in small: 11 regions of 20 replays, synchronize around each, median
        (`us_per_call_replayed_200`: regions of 200 replays, where the one synchronisation per region no longer shows); the eager call (launches
        issued from Python) beside it."""
        from lc_amd import splitws, synth
        from lc_amd.config import AttrDict
        from lc_amd.inference import GraphedSolvePnP, solve_pnp
        from lc_amd.transforms import quaternion_rep_to_RT

        if name == "gsplmo":  # the sparse head's test-time path (test.py:47-64; configs/gsplmo.yaml: sparse_cnt 16, solvers [ransac, weighted], reprojection error 2 px)
            b = synth.make_batch(objects, 16, seed=3, noise_px=0.3, outlier_frac=0.0)  # (the weighted solve takes every keypoint: a network marks its bad ones by a large predicted deviation)
            gt = dict(out_K=b["K"], pts3d=b["pts3d"], pose_best=b["pose"])
            net = dict(pts2d=b["pts2d"], pts2d_std=1 / b["inv_std"])
            cfg = dict(rel_reproj_err=False, solvers=["ransac", "weighted"])
            key, what = "weighted", "16 keypoints with predicted standard deviations (sparse head), reprojection error 2 px, solvers ransac + weighted"
        elif name == "hybrid_r03":  # rounds 2-3's block (neither config): kept as history only
            gt, net = synth.dense_inputs(B=objects, H=64, W=64, seed=3)
            net["xyz_weight_logits"] = net["xyz_weight_logits"] + 3 * gt["msk_vis"][:, None]
            net["msk_vis_logits"] = (gt["msk_vis"][:, None] * 2 - 1) * 4
            cfg = dict(dense_point_select="quantile_in_mask", quantile=0.5, dense_sample=2, solvers=["weighted", "weighted_filtered"])
            key, what = "weighted", "64x64 maps, stride 2 (1024 candidates), quantile_in_mask 0.5, solvers weighted + weighted_filtered (rounds 2-3's hybrid)"
        else:
            half = name.endswith("_bf16")  # the same chain on the maps a bf16-autocast backbone hands over (BASELINE.json configs[2]): read natively, no cast
            name = name[:-5] if half else name
            cfg, gt, net = synth.test_time_inputs(name, B=objects, seed=3)
            if half:
                net = {k: (v.to(torch.bfloat16) if v.is_floating_point() and k != "xyz_weights_scale" else v) for k, v in net.items()}
            spec = synth.TEST_TIME_CONFIGS[name]
            stride = cfg.get("dense_sample", 2)
            key = "weighted-filtered" if "weighted_filtered" in cfg["solvers"] else "weighted"
            what = (f"{spec['H']}x{spec['W']} maps, stride {stride} ({-(-spec['H'] // stride) * -(-spec['W'] // stride)} candidates), {cfg['dense_point_select']} "
                    f"{cfg['quantile']}, solvers {cfg['solvers']}" + (f", {sum(spec['bits'])} code planes + model_transform, rel_reproj_err" if spec["bits"] else "")
                    + (", bf16 maps" if half else ""))
        cfg = AttrDict(cfg)
        gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
        net = {k: v.to(dev).contiguous() for k, v in net.items()}  # a convolution's output is contiguous NCHW
        eager = solve_pnp(cfg, net, gt)
        solver = GraphedSolvePnP(cfg, net, gt)
        solver.graph.replay()
        torch.cuda.synchronize(dev)
        same = all(torch.equal(solver._res[k], eager[k]) for k in eager)

        def regions(fn, reps=20, n=11):
            out = []
            for _ in range(n):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(dev)
                out.append((time.perf_counter() - t0) / reps)
            return sorted(out)[n // 2]
        t_replay = regions(solver.graph.replay)
        t_long = regions(solver.graph.replay, reps=200, n=9)  # the same replays in regions long enough that the one synchronisation per region no longer shows
        t_eager = regions(lambda: solve_pnp(cfg, net, gt))
        Rg, tg = quaternion_rep_to_RT(gt["pose_best"].double())  # pose error of the config's solver against the synthetic ground truth
        Re, te = quaternion_rep_to_RT(eager[key].double())
        return {"workload": f"{objects} objects x {what}, 150 hypotheses",
                "us_per_call_replayed": t_replay * 1e6, "us_per_call_replayed_200": t_long * 1e6, "us_per_call_eager": t_eager * 1e6,
                "objects_per_s_replayed": objects / t_long, "replay_equals_eager": bool(same), "solver": key,
                "split_forms": not splitws.is_off(),  # several workgroups per object where the shapes take them (lc_amd/splitws.py; LC_AMD_PNP_SPLIT=0 turns them off)
                "median_translation_error_mm": float((te - tg).norm(dim=-1).median()), "max_translation_error_mm": float((te - tg).norm(dim=-1).max()),
                "max_rotation_error": float((Re - Rg).abs().max())}

    if rank == 0:
        t_loss = kernel_ms(main_unit.launch_loss)
        t_pnp = kernel_ms(main_unit.launch_pnp)
        by_loss, by_pnp = algorithmic_bytes(N, True)
        dom = ("lc_cov_loss_kernel", t_loss, by_loss) if t_loss >= t_pnp else ("lc_pnp_lm_kernel", t_pnp, by_pnp)
        kernel_us = {"lc_cov_loss_kernel": t_loss * 1e3, "lc_pnp_lm_kernel": t_pnp * 1e3}
        fused_forms = args.launch in ("fused", "graph_fused", "graph_region")
        if fused_forms:
            t_unit = kernel_ms(step_fused)  # launches issued one by one (stream order)
            kernel_us["lc_pose_unit_kernel"] = t_unit * 1e3
            if args.launch == "graph_region":  # the pattern of the timed region: the kernel as a node of the K-step graph (events around replays)
                kernel_us["lc_pose_unit_kernel_stream_order"] = t_unit * 1e3
                t_unit = kernel_ms(graph.replay, reps=max(1, 200 // args.steps)) / args.steps
                kernel_us["lc_pose_unit_kernel"] = t_unit * 1e3
            dom = ("lc_pose_unit_kernel", t_unit, by_loss + by_pnp)
            kernel_us["lc_pose_unit_kernel_p10_p50_p90"] = kernel_percentiles_us(step_fused)
        # every figure of the roofline block is formed from the PROTOCOL's step (ms_per_step of this rank's line), not from the
        # event-timed replay; the event-timed launch durations ride along in kernel_us
        step_s = elapsed / args.steps if fused_forms else dom[1] * 1e-3
        hbm_gbs = dom[2] * B / step_s / 1e9
        poses_per_s = B / step_s
        it_mean, it_max = main_unit.mean_lm_iterations()
        fl_loss, fl_pnp = algorithmic_flops(N, it_mean + 1.0)
        tflops = (fl_loss + fl_pnp) * B / step_s / 1e12
        fl_nom = 350e3 * N / 64.0  # SURVEY.md 8(d)'s nominal pose unit: 115 + 23 x ~10 kflop = ~350 kflop, whatever the batch executes
        flops = {"bound": "fp64_vector", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                 "nominal_frac": fl_nom * B / step_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                 "nominal_note": "`frac` counts the LM iterations this batch executes; `nominal_frac` credits SURVEY.md 8(d)'s nominal ~350 kflop "
                                 "pose unit (~10 iterations) on the same step time -- the two figures people quote, both from this line",
                 "algorithmic_kflop_per_pose": {"loss": fl_loss / 1e3, "pnp": fl_pnp / 1e3},
                 "lm_iterations": {"mean": it_mean, "max": it_max},
                 "note": "SURVEY.md 8(d): 115 kflop (loss fwd+bwd) + 23 kflop x (LM iterations + 1 initial evaluation) per pose, the "
                         "iteration count read from the solve's `iters` output on this batch; peak = fp64 vector (the LM runs in fp64)"}
        ctr, src = static_counters(dom[0], B, N)
        hbm = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_pose": {"loss": by_loss, "pnp": by_pnp},
               "note": "secondary: 5 KB working set per pose, one wave per pose -- HBM never binds this kernel"}
        # Top level = SURVEY.md 8(d)'s figure: algorithmic flops (measured LM iterations) against the fp64 vector peak.  The kernel's own
        # VALU-issue ratio rides along as `valu_issue` -- its peak is derived from the kernel's instruction count, so it can show latency
        # hiding but never a wasted instruction (VERDICT r4 #4a).
        roof = dict(flops, kernel=dom[0], traffic=ctr.get("bytes_per_launch") if ctr is not None else None, kernel_us=kernel_us, hbm=hbm)
        if ctr is not None:
            roof["counters_from"] = src
        if ctr is not None and ctr.get("sq", {}).get("SQ_ACTIVE_INST_VALU"):
            cyc, peak = valu_bound(ctr["sq"], B)
            roof["valu_issue"] = {"bound": "valu_issue", "achieved": poses_per_s, "peak": peak, "unit": "poses/s", "frac": poses_per_s / peak,
                                  "valu_insts_per_pose": ctr["sq"]["SQ_INSTS_VALU"] / B, "simd_cycles_per_pose": cyc,
                                  "note": "how much of the kernel's OWN VALU issue time the launch hides: peak = 1024 SIMDs x 2.4 GHz / VALU-issue "
                                          "SIMD-cycles per pose (SQ_ACTIVE_INST_VALU of the committed counter pass).  Self-derived -- a latency-hiding "
                                          "ratio, not a roofline: it cannot show a wasted instruction"}
        out = {
            "metric": METRIC,
            "value": B * world * args.steps / elapsed,
            "unit": "poses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "dtype_note": "fp32 I/O (the reference's precision); Jacobians, 6x6 algebra and LM state in fp64 like Ceres",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic B={B} N={N} 2D-3D correspondences per GPU, HIP weighted-PnP + cov-loss",
                       "global_batch": B * world, "n_points": N, "sharding": f"poses over {world} rank(s), no data-path collective",
                       "launch": args.launch, "launch_agreed_by_all_ranks": True,
                       **({"slots_in_flight": args.slots} if args.launch == "streams" else {})},
            "timing": {"protocol": f"{agg['regions']} regions of exactly {args.steps} steps, barrier + synchronize around each, MAX over ranks "
                                   f"per region, MEDIAN region reported"
                                   + ("; the K steps of a region are K kernel nodes of one hipGraph launch" if args.launch == "graph_region" else ""),
                       "region_ms_per_step": agg["region_ms_per_step"]},
            "ranks_seen": agg["ranks_seen"], "collective_backend": coll["backend"] if coll else "none",
            "region_times_gathered_over": agg["backend"],
            "rccl_version": rccl_version() if (coll and coll["backend"] == "nccl") else None,
            "per_rank_ms_per_step": agg["per_rank_ms_per_step"],
            "roofline": roof,
            "library": {"embedded_source_hash": lib.lc_amd_source_hash().decode(), "source_hash_on_disk": __import__("lc_amd.build", fromlist=["x"]).source_hash()},
        }
        if coll and coll["rccl_error"]:
            out["rccl_error"] = coll["rccl_error"]
        if strong is not None:
            out["strong"] = strong
        if agg_so is not None:
            out["value_stream_order"] = B * world * args.steps / agg_so["median_region_s"]
            out["ms_per_step_stream_order"] = agg_so["ms_per_step"]
            out["timing"]["stream_order"] = {"what": "the same K fused launches per region issued one by one from Python (per-launch host "
                                                     "cost included), same regions / barrier / MAX-over-ranks / median protocol",
                                             "region_ms_per_step": agg_so["region_ms_per_step"],
                                             "per_rank_ms_per_step": agg_so["per_rank_ms_per_step"]}
        if args.steady_batch > 0 and world == 1 and args.launch in ("fused", "graph_region"):
            Bs, pps, ms = steady_state(args.steady_batch)
            ss = {"B": Bs, "poses_per_s": pps, "ms_per_launch": ms}
            if "valu_issue" in roof:
                ss["valu_frac"] = pps / roof["valu_issue"]["peak"]
            fl_l, fl_p = algorithmic_flops(N, it_mean + 1.0)
            ss["flops_frac"] = (fl_l + fl_p) * pps / 1e12 / FP64_VECTOR_PEAK_TFLOPS
            out["steady_state"] = ss
        if world == 1 and args.workload != "metric":
            out["dense"] = {k: dense_block(k, *v) for k, v in DENSE_WORKLOADS.items() if args.workload in ("all", k)}
        if world == 1 and args.workload == "all":
            out["test_time"] = {}
            for tt in ("zlmo", "zlmo_bf16", "glmo", "gsplmo", "hybrid_r03"):
                try:  # auxiliary blocks: whatever happens in one, the headline above is printed
                    out["test_time"][tt] = test_time_block(tt)
                except Exception as e:  # noqa: BLE001
                    out["test_time"][tt] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_head:
            # the third kernel family of the path (keypoint head), HBM-bound; its own line: bench_head.py
            from bench_head import measure_head
            h = measure_head(dev, steps=10, warmup=2)
            out["head"] = {"metric": h["metric"], "value": h["value"], "unit": h["unit"], "config": h["config"], "roofline": h["roofline"]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, N, seed=rank, budget_s=args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
