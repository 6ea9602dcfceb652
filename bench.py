#!/usr/bin/env python3
"""Headline benchmark: poses/sec of one "pose unit" = LC-loss forward+backward + one weighted-PnP solve,
B=256 poses x N=64 correspondences per GPU (BASELINE.json metric, configs[1]), fp32 I/O, synthetic inputs resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; the batch is sharded by pose (every pose is independent: SURVEY.md 8e), so there is no data-path
collective -- only the timing barrier/all-reduce.  Scaling is weak: each rank runs its own B=256 batch.
A step = one launch of the fused loss kernel (loss + d/d pts2d, d/d inv_std, d/d pts3d for the `.mean()` cotangent) and one
batched LM solve; the two are independent, so by default their workgroups share one grid (see --launch).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "poses/sec (cov-loss fwd+bwd + weighted PnP), B=256 N=64, 1/2/4/8 MI355X"  # BASELINE.json:metric
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(N: int, want_pts3d: bool):
    """SURVEY.md 8(d) per-pose figures."""
    loss = (36 + 28 + 96 + N * (12 + 8 + 8)) + (4 + N * (8 + 8)) + (N * 12 if want_pts3d else 0)
    pnp = (36 + 28 + N * (12 + 8 + 8)) + 36
    return loss, pnp


def static_counters(kernel: str, B: int, N: int):
    """Per-launch PMC counters of `kernel` on the default workload, from the newest committed rocprofv3 --pmc passes
    (profiles/<round>/pmc_traffic.json, written by scripts/profile_round.sh; PMC cannot be collected from inside a run).
    Returns (entry, source) -- source names the file, the round and the git SHA the passes were taken at -- or (None, None)."""
    if (B, N) != (256, 64):
        return None, None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get(kernel):
                meta = d.get("_meta", {})
                return d[kernel], {"file": os.path.relpath(f, ROOT), "git_sha": meta.get("git_sha", "unrecorded (round 1)"),
                                   "command": meta.get("command")}
        except Exception:
            pass
    return None, None


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


SIMD_CYCLES_PER_S = 1024 * 2.4e9  # 256 CUs x 4 SIMDs x 2.4 GHz (MI355X_MICROARCH.md chip-level parameters)


def valu_bound(sq: dict, B: int):
    """VALU-issue bound from SQ counters of one launch of B poses: SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD spends issuing
    VALU work; poses/s if all 1024 SIMDs issued such work every cycle."""
    cyc = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / B
    return cyc, SIMD_CYCLES_PER_S / cyc


def cpu_baseline(B, N, seed, budget_s=15.0):
    """The oracle (CPU restatement of the reference path) timed on this host: torch closed-form LC loss fwd+bwd on all
    cores + the C/OpenMP LM solve on all cores, over a bounded number of B-sized batches."""
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

    # B=256 poses of ~5 KB each do not scale to hundreds of host threads: pick the fastest of a few thread counts
    # (1 and 4 threads are reported too: BASELINE.md 3.3 -- 4 = the reference's num_workers, test.py:62,127)
    best, by_threads = None, {}
    for nt in sorted({min(avail, c) for c in (1, 4, 8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one(nt)
        dt1 = float("inf")
        for _ in range(3):  # best of three: shared hosts are noisy
            t0 = time.perf_counter()
            one(nt)
            dt1 = min(dt1, time.perf_counter() - t0)
        by_threads[str(nt)] = B / dt1
        if best is None or dt1 < best[1]:
            best = (nt, dt1)
    cores = best[0]
    torch.set_num_threads(cores)
    one_ = one
    one = lambda: one_(cores)  # noqa: E731
    one()  # warm-up (thread pools, page-in)
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 2000:
            break
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit="poses/s", cores=cores, kind="port", host_cores_available=avail, host_cpu=host_cpu_model(),
                poses_per_s_by_threads=by_threads,
                reference_anchor="un-restated reference, survey container (8-core Xeon, BASELINE.md section 2): lib.cov_mixed.Loss_cov_mixed "
                                 "fwd+bwd alone 36.9 ms per 256 poses = 6.9 k poses/s; the Ceres solve cannot be built or timed anywhere here",
                sample=f"{n} batches of B={B} N={N} (oracle: torch-CPU closed-form loss fwd+bwd + C/OpenMP LM, "
                       f"{cores} threads = fastest of 4..64), {dt:.1f} s")


def main():
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
                    help="graph_region (default): a step is ONE fused launch (lc_pose_unit_f32: loss and PnP workgroups share a grid) and "
                         "the K steps of a timed region are K kernel nodes of one hipGraph launch; fused: the same K launches issued one by "
                         "one from Python; eager: two launches per step on one stream; eager2: LM solve forked onto a second stream; "
                         "graph / graph2 / graph_fused: one step replayed as a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-head", action="store_true", help="skip the keypoint-head measurement attached as out['head']")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    # LC_BENCH_SHARE_GPU=1 (test only): every rank uses GPU 0 and the timing collectives run over gloo, so that the N > 1
    # control flow (barrier, max-over-ranks, aggregation) can be exercised on a one-GPU box; RCCL refuses two ranks per device.
    share_gpu = os.environ.get("LC_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            try:  # RCCL over xGMI; one tiny all-reduce proves the communicator works before anything is timed
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
                assert int(probe.item()) == world
            except Exception as e:  # noqa: BLE001 -- the data path has no collective: the backend only carries the timing barrier,
                # so a node whose RCCL cannot initialise still yields a valid per-N number (reported in collective_backend)
                print(f"bench.py rank {rank}: RCCL unavailable ({type(e).__name__}: {e}); timing barrier falls back to gloo", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                dist.init_process_group("gloo")
                share_gpu = True  # from here on: CPU tensors for the timing collectives (each rank keeps its own GPU)

    from lc_amd import _lib, synth

    lib = _lib.load()
    B, N = args.batch, args.npts
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=rank).items()}
    go = torch.full((B,), 1.0 / B, device=dev)
    sqrt_diag = b["inv_std"].contiguous()  # icov = inv_std^2 -> sqrt factor = inv_std (cer_solver.py:37-38)
    loss = torch.empty(B, device=dev)
    d_u, d_s, d_x = torch.empty_like(b["pts2d"]), torch.empty_like(b["inv_std"]), torch.empty_like(b["pts3d"])
    states = torch.empty_like(b["start"])
    tr = torch.empty(B, device=dev)
    ret = torch.empty(B, device=dev, dtype=torch.int32)
    P = _lib.ptr

    def launch_loss():
        rc = lib.lc_cov_loss_fwd_bwd_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None,
                                         P(b["bbox_3d"]), P(go), B, N, 32.0, 3.0, 4.0, P(loss), P(d_u), P(d_s), P(d_x), None,
                                         _lib.stream_ptr(dev))
        assert rc == 0

    def launch_pnp():
        # start poses are read-only input, states is output: every step solves from the same perturbed pose
        rc = lib.lc_pnp_lm_f32(P(b["K"]), P(b["pts3d"]), P(b["pts2d"]), None, P(sqrt_diag), None, P(b["start"]), P(states), P(tr),
                               P(ret), None, B, N, 50, 1e-6, _lib.stream_ptr(dev))
        assert rc == 0

    side = torch.cuda.Stream(dev)

    def step_serial():
        launch_loss()
        launch_pnp()

    def step_forked():
        # the two kernels are independent (the loss linearises at the GT pose, the solve starts from `start`):
        # fork the LM solve onto a second HIP stream, join at the end of the step
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            launch_pnp()
        launch_loss()
        main.wait_stream(side)

    def step_fused():
        # one grid for both halves of the pose unit (lc_amd/csrc/lc_fused.hip)
        rc = lib.lc_pose_unit_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]),
                                  P(go), B, N, 32.0, 3.0, 4.0, P(loss), P(d_u), P(d_s), P(d_x), P(sqrt_diag), P(b["start"]),
                                  P(states), P(tr), P(ret), 50, 1e-6, _lib.stream_ptr(dev))
        assert rc == 0

    # --launch streams (NOT the default, reported separately in DESIGN.md): `slots` independent B-sized batches in flight, each
    # with its own inputs, outputs and stream -- the serving-side picture (independent requests), where a 512-workgroup step
    # leaves most of the chip idle.  A step is still one fused launch over one batch; steps of different slots overlap.
    slot_state = []
    if args.launch == "streams":
        for s_i in range(args.slots):
            bb = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1000 * (rank + 1) + s_i).items()}
            outs = dict(loss=torch.empty(B, device=dev), d_u=torch.empty_like(bb["pts2d"]), d_s=torch.empty_like(bb["inv_std"]),
                        d_x=torch.empty_like(bb["pts3d"]), states=torch.empty_like(bb["start"]), tr=torch.empty(B, device=dev),
                        ret=torch.empty(B, device=dev, dtype=torch.int32), sd=bb["inv_std"].contiguous())
            slot_state.append((bb, outs, torch.cuda.Stream(dev)))
        torch.cuda.synchronize(dev)
    slot_i = [0]

    def step_streams():
        bb, o, stream = slot_state[slot_i[0] % len(slot_state)]
        slot_i[0] += 1
        rc = lib.lc_pose_unit_f32(P(bb["K"]), P(bb["pose"]), P(bb["pts3d"]), P(bb["pts2d"]), P(bb["inv_std"]), None, P(bb["bbox_3d"]),
                                  P(go), B, N, 32.0, 3.0, 4.0, P(o["loss"]), P(o["d_u"]), P(o["d_s"]), P(o["d_x"]), P(o["sd"]),
                                  P(bb["start"]), P(o["states"]), P(o["tr"]), P(o["ret"]), 50, 1e-6, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0

    if args.launch in ("fused", "graph_region") and N > 64:
        args.launch = "eager"
    step_eager = {"eager": step_serial, "graph": step_serial, "fused": step_fused, "graph_fused": step_fused, "graph_region": step_fused,
                  "streams": step_streams}.get(args.launch, step_forked)
    graph = None
    if args.launch.startswith("graph"):
        warm = torch.cuda.Stream(dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):
            step_eager()
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        from lc_amd.inference import quiet_capture  # no Python GC while the stream is capturing

        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: calls of other threads of the process (e.g. the RCCL watchdog of a multi-rank run) neither fail nor invalidate the capture
            with quiet_capture(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(args.steps if args.launch == "graph_region" else 1):  # graph_region: the K steps of a region as ONE graph
                    step_eager()
            graph.replay()
            torch.cuda.synchronize(dev)
        except Exception as e:  # noqa: BLE001 -- a runtime that cannot capture still yields the stream-order number, labelled as such
            if args.launch != "graph_region":
                raise
            print(f"bench.py rank {rank}: hipGraph capture of the region failed ({type(e).__name__}: {e}); issuing the launches one by one",
                  file=sys.stderr)
            graph, args.launch = None, "fused"
            torch.cuda.synchronize(dev)
    step = step_eager if graph is None else graph.replay
    steps_per_call = args.steps if args.launch == "graph_region" else 1

    for _ in range(max(1, args.warmup // steps_per_call)):
        step()

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    from lc_amd import dist as lcd

    region_s = []
    for _ in range(max(1, args.regions)):
        fence()  # synchronize + barrier + synchronize: every rank starts the region together
        t0 = time.perf_counter()
        for _ in range(args.steps // steps_per_call):
            step()
        torch.cuda.synchronize(dev)  # this rank's K steps are done ...
        region_s.append(time.perf_counter() - t0)
        if dist is not None:
            dist.barrier()  # ... and the region ends when the slowest rank is (MAX over ranks below); the collective's own
            # latency (tens of us over RCCL) is not part of the K steps
    agg = lcd.aggregate_regions(region_s, args.steps, device="cpu" if (share_gpu or dist is None) else dev)
    elapsed = agg["median_region_s"]
    if args.launch == "streams":
        assert all(int(o["ret"].sum().item()) == 0 and bool(torch.isfinite(o["loss"]).all()) for _, o, _ in slot_state)
    else:
        assert int(ret.sum().item()) == 0 and bool(torch.isfinite(loss).all())

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
        bb = {k: v.to(dev) for k, v in synth.make_batch(Bs, N, seed=977).items()}
        o_loss = torch.empty(Bs, device=dev)
        o = [torch.empty_like(bb["pts2d"]), torch.empty_like(bb["inv_std"]), torch.empty_like(bb["pts3d"]), torch.empty_like(bb["start"]),
             torch.empty(Bs, device=dev), torch.empty(Bs, device=dev, dtype=torch.int32)]
        gos = torch.full((Bs,), 1.0 / Bs, device=dev)
        sd = bb["inv_std"].contiguous()

        def one():
            rc = lib.lc_pose_unit_f32(P(bb["K"]), P(bb["pose"]), P(bb["pts3d"]), P(bb["pts2d"]), P(bb["inv_std"]), None, P(bb["bbox_3d"]),
                                      P(gos), Bs, N, 32.0, 3.0, 4.0, P(o_loss), P(o[0]), P(o[1]), P(o[2]), P(sd), P(bb["start"]),
                                      P(o[3]), P(o[4]), P(o[5]), 50, 1e-6, _lib.stream_ptr(dev))
            assert rc == 0
        ms = kernel_ms(one, reps=6, windows=5)
        assert int(o[5].sum().item()) == 0
        return Bs, Bs / (ms * 1e-3), ms

    if rank == 0:
        t_loss = kernel_ms(launch_loss)
        t_pnp = kernel_ms(launch_pnp)
        by_loss, by_pnp = algorithmic_bytes(N, True)
        dom = ("lc_cov_loss_kernel", t_loss, by_loss) if t_loss >= t_pnp else ("lc_pnp_lm_kernel", t_pnp, by_pnp)
        kernel_us = {"lc_cov_loss_kernel": t_loss * 1e3, "lc_pnp_lm_kernel": t_pnp * 1e3}
        if args.launch in ("fused", "graph_fused", "graph_region"):
            t_unit = kernel_ms(step_fused)  # launches issued one by one (stream order)
            kernel_us["lc_pose_unit_kernel"] = t_unit * 1e3
            if args.launch == "graph_region":  # the pattern of the timed region: the kernel as a node of the K-step graph (events around replays)
                kernel_us["lc_pose_unit_kernel_stream_order"] = t_unit * 1e3
                t_unit = kernel_ms(graph.replay, reps=max(1, 200 // args.steps)) / args.steps
                kernel_us["lc_pose_unit_kernel"] = t_unit * 1e3
            dom = ("lc_pose_unit_kernel", t_unit, by_loss + by_pnp)
            kernel_us["lc_pose_unit_kernel_p10_p50_p90"] = kernel_percentiles_us(step_fused)
        hbm_gbs = dom[2] * B / (dom[1] * 1e-3) / 1e9
        kernel_poses_per_s = B / (dom[1] * 1e-3)
        ctr, src = static_counters(dom[0], B, N)
        hbm = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_pose": {"loss": by_loss, "pnp": by_pnp},
               "note": "secondary: 5 KB working set per pose, one wave per pose -- HBM never binds this kernel"}
        if ctr is not None and ctr.get("sq", {}).get("SQ_ACTIVE_INST_VALU"):
            cyc, peak = valu_bound(ctr["sq"], B)
            roof = {"bound": "valu_issue", "kernel": dom[0], "achieved": kernel_poses_per_s, "peak": peak, "unit": "poses/s",
                    "frac": kernel_poses_per_s / peak, "traffic": ctr.get("bytes_per_launch"),
                    "valu_insts_per_pose": ctr["sq"]["SQ_INSTS_VALU"] / B, "simd_cycles_per_pose": cyc,
                    "note": "the bound that binds: 1024 SIMDs x 2.4 GHz / VALU-issue SIMD-cycles per pose (SQ_ACTIVE_INST_VALU); achieved = "
                            "B / event-timed launch duration of the kernel, measured in this run",
                    "counters_from": src, "kernel_us": kernel_us, "hbm": hbm}
        else:  # no committed counter pass for this workload: only the HBM figure can be formed in-run
            roof = dict(hbm, kernel=dom[0], traffic=None, kernel_us=kernel_us)
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
                       "launch": args.launch, **({"slots_in_flight": args.slots} if args.launch == "streams" else {})},
            "timing": {"protocol": f"{agg['regions']} regions of exactly {args.steps} steps, barrier + synchronize around each, MAX over ranks "
                                   f"per region, MEDIAN region reported"
                                   + ("; the K steps of a region are K kernel nodes of one hipGraph launch" if args.launch == "graph_region" else ""),
                       "region_ms_per_step": agg["region_ms_per_step"]},
            "ranks_seen": agg["ranks_seen"], "collective_backend": agg["backend"],
            "rccl_version": rccl_version() if (world > 1 and not share_gpu) else None,
            "per_rank_ms_per_step": agg["per_rank_ms_per_step"],
            "roofline": roof,
        }
        if args.steady_batch > 0 and world == 1 and args.launch in ("fused", "graph_region"):
            Bs, pps, ms = steady_state(args.steady_batch)
            ss = {"B": Bs, "poses_per_s": pps, "ms_per_launch": ms}
            if roof.get("bound") == "valu_issue":
                ss["valu_frac"] = pps / roof["peak"]
            out["steady_state"] = ss
        if world == 1 and not args.no_head:
            # the third kernel family of the path (keypoint head), HBM-bound; its own line: bench_head.py
            from bench_head import measure_head
            h = measure_head(dev, steps=10, warmup=2)
            out["head"] = {"metric": h["metric"], "value": h["value"], "unit": h["unit"], "config": h["config"], "roofline": h["roofline"]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, N, seed=rank, budget_s=args.cpu_budget)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
