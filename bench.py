#!/usr/bin/env python
"""FF-RAFT hot-path benchmark (BASELINE.json metric: frame-pairs/sec, 384x512, iters=12).

One process per GPU.  A "step" = one forward pass of the full hot path
(CCE encoders -> corr volume + pyramid -> 12 x {lookup, update block, convex
upsample}) over one batch of synthetic frame pairs already resident in HBM.
Forward is embarrassingly data-parallel: ranks hold independent batches, no
data-path collective ("scaling": "weak").

Prints ONE JSON line on rank 0 (see the driver contract); extra objects:
  roofline     - the CorrBlock lookup kernel (HBM-bound), timed live with HIP
                 events on the launch stream inside the timed region
  cpu_baseline - the CPU oracle (a restatement of the reference pinned by golden
                 vectors; kind "port") timed on this box's host cores, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DTYPES = {"f16x3": "f32 via fp16x3 split operands on the f16 MFMA pipe (f32 accumulate, fp32-level accuracy)",
          "fp32": "f32", "f16": "f16 operands, f32 accumulate (reduced precision)"}
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md: dense f16/bf16 MFMA, fp32 MFMA
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
# L*(2r+2)^2*s_corr (windows) + L*(2r+1)^2*4 (fp32 output) + 8 (coords); s_corr = 4 (fp32 pyramid) or 2 (fp16)
LOOKUP_BYTES_PER_QUERY = {"fp32": 4 * 100 * 4 + 4 * 81 * 4 + 8, "fp16": 4 * 100 * 2 + 4 * 81 * 4 + 8}   # 2904 / 2104


def cfg():
    from argparse import Namespace
    return Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                     MODEL=Namespace(FUSION_TYPE="1x1conv", LOAD_MODULE_TO_BRANCH=False))


def shard_units(total_units: int, world: int, rank: int):
    """Contiguous split of `total_units` independent frame pairs over ranks."""
    base, rem = divmod(total_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def synthetic_batch(b, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    i1 = torch.randint(0, 256, (b, 3, h, w), generator=g).float()
    i2 = torch.roll(i1, shifts=(3, -5), dims=(2, 3)) + torch.randn(b, 3, h, w, generator=g) * 2
    m1 = (torch.rand(b, 1, h, w, generator=g) < 500.0 / (h * w)).float() * 255   # ORB-like, 500 points
    return [t.contiguous().to(device) for t in (i1, i2.clamp(0, 255), m1, torch.zeros_like(m1))]


def host_cores() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def refuse_lab_switches():
    """Timing-only ablations return WRONG results by design and tuning overrides change what is measured: a stray variable in
    the caller's shell must not reach a bench line.  (The library refuses the ablation variables as well: _hip.load.)"""
    from focusflow_official_amd import _hip
    bad = _hip.lab_variables_set()
    if bad:
        raise SystemExit(f"bench.py: refusing to run with lab switches set in the environment: {', '.join(bad)} "
                         "(timing-only ablations and tuning overrides; unset them)")


def host_issue_time(step, n=5):
    """Host time to ISSUE one step (no synchronisation inside, the queue drained before every sample) next to the wall
    time of the same step: with N ranks sharing one host, a rank whose issue time approaches its step time is host-starved,
    not slow on the GPU - the two cannot be told apart from a throughput number alone."""
    issue, total = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        issue.append(t1 - t0)
        total.append(time.perf_counter() - t0)
    issue.sort()
    total.sort()
    return round(issue[len(issue) // 2] * 1e3, 3), round(total[len(total) // 2] * 1e3, 3)


def pmc_traffic(queries, pyramid):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r05_lookup_traffic.json, r05_lookup_traffic_fp16.json:
    TCC_EA0_RDREQ x 128 B + WRITE_SIZE), scaled per query, and where the number comes from; (None, reason) if the
    profile is absent.  PMC passes cannot run inside the timed process: both profiles hold the lookup launches of bench.py
    itself under rocprofv3 --pmc (tools/prof_pmc.sh ... bench.py: the headline command for fp32, --batch 16 --height 544
    --width 960 --iters 32 --pyramid fp16 for BASELINE configs[4]) - NOT a counter read in this run."""
    name = "r05_lookup_traffic.json" if pyramid == "fp32" else "r05_lookup_traffic_fp16.json"
    if not os.path.exists(os.path.join(ROOT, "profiles", name)):
        name = "r04_lookup_traffic.json" if pyramid == "fp32" else "r03_lookup_traffic_fp16.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        return (int(d["traffic_bytes_per_launch"] / d["queries_per_launch"] * queries),
                f"profiles/{name}: separate rocprofv3 --pmc passes ({d.get('workload', 'same kernel')}; {d['queries_per_launch']} queries "
                f"per launch), scaled per query; not measured in this run")
    except (OSError, KeyError, ValueError):
        return None, f"profiles/{name} missing"


def cpu_baseline(h, w, iters, budget_s=20.0, keep=None):
    """Time the CPU oracle on the SAME workload shape (B=1), bounded to ~budget_s.  keep: a dict that receives the
    checker's weights, input and output of that forward (the parity leg compares the HIP path against them)."""
    from oracle import ffraft_ref as orc
    from oracle.weights import det_tensor
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_spec.json")) as f:
        sd = {k: det_tensor(k, s) for k, s, _ in json.load(f)}
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    inp = orc.shifted_pair(1, h, w, seed=1234)
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = orc.ffraft_forward(sd, *inp, raft_iters=iters, test_mode=True)  # warm-up
        log(f"cpu_baseline: warm-up forward {time.perf_counter() - t0:.2f} s")
        if keep is not None:
            keep.update(sd=sd, inp=inp, flow_up=ref[1])
        times = []
        t_end = time.perf_counter() + budget_s
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 30):
            t0 = time.perf_counter()
            orc.ffraft_forward(sd, *inp, raft_iters=iters, test_mode=True)
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} forwards of B=1 {h}x{w} iters={iters} fp32 (median {med * 1e3:.1f} ms), "
                      f"oracle/ffraft_ref.py on torch CPU"}


def timed_region(step, steps, world, sync, device=None):
    """Barrier + sync, K steps, sync + barrier; returns MAX elapsed over ranks (contract)."""
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed, out


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as CHILD processes with the
    torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, train.py:94-132 of the reference reads the same
    variables) and relay their output.  The parent never touches the GPU and never replaces itself (no exec):
    it waits for the children and exits with the worst return code.  If a rank dies the others are terminated."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    alive = set(range(n))
    kill_at = None                     # a rank stuck in a collective or a GPU wait may ignore SIGTERM: SIGKILL after a grace period
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                log(f"rank {r} exited with {code}: stopping the other ranks")
                for o in alive:
                    procs[o].terminate()
                kill_at = time.monotonic() + 10.0
        if kill_at is not None and alive and time.monotonic() > kill_at:
            for o in alive:
                log(f"rank {o} did not stop within 10 s of SIGTERM: killing it")
                procs[o].kill()
            kill_at = None
        time.sleep(0.05)
    return rc


def harness_selftest(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    lo, hi = shard_units(args.batch * world, world, rank)
    owned = torch.zeros(args.batch * world, dtype=torch.int64)
    owned[lo:hi] = 1
    if world > 1:
        dist.all_reduce(owned)
    assert bool((owned == 1).all()), "shards must tile the global batch exactly once"
    delay = 0.01 * (1 + rank)                       # rank r is slower: MAX must pick the last rank's time
    elapsed, _ = timed_region(lambda: time.sleep(delay), args.steps, world, lambda: None)
    assert elapsed >= 0.01 * world * args.steps * 0.95
    if rank == 0:
        pairs = args.batch * world * args.steps
        print(json.dumps({"metric": "harness-selftest", "value": pairs / elapsed, "n_gpus": world, "steps": args.steps,
                          "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak", "shard": [lo, hi],
                          # what the real forward bench would do with these flags (N > 1: hipGraph replay unless --eager)
                          "hipgraph": bool(args.graph or (world > 1 and not args.eager))}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def train_setup(args, world, rank, local_rank, device):
    """One training step = train.py:291-328 on synthetic FlyingChairs-shaped data: forward (train mode),
    MixLoss (ffraft_chairs_orb.yaml:35-39), loss *= world_size, backward (+ DDP all-reduce over RCCL),
    clip_grad_norm_(1.0), AdamW + OneCycleLR step.  Optimiser/scheduler are stock PyTorch as in the reference.
    Returns (step closure, h, w)."""
    from torch.nn.parallel import DistributedDataParallel
    from focusflow_official_amd import FF_RAFT_FUSION
    from focusflow_official_amd.losses import build_losses
    h, w = (368, 496) if (args.height, args.width) == (384, 512) else (args.height, args.width)
    torch.manual_seed(1234)
    model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg()).to(device).train()
    net = DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=False) if world > 1 else model
    opt = torch.optim.AdamW(model.parameters(), lr=4e-4, weight_decay=1e-5, eps=1e-8)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, 4e-4, 100000, pct_start=0.05, cycle_momentum=False, anneal_strategy="linear")
    crit = build_losses("MixLoss", gamma=0.8, max_flow=400, kernel_size=1, sigma=0.01, lamda=1)
    batch = synthetic_batch(args.batch, h, w, 1234 + rank, device)
    g = torch.Generator().manual_seed(99 + rank)
    flow_gt = (torch.randn(args.batch, 2, h, w, generator=g) * 5).clamp(-400, 400).to(device)
    valid = torch.ones(args.batch, h, w, device=device)

    def step():
        preds = net(*batch, raft_iters=args.iters)
        loss, _ = crit(preds, flow_gt, valid, batch[2])
        opt.zero_grad(set_to_none=True)
        (loss * world).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sched.step()
        return loss

    return step, h, w


def train_mode(args, world, rank, local_rank, device, ranks_seen=1):
    step, h, w = train_setup(args, world, rank, local_rank, device)
    log(f"train mode rank {rank}/{world}: {args.batch} pairs {h}x{w}")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    elapsed, loss = timed_region(step, args.steps, world, torch.cuda.synchronize, device)
    assert torch.isfinite(loss)
    if rank == 0:
        pairs = args.batch * world * args.steps
        print(json.dumps({
            "metric": "training frame-pairs/sec FF-RAFT 368x496 iters=12 (fwd+MixLoss+bwd+AdamW)",
            "value": round(pairs / elapsed, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (forward convs: " + ops_precision() + "; backward convs: f16x3 with power-of-two gradient scaling; corr-volume backward fp32 MFMA)",
            "data": "synthetic",
            "config": {"workload": f"FF-RAFT training step, {args.batch} pairs/GPU {h}x{w}, iters={args.iters}, MixLoss "
                                   f"(k=1, sigma=0.01, lamda=1), AdamW + OneCycleLR, clip 1.0 (BASELINE configs[2] shape)",
                       "pairs_per_gpu": args.batch, "parallelism": f"dp{world}" + (" DDP/RCCL all-reduce 30.65 MB" if world > 1 else "")},
            "final_loss": loss.detach().item(), "rccl_ranks_seen": ranks_seen}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def memory_only_companion(launch_bytes, achieved_gbs, device, forward=None):
    """What this part delivers to ANY kernel that moves as many bytes as one lookup launch: ff_probe_memory_kernel
    (probe.hip) - 4096 one-wave blocks like the lookup, per wave and trip five 1 KB loads from random 128-byte segments
    of a 2 GB buffer (nothing left in the 256 MB last-level cache between launches) and four 1 KB stores, no arithmetic -
    timed with the same dispatch-bound events.  The lookup's distance to THIS number is what is left for the kernel."""
    from focusflow_official_amd import ops
    blocks = 4096
    trips = max(1, int(round(launch_bytes / (blocks * 9 * 1024))))
    src = torch.empty(2 << 30, dtype=torch.uint8, device=device)
    src.fill_(1)
    dst = torch.empty(blocks * trips * 4096 + 16, dtype=torch.uint8, device=device)
    for salt in range(3):
        rd, wr = ops.probe_memory_kernel(src, dst, 128, blocks, trips, salt)
    torch.cuda.synchronize()
    ops.launch_timing_begin(ops.TIME_PROBE)
    for salt in range(3, 23):
        ops.probe_memory_kernel(src, dst, 128, blocks, trips, salt)
    n, tot, lo, hi = ops.launch_timing_end(ops.TIME_PROBE)
    # ... and the same kernel INSIDE the pipeline: one extra (untimed) forward in which every lookup is followed by a
    # probe launch on the same stream - behind the same convolutions, at the clocks and with the cache contents the
    # lookup itself meets there
    piped, placed = None, {}
    if forward is not None:
        salt = [100]

        def probe():
            salt[0] += 1
            ops.probe_memory_kernel(src, dst, 128, blocks, trips, salt[0])

        def run_with(name, before):
            """One extra forward with a probe launch before (or after) every call of ops.<name>."""
            orig = getattr(ops, name)

            def hooked(*a, **k):
                if before:
                    probe()
                r = orig(*a, **k)
                if not before:
                    probe()
                return r
            setattr(ops, name, hooked)
            try:
                ops.launch_timing_begin(ops.TIME_PROBE)
                forward()
                return ops.launch_timing_end(ops.TIME_PROBE)
            finally:
                setattr(ops, name, orig)
        piped = run_with("corr_lookup_tiled", False)
        # Two more placements: where the lookup itself stands - its predecessor is the up-sampling kernel,
        # which has just written 25 MB - and behind the flow head, a predecessor that leaves 0.2 MB dirty
        for key, name, note in (("in_pipeline_before_lookup", "corr_lookup_tiled", "issued right before every lookup: behind the up-sampling kernel (25 MB of flow_up just written) - the lookup's own place"),
                                ("in_pipeline_behind_flow_head", "mask_upsample", "issued right behind the flow head's last convolution (0.2 MB written), before the up-sampling kernel")):
            try:
                t = run_with(name, True)
                if t and t[0]:
                    placed[key] = {"launches": t[0], "avg_launch_us": round(t[1] / t[0], 2), "min_launch_us": round(t[2], 2), "max_launch_us": round(t[3], 2), "note": note}
            except Exception as e:          # noqa: BLE001
                placed[key] = {"error": f"{type(e).__name__}: {e}"}
    del src, dst
    torch.cuda.empty_cache()
    avg = tot / max(1, n)
    gbs = (rd + wr) / (avg * 1e-6) / 1e9 if avg > 0 else 0.0
    return {"kernel": "probe_kernel (ff_probe_memory_kernel): loads from random 128-byte segments + streaming stores 5 : 4, no arithmetic, 4096 one-wave blocks",
            "bytes_per_launch": rd + wr, "launches": n, "avg_launch_us": round(avg, 2), "min_launch_us": round(lo, 2), "max_launch_us": round(hi, 2),
            "achieved": round(gbs, 1), "unit": "GB/s", "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
            "lookup_over_this": round(achieved_gbs / gbs, 4) if gbs > 0 else None,
            "note": "back-to-back launches on an otherwise idle chip; the lookup is timed inside the pipeline",
            "in_pipeline": None if not piped or not piped[0] else {
                "launches": piped[0], "avg_launch_us": round(piped[1] / piped[0], 2), "min_launch_us": round(piped[2], 2), "max_launch_us": round(piped[3], 2),
                "note": "the same launch issued right behind every lookup of one extra untimed forward"},
            **placed}


def _epe(a, b):
    return torch.sqrt(((a - b) ** 2).sum(1))


def parity_and_reduced_precision(args, device, checker):
    """Against the checker's forward of the cpu_baseline leg (the pinned CPU port of the reference, same weights, same
    pair): the headline arithmetic (f16x3) and the reduced-precision throughput mode (FF_CONV_PRECISION=f16: single-term
    fp16 operands, 10 mantissa bits like the TF32 the reference runs on a GPU - common.py:25-27) - BASELINE configs[1]
    "... vs reference EPE".  Then the throughput of the reduced-precision mode on the headline workload."""
    from focusflow_official_amd import FF_RAFT_FUSION, ops
    out = {}
    prev = ops.conv_precision()
    try:
        for mode in ("f16x3", "f16"):
            ops.set_conv_precision(mode)
            m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg())
            if checker:
                m.load_state_dict(checker["sd"], strict=True)
                m = m.to(device).eval()
                with torch.no_grad():
                    fu = m(*[t.to(device) for t in checker["inp"]], raft_iters=args.iters, test_mode=True)[1].cpu()
                e = _epe(fu, checker["flow_up"])
                res = {"epe_vs_reference_px": round(float(e.mean()), 6), "epe_max_px": round(float(e.max()), 6),
                       "max_abs_px": round(float((fu - checker["flow_up"]).abs().max()), 6),
                       "flow_absmax_px": round(float(checker["flow_up"].abs().max()), 3),
                       "sample": f"1 pair {args.height}x{args.width} iters={args.iters}, deterministic test weights, against the CPU port of the reference (cpu_baseline leg)"}
            else:
                res = {"epe_vs_reference_px": None}
            if mode == "f16x3":
                out["parity"] = dict(res, dtype=DTYPES[mode])
                continue
            torch.manual_seed(1234)
            m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg()).to(device).eval()
            batch = synthetic_batch(args.batch, args.height, args.width, 1234, device)
            with torch.no_grad():
                for _ in range(3):
                    o = m(*batch, raft_iters=args.iters, test_mode=True)
                torch.cuda.synchronize()
                n = 10
                t0 = time.perf_counter()
                for _ in range(n):
                    o = m(*batch, raft_iters=args.iters, test_mode=True)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            out["reduced_precision"] = dict(res, metric=f"frame-pairs/sec FF-RAFT {args.height}x{args.width} iters={args.iters}",
                                            value=round(args.batch / dt, 2), unit="frame-pairs/s", ms_per_step=round(dt * 1e3, 3), steps=n, warmup=3,
                                            dtype=DTYPES[mode], finite=bool(torch.isfinite(o[1]).all()),
                                            workload=f"BASELINE configs[1] in its reduced-precision reading: {args.batch} pairs, NOT the headline (north_star: 1e-3 px vs fp32)",
                                            tf32_level_px="tests/golden/tf32_epe_384x512.json: the reference's own TF32 arithmetic costs 0.022 px mean / 0.21 px max on the test pair")
            del m, batch
    except Exception as e:          # noqa: BLE001 - reported, never fatal for the headline line
        out["reduced_precision_error"] = f"{type(e).__name__}: {e}"
    finally:
        ops.set_conv_precision(prev)
    torch.cuda.empty_cache()
    return out


def config4_measurements(device):
    """BASELINE configs[4]: FF-RAFT 540x960 (padded to 544x960), 32 iterations, fp16 correlation pyramid; 1, 4, 16 and 32 pairs per step."""
    from focusflow_official_amd import FF_RAFT_FUSION, ops
    out = {}
    try:
        torch.manual_seed(1234)
        m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg()).to(device).eval()
        m.flow_net.corr_pyramid_dtype = "fp16"
        for b in (1, 4, 16, 32):      # (177 MB of fp16 pyramid per pair: 5.7 GB at 32 pairs - the batch at which this config is a bandwidth stress)
            batch = synthetic_batch(b, 544, 960, 77 + b, device)
            with torch.no_grad():
                for _ in range(2):
                    o = m(*batch, raft_iters=32, test_mode=True)
                torch.cuda.synchronize()
                n = 5 if b <= 4 else 3
                ops.launch_timing_begin(ops.TIME_LOOKUP)
                t0 = time.perf_counter()
                for _ in range(n):
                    o = m(*batch, raft_iters=32, test_mode=True)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            lk = ops.launch_timing_end(ops.TIME_LOOKUP)
            q = b * 68 * 120 * 32 * n // max(1, lk[0])     # (beyond the 4 GB buffer resource the pyramid is looked up in batch chunks: several launches per iteration)
            us = lk[1] / max(1, lk[0])
            out[f"pairs_{b}"] = {"value": round(b / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": n, "warmup": 2,
                                 "finite": bool(torch.isfinite(o[1]).all()),
                                 "lookup": {"queries_per_launch": q, "launches": lk[0], "avg_launch_us": round(us, 2),
                                            "frac_of_hbm_peak": round(q * LOOKUP_BYTES_PER_QUERY["fp16"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None}}
            del batch
        out["workload"] = "BASELINE configs[4]: FF-RAFT forward 544x960 (540 padded), iters=32, fp16 correlation pyramid, SiLK-free ORB-like mask"
    except Exception as e:          # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    torch.cuda.empty_cache()
    return out


def graph_replay_measurement(args, device):
    """The headline step replayed from a captured hipGraph (focusflow_official_amd.graph.GraphedForward: the context encoder
    forked beside the feature encoder inside the capture) - the same work, all twelve up-samplings, no host in the loop.
    Not the headline: bench.py's roofline needs the library's own launches (dispatch-bound events), which a replay does not make."""
    from focusflow_official_amd import FF_RAFT_FUSION
    from focusflow_official_amd.graph import GraphedForward
    try:
        torch.manual_seed(1234)
        m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg()).to(device).eval()
        batch = synthetic_batch(args.batch, args.height, args.width, 7, device)
        g = GraphedForward(m, batch, raft_iters=args.iters)
        for _ in range(3):
            o = g(*batch)
        torch.cuda.synchronize()
        n, dt = 10, None
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(n):
                o = g(*batch)
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / n
            dt = d if dt is None or d < dt else dt
        with torch.no_grad():
            ref = m(*batch, raft_iters=args.iters, test_mode=True)
        res = {"value": round(args.batch / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": n, "batches": "faster of 2",
               "finite": bool(torch.isfinite(o[1]).all()), "max_abs_vs_eager_px": float((o[1] - ref[1]).abs().max()),
               "workload": f"the headline step ({args.batch} pairs {args.height}x{args.width}, iters={args.iters}) replayed from a hipGraph"}
        del g
    except Exception as e:          # noqa: BLE001
        res = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return res


def batch16_measurement(args, device):
    """The headline workload at 16 pairs per GPU instead of the 8 that BASELINE configs[1] names: what the per-launch ramps of
    the update loop cost at batch 8 (every kernel of the loop is a 5-100 us launch at 1/8 resolution)."""
    from focusflow_official_amd import FF_RAFT_FUSION, ops
    try:
        torch.manual_seed(1234)
        m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg()).to(device).eval()
        batch = synthetic_batch(16, args.height, args.width, 99, device)
        with torch.no_grad():
            for _ in range(2):
                o = m(*batch, raft_iters=args.iters, test_mode=True)
            torch.cuda.synchronize()
            n = 5
            dt, lk = None, None
            for _ in range(2):       # two batches of n steps, the faster one is reported (a secondary figure measured once, seconds
                                     # after other models' buffers were released: a single allocator stall would halve it)
                ops.launch_timing_begin(ops.TIME_LOOKUP)
                t0 = time.perf_counter()
                for _ in range(n):
                    o = m(*batch, raft_iters=args.iters, test_mode=True)
                torch.cuda.synchronize()
                d = (time.perf_counter() - t0) / n
                l = ops.launch_timing_end(ops.TIME_LOOKUP)
                if dt is None or d < dt:
                    dt, lk = d, l
        q = 16 * (args.height // 8) * (args.width // 8)
        us = lk[1] / max(1, lk[0])
        res = {"value": round(16 / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": n, "warmup": 2, "batches": "faster of 2",
               "finite": bool(torch.isfinite(o[1]).all()),
               "lookup": {"queries_per_launch": q, "launches": lk[0], "avg_launch_us": round(us, 2),
                          "frac_of_hbm_peak": round(q * LOOKUP_BYTES_PER_QUERY["fp32"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None},
               "workload": f"FF-RAFT forward {args.height}x{args.width}, iters={args.iters}, fp32 pyramid, 16 pairs per step (not the headline batch)"}
    except Exception as e:          # noqa: BLE001
        res = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return res


def pwc_measurements(device):
    """BASELINE configs[3]: FF-PWC forward on 448x1024 pairs - one pair eager and replayed from a hipGraph (a one-pair forward is
    a chain of small launches: host-bound when issued eagerly), eight pairs eager, and the cost-volume kernel (correlation.py:34-102)
    against the HBM roofline: algorithmic bytes of a launch (both feature maps read once, 81 channels written once) / its time."""
    from argparse import Namespace
    from focusflow_official_amd import ops
    from focusflow_official_amd.pwcnet import FF_PWCNET
    out = {"workload": "BASELINE configs[3]: FF_PWCNET test_mode, 448x1024 pairs, SIFT-like mask (2000 points), random-init weights"}
    try:
        pcfg = Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"), MODEL=Namespace(FUSION="parallel", FUSION_TYPE="1x1conv"))
        torch.manual_seed(0)
        m = FF_PWCNET(pcfg).to(device).eval()
        with torch.no_grad():       # unnormalised 0..255 inputs: keep the first layer's activations in fp16 range
            m.netExtractor.netOne[0].weight.mul_(1 / 255.0)
            m.netExtractor.mask_netOne[0].weight.mul_(1 / 255.0)

        def inputs(b):
            g = torch.Generator().manual_seed(0)
            i1 = torch.randint(0, 256, (b, 3, 448, 1024), generator=g).float().to(device)
            i2 = torch.roll(i1, (3, -5), (2, 3))
            m1 = ((torch.rand(b, 1, 448, 1024, generator=g) < 2000 / (448 * 1024)).float() * 255).to(device)
            return i1, i2, m1, m1

        def timeit(f, n):
            for _ in range(3):
                o = f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                o = f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, o

        with torch.no_grad():
            inp1 = inputs(1)
            dt, ref = timeit(lambda: m(*inp1, test_mode=True), 10)
            out["pairs_1_eager"] = {"value": round(1 / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": 10, "warmup": 3,
                                    "finite": bool(torch.isfinite(ref).all())}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    m(*inp1, test_mode=True)
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                gout = m(*inp1, test_mode=True)

            def rep():
                gr.replay()
                return gout
            dt, o = timeit(rep, 20)
            out["pairs_1_hipgraph_replay"] = {"value": round(1 / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": 20, "warmup": 3,
                                              "max_abs_vs_eager_px": float((o - ref).abs().max())}
            del gr, gout
            inp8 = inputs(8)
            dt, o = timeit(lambda: m(*inp8, test_mode=True), 5)
            out["pairs_8_eager"] = {"value": round(8 / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 3), "steps": 5, "warmup": 3,
                                    "finite": bool(torch.isfinite(o).all())}
            # the cost-volume launches of one more 8-pair forward, each bracketed by an event pair on its stream
            ops.profile_begin("pwc_costvolume")
            m(*inp8, test_mode=True)
            notes = ops.profile_notes("pwc_costvolume")
            times = ops.profile_end()["pwc_costvolume"]
            levels = []
            for ms, (nbytes, shape, splits) in zip(times, notes):
                levels.append({"B_H_W_C": list(shape), "us": round(ms * 1e3, 2), "algorithmic_bytes": nbytes, "k_splits": splits,
                               "achieved_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            big = max(levels, key=lambda l: l["algorithmic_bytes"]) if levels else None
            out["costvolume_roofline"] = {"kernel": "costvolume_fwd_kernel (ff_pwc_costvolume_fwd_ex), 8 pairs", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "dominant_level": big, "levels": levels,
                                          "timing": "event pair recorded around each launch on its stream (includes the dispatch gap; the launches are 5-60 us)"}
    except Exception as e:          # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    torch.cuda.empty_cache()
    return out


def secondary_measurements(args, device):
    """The secondary BASELINE configs on the same GPU, after the headline measurement (N = 1 only, a few seconds each):
    the training step (configs[2] shape, one GPU), the FF-PWC forward (configs[3]) and configs[4] (544x960, 32 iterations,
    fp16 pyramid).  Reported next to the headline line, never mixed into it; a failure is reported as text and does not
    touch the headline."""
    import copy
    out = {}
    try:
        targs = copy.copy(args)
        targs.batch, targs.height, targs.width, targs.iters = 8, 384, 512, 12      # -> 368 x 496 crops (train_setup)
        step, h, w = train_setup(targs, 1, 0, 0, device)
        import gc
        gc.collect()    # (the models of the legs before this one are garbage by now: their packed-weight caches should not be walked by this one's steps)
        nw = 4          # (the caching allocator and the backward's zero arena settle over the first three steps)
        for _ in range(nw):
            step()
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        host_ms = (t1 - t0) / n * 1e3
        out["train_step"] = {"metric": f"training frame-pairs/sec FF-RAFT {h}x{w} iters=12 (fwd+MixLoss+bwd+clip+AdamW)",
                             "value": round(8 / dt, 2), "unit": "frame-pairs/s", "ms_per_step": round(dt * 1e3, 2), "steps": n, "warmup": nw,
                             "host_ms_per_step": round(host_ms, 2),      # time until the host has issued a step (it synchronises once per step, in the loss: losses.py:39-45)
                             "workload": "BASELINE configs[2] shape on ONE GPU: 8 pairs, MixLoss, no DDP", "finite_loss": bool(torch.isfinite(loss))}
        del step
    except Exception as e:          # noqa: BLE001 - reported, never fatal for the headline line
        out["train_step"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    out["ff_pwc_forward"] = pwc_measurements(device)
    out["config4_544x960_it32_fp16_pyramid"] = config4_measurements(device)
    out["headline_shape_16_pairs"] = batch16_measurement(args, device)
    out["hipgraph_replay"] = graph_replay_measurement(args, device)
    return out


def TiledPyramidBytes(h8, w8, half):
    """Bytes of one query's four tiled planes (ops.TiledPyramid / csrc/corr_layout.h)."""
    from focusflow_official_amd import ops
    return sum(ops.TiledPyramid.plane_elems(h8, w8, l, half) for l in range(4)) * (2 if half else 4)


def ops_precision():
    from focusflow_official_amd import ops
    return ops.conv_precision()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frame pairs per GPU per step (BASELINE config 2: 8)")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary configs (training step, FF-PWC forward) measured after the headline at N = 1")
    ap.add_argument("--pyramid", choices=["fp32", "fp16"], default="fp32",
                    help="storage type of the correlation pyramid: fp32 = the reference's arithmetic (headline); fp16 = "
                         "BASELINE configs[4] (540x960 padded to 544x960, iters 32: --height 544 --width 960 --iters 32 --batch 1)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (the default when --gpus > 1)")
    ap.add_argument("--eager", action="store_true", help="--gpus > 1: issue the steps eagerly instead of replaying a hipGraph")
    ap.add_argument("--skip-unused-upsample", action="store_true",
                    help="NOT the default / not the headline number: compute the mask head + convex up-sampling only for "
                         "the last iteration (the reference discards the other 11 in test_mode); outputs are bit-identical")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward = BASELINE configs[1] (the headline metric); train = configs[2]-shaped step "
                         "(forward + MixLoss + backward + clip + AdamW, DDP over RCCL when --gpus > 1)")
    ap.add_argument("--harness-selftest", action="store_true",
                    help="no model: a sleep() stands in for the step so the multi-rank harness (sharding, barrier, "
                         "max-over-ranks timing, single JSON line) can be tested on CPU with gloo")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (the N=1 command with a different --gpus): become the launcher; nothing above initialised HIP
        sys.exit(self_launch(args.gpus))
    if args.harness_selftest:
        return harness_selftest(args)

    refuse_lab_switches()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # N ranks share the host: each keeps its CPU-side torch work (fills, index arithmetic, the optimiser in train mode) to
    # its share of the cores instead of N pools of `cores` threads fighting the N issuing threads
    host_threads = max(1, host_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    torch.set_num_threads(host_threads)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # every rank adds one: the line then proves how many RCCL ranks really took part in the collective (1 without a group)
    ranks_seen = 1
    if world > 1:
        t = torch.ones(1, device=device)
        dist.all_reduce(t)
        ranks_seen = int(t.item())

    from focusflow_official_amd import FF_RAFT_FUSION, ops
    if args.mode == "train":
        return train_mode(args, world, rank, local_rank, device, ranks_seen)
    torch.manual_seed(1234)
    model = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=cfg()).to(device).eval()
    model.flow_net.skip_unused_upsample = bool(args.skip_unused_upsample)
    model.flow_net.corr_pyramid_dtype = args.pyramid
    # weak scaling: every rank owns args.batch independent pairs of the global batch
    lo, hi = shard_units(args.batch * world, world, rank)
    batch = synthetic_batch(hi - lo, args.height, args.width, 1234 + rank, device)

    def step():
        with torch.no_grad():
            return model(*batch, raft_iters=args.iters, test_mode=True)

    # N > 1 ranks on one host: the step is replayed from a captured hipGraph by default - an eager step costs the host 6 ms of a
    # 12 ms step, and N ranks' launch threads share the host (--eager switches back; N = 1 stays eager so that the roofline's
    # dispatch-bound events see the timed region itself)
    if world > 1 and not args.eager:
        args.graph = True
    if args.graph:
        from focusflow_official_amd.graph import GraphedForward
        graphed = GraphedForward(model, batch, raft_iters=args.iters)
        step = lambda: graphed(*batch)  # noqa: E731

    log(f"rank {rank}/{world}: model + {hi - lo} pairs on {torch.cuda.get_device_name(device)}")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    # HIP events bound to every lookup / corr-build dispatch of the timed region (ff_launch_timing_begin: the library hands
    # an event pair to hipExtLaunchKernelGGL on the stream it launches on - the kernels' own execution times)
    ops.launch_timing_begin(ops.TIME_LOOKUP, ops.TIME_CORR_BUILD)
    elapsed, out = timed_region(step, args.steps, world, torch.cuda.synchronize, device)
    log(f"{args.steps} timed steps in {elapsed:.3f} s")
    lk, vb = ops.launch_timing_end(ops.TIME_LOOKUP), ops.launch_timing_end(ops.TIME_CORR_BUILD)
    assert torch.isfinite(out[1]).all()
    ops.guard_check(sync=True)          # the always-on range guard looks at a forward when the next one starts: the last one here
    # host time to issue one step, MAX over ranks like the step time itself (outside the timed region)
    issue_ms, issue_total_ms = host_issue_time(step)
    if world > 1:
        t = torch.tensor([issue_ms, issue_total_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        issue_ms, issue_total_ms = round(t[0].item(), 3), round(t[1].item(), 3)

    if rank == 0:
        pairs = args.batch * world * args.steps
        q = (hi - lo) * (args.height // 8) * (args.width // 8)
        if args.graph:   # a replayed graph launches no kernels through the library: time one eager forward instead
            ops.launch_timing_begin(ops.TIME_LOOKUP, ops.TIME_CORR_BUILD)
            with torch.no_grad():
                model(*batch, raft_iters=args.iters, test_mode=True)
            lk, vb = ops.launch_timing_end(ops.TIME_LOOKUP), ops.launch_timing_end(ops.TIME_CORR_BUILD)
        # the same launches bracketed by event pairs recorded around them (one extra untimed step): adds the dispatch gaps
        ops.profile_begin("lookup")
        with torch.no_grad():
            model(*batch, raft_iters=args.iters, test_mode=True)
        bracketed = ops.profile_end()["lookup"]
        n_lookups = lk[0]
        per_launch_ms = lk[1] / max(1, n_lookups) * 1e-3
        per_q = LOOKUP_BYTES_PER_QUERY[args.pyramid]
        # queries per LAUNCH from the launches actually seen (one lookup per iteration covers the rank's whole batch)
        timed_steps = 1 if args.graph else args.steps
        q = int(round(q * args.iters * timed_steps / max(1, n_lookups)))
        achieved = per_q * q / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        c2 = (args.height, args.width, args.iters, args.batch) == (384, 512, 12, 8) and args.pyramid == "fp32"
        line = {
            "metric": f"frame-pairs/sec FF-RAFT {args.height}x{args.width} iters={args.iters}", "value": round(pairs / elapsed, 3),
            "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPES[ops.conv_precision()], "data": "synthetic",
            "config": {"workload": f"FF-RAFT forward (test_mode), {args.batch} pairs/GPU {args.height}x{args.width}, "
                                   f"iters={args.iters}, {args.pyramid} correlation pyramid, random-init weights, ORB-like masks "
                                   f"({'BASELINE configs[1]' if c2 else 'BASELINE configs[4]' if args.pyramid == 'fp16' else 'non-headline shape'})",
                       "pairs_per_gpu": args.batch, "corr_pyramid": args.pyramid, "conv_precision": ops.conv_precision(), "hipgraph": bool(args.graph), "skip_unused_upsample": bool(args.skip_unused_upsample),
                       "parallelism": f"dp{world} (independent shards, no collective)",
                       # feature switches (A/B runs: FF_SPLIT_ACT=0, FF_GRU_PASS=0 ...) present in the environment; [] for the default line
                       "env_switches": sorted(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FF_"))},
            "roofline": {"kernel": f"lookup_dma_kernel<{'fp16' if args.pyramid == 'fp16' else 'fp32'}> (ff_corr_lookup_tiled_fwd)",
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(q, args.pyramid)[0],
                         "traffic_source": pmc_traffic(q, args.pyramid)[1],
                         "launches": n_lookups, "avg_launch_us": round(per_launch_ms * 1e3, 2),
                         "min_launch_us": round(lk[2], 2), "max_launch_us": round(lk[3], 2),
                         "timing": "HIP events bound to each dispatch of the timed region (hipExtLaunchKernelGGL start/stop events on the launching stream)",
                         "avg_launch_us_bracketed": round(sum(bracketed) / max(1, len(bracketed)) * 1e3, 2),
                         "algorithmic_bytes_per_query": per_q, "algorithmic_bytes_per_launch": per_q * q},
        }
        if not args.graph:
            try:
                def one_forward():
                    with torch.no_grad():
                        model(*batch, raft_iters=args.iters, test_mode=True)
                mo = memory_only_companion(per_q * q, achieved, device, one_forward)
                traffic = line["roofline"]["traffic"]
                if traffic and mo["bytes_per_launch"]:
                    # the same kernel's time scaled to the HBM bytes the counters see for one lookup launch (128-byte lines
                    # around 64-80-byte window rows: 1.245 x the algorithmic bytes) - a derived figure, not a measurement
                    mo["scaled_to_counted_traffic_us"] = round(mo["avg_launch_us"] * traffic / mo["bytes_per_launch"], 2)
                    mo["lookup_over_scaled"] = round(mo["scaled_to_counted_traffic_us"] / (per_launch_ms * 1e3), 4) if per_launch_ms > 0 else None
                for key in ("in_pipeline", "in_pipeline_before_lookup", "in_pipeline_behind_flow_head"):
                    ip = mo.get(key)
                    if ip and ip.get("avg_launch_us", 0) > 0:
                        gbs = mo["bytes_per_launch"] / (ip["avg_launch_us"] * 1e-6) / 1e9
                        ip.update(achieved=round(gbs, 1), unit="GB/s", frac_of_peak=round(gbs / HBM_PEAK_GBS, 4), lookup_over_this=round(achieved / gbs, 4))
                        if traffic:
                            ip["scaled_to_counted_traffic_us"] = round(ip["avg_launch_us"] * traffic / mo["bytes_per_launch"], 2)
                            ip["lookup_over_scaled"] = round(ip["scaled_to_counted_traffic_us"] / (per_launch_ms * 1e3), 4) if per_launch_ms > 0 else None
                line["roofline"]["memory_only_kernel"] = mo
            except Exception as e:          # noqa: BLE001 - a measurement aid must not cost the bench line
                line["roofline"]["memory_only_kernel"] = {"error": f"{type(e).__name__}: {e}"}
        # corr-volume build (BASELINE.md "also reported"): dense HWxC x CxHW contraction on the matrix pipe
        q1 = (args.height // 8) * (args.width // 8)
        vol_flop = 2.0 * (hi - lo) * q1 * q1 * 256
        vol_avg_ms = vb[1] / max(1, vb[0]) * 1e-3
        terms = {"f16x3": 3, "f16": 1, "fp32": 1}[ops.conv_precision()]
        peak = MFMA_PEAK_TFLOPS["fp32" if ops.conv_precision() == "fp32" else "f16"]
        issued = vol_flop * terms / (vol_avg_ms * 1e-3) / 1e12 if vol_avg_ms > 0 else 0.0
        pyr_bytes = (hi - lo) * q1 * TiledPyramidBytes(args.height // 8, args.width // 8, args.pyramid == "fp16")
        line["roofline_corr_build"] = {
            "kernel": "corr_build_kernel (ff_corr_build: f16x3 volume + 3 pooled levels + tiled store, one launch)" if ops.conv_precision() == "f16x3"
                      else "conv kernel, groups=B (ff_conv2d_fwd) + pooling pass + retile", "bound": "mfma",
            "achieved": round(issued, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(issued / peak, 4),
            "note": f"{terms} MFMA term(s) per fp32-accurate product; useful rate {(vol_flop / (vol_avg_ms * 1e-3) / 1e12 if vol_avg_ms > 0 else 0.0):.1f} TFLOP/s; "
                    f"the launch also writes the whole pyramid once ({pyr_bytes / 1e6:.1f} MB)",
            "write_gbs": round(pyr_bytes / (vol_avg_ms * 1e-3) / 1e9, 1) if vol_avg_ms > 0 else 0.0, "write_bytes": pyr_bytes,
            "launches": vb[0], "avg_launch_us": round(vol_avg_ms * 1e3, 1)}
        # all convolutions of one extra (untimed) step, each launch bracketed by HIP events: where 80 % of the step goes
        ops.policy.single_stream = True          # bracketed launches must not overlap: this extra step runs on one stream
        try:
            ops.profile_begin("conv")
            with torch.no_grad():
                model(*batch, raft_iters=args.iters, test_mode=True)
            conv_ms = ops.profile_end()["conv"]
        finally:
            ops.policy.single_stream = False
        notes = ops.profile_notes("conv")
        useful = sum(n[0] for n in notes)
        issued_fl = sum(n[0] * (3 if n[1] == 1 else 1) for n in notes if n[1] != 0)
        mfma_ms = sum(t for t, n in zip(conv_ms, notes) if n[1] != 0)
        tot_ms = sum(conv_ms)
        line["roofline_conv"] = {
            "kernel": "every ff_conv2d_fwd, ff_gru_pass and ff_fusion_pair_fwd launch of one step (conv_patch / conv_split / conv_small / conv_dma / gru_pass / fusion_pair kernels)", "bound": "mfma",
            "achieved": round(issued_fl / (mfma_ms * 1e-3) / 1e12, 1) if mfma_ms > 0 else 0.0, "peak": MFMA_PEAK_TFLOPS["f16"],
            "unit": "TFLOP/s", "frac": round(issued_fl / (mfma_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["f16"], 4) if mfma_ms > 0 else 0.0,
            "note": f"f16 MFMA FLOP issued (3 per fp32-accurate product) over the summed launch durations; useful "
                    f"{useful / (tot_ms * 1e-3) / 1e12:.1f} TFLOP/s; the f16 pipe sustains ~1600 TFLOP/s on dense data "
                    f"(tools/proto/mfma_peak.hip)",
            "launches": len(conv_ms), "sum_launch_ms": round(tot_ms, 3), "useful_gflop_per_step": round(useful / 1e9, 1)}
        line["rccl_ranks_seen"] = ranks_seen
        line["host"] = {"host_issue_ms": issue_ms, "step_ms_same_sample": issue_total_ms, "threads_per_rank": host_threads, "cores": host_cores(),
                        "note": "median host time to issue one step with an empty queue (no sync inside) and the wall time of the same step, "
                                "MAX over ranks; a rank is host-bound where the two meet"}
        checker = {}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.height, args.width, args.iters, keep=checker)
        if world == 1 and c2 and not args.no_secondary and not args.graph:
            step = model = batch = None          # noqa: F841 - release the forward model before the secondary configs
            torch.cuda.empty_cache()
            line["secondary"] = parity_and_reduced_precision(args, device, checker)
            line["secondary"].update(secondary_measurements(args, device))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
