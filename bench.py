#!/usr/bin/env python3
"""bench.py -- depth maps/sec of the MI355X-native MVSNet depth path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic depth-map problem (features already
resident in HBM -> depth + confidence), i.e. reference models/mvsnet.py:145-218.  Every rank
processes its own independent maps (reference views shard embarrassingly); the
only collective is one RCCL all-gather of the [K,2,h,w] results at the end (inside the timed
region).  Rank 0 prints ONE JSON line.

The timed region is K x one mvs_depth_infer call per map (what the drop-in's forward enqueues),
nothing else.  Per-kernel durations -- and the roofline figure of the dominant kernel -- are
measured live in the same process right after it: the same kernels on the same inputs issued
through the per-stage C-ABI calls with a HIP event (torch event on the launch stream) after each.
(An event between every pair of kernels costs ~3 us, 4 % of a map, so that pass is not `value`.)

`python bench.py --gpus N` with N > 1 and no torchrun environment launches itself: the parent,
before any GPU call, starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child process and relays rank 0's JSON line (the reference's counterpart: nn.DataParallel,
eval.py:309).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

# torch and the package are imported inside main(), after the self-launch decision: the parent of a
# self-launched multi-GPU run never loads the HIP runtime at all.
CONFIG_NAMES = ("cfg1", "cfg2", "cfg3", "cfg5")

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (= vector) peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / fp16 MFMA, dense (~2.5 PF; not the 2:1-sparsity figure)
FUSED_STAGES = ("conv11_prob",)   # kernels that replace two d3 stages: never part of the d3 totals


def mfma_peak_tflops(storage, mfma16=True):
    """MFMA peak of the arithmetic the conv layers run in: fp32 MFMA for fp32 storage (and for 16-bit storage with
    MVS_MFMA16=0), the 16-bit matrix cores otherwise."""
    return MFMA_F32_PEAK_TFLOPS if (storage == "f32" or not mfma16) else MFMA_16BIT_PEAK_TFLOPS


def path_totals(costs, mfma_peak):
    """SURVEY.md §8 d3 whole-path totals: layer by layer, no fusion credited (conv11 and prob count as two
    stages whichever kernels ran) -> (bytes, flops, stage-wise roofline seconds)."""
    ref = {k: v for k, v in costs.items() if k not in FUSED_STAGES}
    b = sum(c["bytes"] for c in ref.values())
    f = sum(c["flops"] for c in ref.values())
    floor_s = sum(max(c["bytes"] / (HBM_PEAK_GBPS * 1e9), c["flops"] / (mfma_peak * 1e12)) for c in ref.values())
    return b, f, floor_s

LAYERS = [  # (name, cin, cout, level_in, level_out, kind)
    ("conv0", 32, 8, 0, 0, "conv"), ("conv1", 8, 16, 0, 1, "conv"), ("conv2", 16, 16, 1, 1, "conv"),
    ("conv3", 16, 32, 1, 2, "conv"), ("conv4", 32, 32, 2, 2, "conv"), ("conv5", 32, 64, 2, 3, "conv"),
    ("conv6", 64, 64, 3, 3, "conv"), ("conv7", 64, 32, 3, 2, "deconv"), ("conv9", 32, 16, 2, 1, "deconv"),
    ("conv11", 16, 8, 1, 0, "deconv"), ("prob", 8, 1, 0, 0, "conv"),
]


def stage_costs(N, D, h, w, es=4):
    """Algorithmic bytes / FLOPs per stage per map (SURVEY.md §8 d3 definitions)."""
    V0 = D * h * w
    # every tensor at the storage dtype (`es` bytes), as d3 defines it -- the 16-bit modes of this build keep the
    # logits in fp32, which is more than the algorithmic bytes and is not credited
    costs = {"warp_variance": dict(bytes=N * 32 * h * w * es + 32 * V0 * es, flops=0.0)}
    for name, ci, co, li, lo, kind in LAYERS:
        vin, vout = V0 >> (3 * li), V0 >> (3 * lo)
        skip = co * vout * es if kind == "deconv" else 0
        flops = 2.0 * 27 * ci * co * (vin if kind == "deconv" else vout)
        costs[name] = dict(bytes=ci * vin * es + co * vout * es + skip, flops=flops)
    # conv11 (+ conv0 skip) and prob in one kernel: the 8-channel tensor between them never reaches HBM
    costs["conv11_prob"] = dict(bytes=16 * (V0 >> 3) * es + 8 * V0 * es + V0 * es,
                                flops=costs["conv11"]["flops"] + costs["prob"]["flops"])
    costs["softargmin"] = dict(bytes=V0 * es + 2 * h * w * 4, flops=0.0)
    return costs


def live_traffic(kernel_substr, what, reps=3, timeout=150):
    """HBM bytes per launch of one kernel, measured NOW: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE;
    each with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over `tools/prof_stage.py <what> <reps>` -- the
    same kernel on the same cfg2 inputs -- run as child processes of this one.  gfx950 correction: FETCH_SIZE under-
    reports wide streaming reads by 2x.  Returns None when rocprofv3 is missing, fails or times out."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    got = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                out_dir = os.path.join(td, ctr)
                cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "--",
                       sys.executable, os.path.join(REPO, "tools", "prof_stage.py"), what, str(reps)]
                proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                        stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    rc = proc.wait(timeout=timeout)
                except subprocess.TimeoutExpired:
                    os.killpg(proc.pid, signal.SIGKILL)     # exactly the process group started above
                    proc.wait()
                    return None
                if rc != 0:
                    return None
                vals = []
                for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                    with open(path) as f:
                        for row in csv.DictReader(f):
                            if row.get("Counter_Name") == ctr and kernel_substr in (row.get("Kernel_Name") or ""):
                                vals.append(float(row["Counter_Value"]))
                if not vals:
                    return None
                got[ctr] = sum(vals) / len(vals)
    except (OSError, ValueError, KeyError):
        return None
    return int((2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=CONFIG_NAMES)
    ap.add_argument("--dtype", default=None, choices=["f32", "f16", "bf16"],
                    help="storage dtype of the private volumes (arithmetic is always fp32); default: "
                         "f32 for cfg1/cfg2, bf16 for cfg3, f16 for cfg5 as BASELINE.json names them")
    ap.add_argument("--prewarm-ms", type=int, default=300,
                    help="untimed milliseconds of the same workload between the first timed pass (right after the "
                         "--warmup steps, reported as `first_pass`) and the pass reported as `value`; 0 = one pass only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (from images) figure")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the committed profiles/rNN_traffic.json instead of measuring it now "
                         "(two rocprofv3 --pmc child runs of ~10 s each, after everything else)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams to round-robin independent maps over (each has its own workspace); two maps "
                         "in flight fill the launch gaps and the tails of the small U-Net layers (+6 %% over 1)")
    ap.add_argument("--staged-steps", type=int, default=10,
                    help="maps of the per-kernel pass after the timed region (per-stage C-ABI calls with a "
                         "HIP event after each kernel); 0 = skip it (no `stages` / `roofline` objects)")
    ap.add_argument("--staged-timed", action="store_true",
                    help="diagnostic: the TIMED steps go through the per-stage calls with events too")
    return ap.parse_args(argv)


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(args, argv, port=None):
    """The torchrun command line `python bench.py --gpus N` turns into (one rank per GPU, RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()),
            os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv) -> int:
    """Parent of a multi-GPU run started without torchrun: never touches the GPU (on this pool a
    process that has initialised HIP must not exec; a child started before that is fine), starts
    the ranks as a child process and relays rank 0's JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it here
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.run(self_launch_command(args, argv), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if proc.returncode or lines else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    global torch, dist, _lib, sharding, synthetic
    import torch
    import torch.distributed as dist
    from scene_3dreconstruction_mvsnet_amd import _lib, sharding, synthetic
    # one process per GPU: keep this rank's host threads on the cores next to its GPU (before any GPU call)
    pinned = sharding.pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # MVS_BENCH_REHEARSAL=1: run every rank on cuda:0 with the gloo backend -- a way to exercise
    # the N>1 code path on a one-GPU box (numbers are meaningless; the driver never sets it)
    rehearsal = os.environ.get("MVS_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = synthetic.CONFIGS[args.config]
    N, D, h, w = cfg["nviews"], cfg["D"], cfg["H"] // 4, cfg["W"] // 4
    K, Wm = args.steps, args.warmup
    storage = args.dtype or {"cfg3": "bf16", "cfg5": "f16"}.get(args.config, "f32")
    dt = _lib.dtype_code(storage)
    es = 4 if storage == "f32" else 2
    _lib.load()

    # ---- synthetic problem (per-rank seed: every rank owns different ref views) -------------
    feats_np = synthetic.random_features(N, 32, h, w, seed=rank)
    proj_np = synthetic.cameras(N, h, w)
    dv_np = synthetic.depth_values(D, interval_scale=cfg["interval_scale"])
    sd = synthetic.random_costreg_state(seed=0)
    feats = torch.from_numpy(feats_np).to(dev)
    proj = torch.from_numpy(proj_np).to(dev)
    dv = torch.from_numpy(dv_np).to(dev)
    blob = _lib.pack_weights(sd).to(dev)
    # fraction of the (pixel, depth, source view) sampling points that land inside the source image
    # (out-of-image taps are cheaper, so the figure travels with every timing)
    in_image_frac = round(synthetic.in_image_fraction(proj_np, dv_np, h, w), 4)
    S = max(1, args.streams)
    wss = [_lib.alloc_workspace(N, 32, D, h, w, dev, dt) for _ in range(S)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    out = torch.zeros((K, 2, h, w), dtype=torch.float32, device=dev)  # depth, conf per step

    # the library ends the path in ONE kernel for conv11 + prob (csrc/conv11_prob.hip) unless MVS_FUSE_PROB=0 (16-bit
    # storage: on the 16-bit matrix cores, so not with MVS_MFMA16=0)
    fused_tail = os.environ.get("MVS_FUSE_PROB") != "0" and (storage == "f32" or os.environ.get("MVS_MFMA16") != "0")
    layer_names = [l[0] for l in LAYERS]
    if fused_tail:
        layer_names = layer_names[:9] + ["conv11_prob"]
    stage_names = ["relative_proj", "warp_variance"] + layer_names + ["softargmin"]
    n_ev = len(stage_names) + 1
    KS = max(0, args.staged_steps)      # maps of the per-kernel event pass after the timed region
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(max(K, KS))]

    # Pre-allocated per-stream buffers and pre-bound C calls keep the host ahead of the GPU in the
    # staged mode (no torch allocations or shape logic inside the timed loop).
    lib = _lib.load()
    V0 = D * h * w

    def lvl(c, l):
        return torch.empty((c // 8, D >> l, h >> l, w >> l, 8), dtype=_lib.TORCH_DTYPES[dt], device=dev)

    bufs = []
    for si in range(S):
        bufs.append(dict(rt=torch.empty((max(N - 1, 1), 12), dtype=torch.float32, device=dev),
                         var=lvl(32, 0),
                         act=[lvl(8, 0), lvl(16, 1), lvl(16, 1), lvl(32, 2), lvl(32, 2), lvl(64, 3),
                              lvl(64, 3), lvl(32, 2), lvl(16, 1), lvl(8, 0)],
                         cost=torch.empty((D, h, w), dtype=torch.float32, device=dev)))
    skips = {7: 4, 8: 2, 9: 0}

    def step_staged(k, ev=None):
        si = k % S
        ws, B = wss[si], bufs[si]
        st = _lib._stream(dev)
        rec = (lambda i: ev[i].record()) if ev is not None else (lambda i: None)
        rec(0)
        _lib.check(lib.mvs_relative_proj(proj.data_ptr(), B["rt"].data_ptr(), N, st))
        rec(1)
        ei = 2
        _lib.check(lib.mvs_warp_variance(feats.data_ptr(), B["rt"].data_ptr(), dv.data_ptr(),
                                         B["var"].data_ptr(), ws.data_ptr(), ws.numel(), N, 32, D,
                                         h, w, dt, st))
        x = B["var"]
        rec(ei)
        for li in range(11):
            if fused_tail and li == 9:
                _lib.check(lib.mvs_conv11_prob(x.data_ptr(), B["act"][0].data_ptr(), B["cost"].data_ptr(),
                                               blob.data_ptr(), D >> 1, h >> 1, w >> 1, dt, st))
                x = B["cost"]
                ei += 1
                rec(ei)
                break
            yb = B["cost"] if li == 10 else B["act"][li]
            sk = B["act"][skips[li]].data_ptr() if li in skips else 0
            lvin = LAYERS[li][3]
            _lib.check(lib.mvs_conv_layer(li, x.data_ptr(), sk, yb.data_ptr(), blob.data_ptr(),
                                          D >> lvin, h >> lvin, w >> lvin, dt, st))
            x = yb
            ei += 1
            rec(ei)
        _lib.check(lib.mvs_softargmin_conf(x.data_ptr(), dv.data_ptr(), out[k, 0].data_ptr(),
                                           out[k, 1].data_ptr(), D, h, w, st))
        rec(ei + 1)

    def step_fused(k, ev=None):
        ws = wss[k % S]
        _lib.depth_infer(feats, proj, dv, blob, ws, out[k, 0], out[k, 1], dtype=dt)

    # Timed steps: one mvs_depth_infer call per map (what the drop-in's forward enqueues), maps
    # round-robin over the S streams (each stream has its own workspace).
    step_one = step_staged if args.staged_timed else step_fused

    def step(k, ev=None):
        if S == 1:
            return step_one(k, ev)
        with torch.cuda.stream(streams[k % S]):
            step_one(k, ev)

    for i in range(Wm):
        step(i % K)
    torch.cuda.synchronize()
    if world > 1:
        # untimed warm-up of the result gather: RCCL builds its rings / buffers on the first
        # collective of a given kind, which must not land inside the timed region
        sharding.gather_maps(out, world * K, rank, world)
        torch.cuda.synchronize()
        dist.barrier()
    per_rank = []   # seconds of the last timed pass on each rank's own clock

    def timed_pass():
        """EXACTLY K steps between barrier + synchronize on both sides -> seconds (max over ranks)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            step(k)
        for st in streams[1:]:
            streams[0].wait_stream(st)
        if world > 1:
            # the final gather (RCCL over xGMI): rank r owns units r::world of the world*K maps
            gathered = sharding.gather_maps(out, world * K, rank, world)
            assert gathered.shape[0] == world * K
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        own = None
        if world > 1:
            # every rank's own clock around the same region (a slow rank shows up), then the max over ranks
            own = torch.zeros(world, dtype=torch.float64, device="cpu" if rehearsal else dev)
            own[rank] = dt_s
            dist.all_reduce(own, op=dist.ReduceOp.SUM)
            own = [float(x) for x in own.tolist()]
            dt_s = max(own)
        per_rank.clear()
        per_rank.extend(own or [dt_s])
        return dt_s

    # The K steps are timed twice.  `first_pass`: right after the W warm-up steps.  W = 5 steps are 5 ms of
    # work, and an MI355X coming from idle needs ~0.1-0.3 s under load before its clocks have settled (measured:
    # K=20 after W=5: 1,075 maps/s; K=20 after W=300: 1,186; K=1000: 1,205 -- profiles/r02_bench_ramp.txt), so
    # that figure is a cold-start figure.  `value`: the same K steps again after `--prewarm-ms` (default 300)
    # of the same workload, untimed -- the sustained rate the metric asks for.  Both are in the JSON line.
    first_elapsed = timed_pass()
    elapsed = first_elapsed
    prewarm_ms = max(0, args.prewarm_ms)
    effective_warmup = Wm      # untimed steps in front of the pass reported as `value`
    if prewarm_ms:
        tp = time.perf_counter()
        effective_warmup += K  # the first pass itself
        while (time.perf_counter() - tp) * 1e3 < prewarm_ms:
            for k in range(K):
                step(k)
            effective_warmup += K
            torch.cuda.synchronize()
        elapsed = timed_pass()

    maps_per_s = world * K / elapsed
    ms_per_step = elapsed / K * 1e3

    # ---- per-kernel durations: the same kernels on the same inputs through the per-stage C-ABI
    # calls, a HIP event (on the launch stream) after each, one stream, right after the timed region
    costs = stage_costs(N, D, h, w, es)
    mfma_peak = mfma_peak_tflops(storage, os.environ.get("MVS_MFMA16") != "0")
    stages = {}
    staged_steps = list(range(KS))
    if staged_steps:
        step_staged(0)
        torch.cuda.synchronize()
        for k in staged_steps:
            step_staged(k % K, events[k])
        torch.cuda.synchronize()
        for si, name in enumerate(stage_names):
            ms = float(np.mean([events[k][si].elapsed_time(events[k][si + 1]) for k in staged_steps]))
            ent = {"ms": round(ms, 4)}
            c = costs.get(name)
            if c and ms > 0:
                ent["GBps"] = round(c["bytes"] / ms / 1e6, 1)
                if c["flops"]:
                    ent["TFLOPs"] = round(c["flops"] / ms / 1e9, 2)
                # fraction of this stage's own roofline: max(HBM time, MFMA time at the arithmetic dtype's peak) / measured
                floor_ms = max(c["bytes"] / (HBM_PEAK_GBPS * 1e6), c["flops"] / (mfma_peak * 1e9))
                ent["bound"] = "mfma" if c["flops"] / (mfma_peak * 1e9) > c["bytes"] / (HBM_PEAK_GBPS * 1e6) else "hbm"
                ent["frac"] = round(floor_ms / ms, 3)
            stages[name] = ent
    roofline = None
    # which conv0 kernel the library picks (csrc/conv3d_direct.hip): F(4,3) unless told otherwise
    wino = "0" if (os.environ.get("MVS_CONV0_WINO") == "0" or D % 4) else "4"
    if stages:
        dom = max((n for n in stages if n in costs), key=lambda n: stages[n]["ms"])
        c, ms = costs[dom], stages[dom]["ms"]
        t_hbm = c["bytes"] / (HBM_PEAK_GBPS * 1e9)
        t_mfma = c["flops"] / (mfma_peak * 1e12)
        if t_mfma > t_hbm:
            ach = c["flops"] / ms / 1e9
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3),
                        "peak": mfma_peak, "unit": "TFLOP/s",
                        "frac": round(ach / mfma_peak, 4), "traffic": None,
                        "avg_launch_ms": ms, "algorithmic_flops": c["flops"],
                        "algorithmic_bytes": c["bytes"]}
            if dom == "conv0" and storage == "f32" and wino == "4":
                num, den, form = 1, 2, "F(4,3)"
                roofline["note"] = (f"conv0 runs Winograd {form} along z on the fp32 4x4x1 MFMA: it issues {num}/{den} "
                                    "of the algorithmic multiply-adds; `achieved` is the ALGORITHMIC flops / time as "
                                    f"SURVEY 8 d3 defines it (it can exceed the MFMA peak), the executed-MFMA rate is "
                                    f"{num}/{den} of that")
                roofline["executed_flops"] = c["flops"] * num // den
        else:
            ach = c["bytes"] / ms / 1e6
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1),
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                        "traffic": None, "avg_launch_ms": ms, "algorithmic_bytes": c["bytes"]}

    # HBM traffic of the dominant kernel: the committed PMC profile (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    # passes, gfx950 FETCH x2 correction) first; replaced by a live measurement at the very end (below) when possible
    if roofline is not None and args.config == "cfg2":
        try:
            import glob
            newest = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")))[-1]
            with open(newest) as f:
                prof = json.load(f)["kernels"]
            want = {"conv0": "conv0_w43_mfma_kernel<0>" if wino == "4" else "conv0_4x4_mfma_kernel<0>",
                    "warp_variance": "warp_variance_tc2_kernel<0, 0, 4, 4, 0, 1>"}.get(roofline["kernel"], "?")
            ent = next((v for k, v in prof.items() if k.endswith(want)), None)
            if ent:
                roofline["traffic"] = ent["hbm_bytes_fetch_x2"]
                roofline["traffic_source"] = (f"profiles/{os.path.basename(newest)} (rocprofv3 --pmc FETCH_SIZE / "
                                              "WRITE_SIZE passes of the same kernels, per launch; FETCH x2 gfx950 correction)")
        except (OSError, KeyError, ValueError):
            pass

    # whole-path totals follow SURVEY.md §8 d3 (layer-by-layer, no fusion credited), independent of
    # which kernels ran
    path_bytes, path_flops, stagewise_floor_s = path_totals(costs, mfma_peak)

    # ---- CPU baseline (rank 0, N=1): the oracle on one full map of the same workload ---------
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        tc = time.perf_counter()
        depth_o, conf_o = orc.depth_infer(feats_np, proj_np, dv_np, sd, storage=storage)
        tc = time.perf_counter() - tc
        cpu_baseline = {"value": round(1.0 / tc, 5), "unit": "depth maps/s", "cores": orc.num_threads(),
                        "kind": "port",
                        "sample": f"1 full {args.config} map (N={N}, {h}x{w}x{D}) through oracle/ "
                                  f"(C + OpenMP restatement), {tc:.1f} s"}
        got = out[K - 1, 0].cpu().numpy()
        parity = float(np.abs(got - depth_o).mean() / np.abs(depth_o).mean())

    # ---- end-to-end figure, outside the timed region, never `value`: the drop-in
    # MVSNet.forward from images = FeatureNet (HIP) + the path, images resident or copied per step
    end_to_end = None
    if rank == 0 and world == 1 and not args.no_e2e:
        from scene_3dreconstruction_mvsnet_amd import MVSNet
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):   # the drop-in prints an init banner like the reference
            model = MVSNet(refine=False)
        synthetic.randomize_bn_(model, seed=0)
        model = model.to(dev).eval()
        model.storage_dtype = storage
        imgs_np, proj_i, dv_i = synthetic.make_inputs(N, cfg["H"], cfg["W"], D, seed=0,
                                                      interval_scale=cfg["interval_scale"])
        imgs_h = torch.from_numpy(imgs_np).pin_memory()
        proj_i, dv_i = torch.from_numpy(proj_i).to(dev), torch.from_numpy(dv_i).to(dev)
        end_to_end = {"unit": "depth maps/s", "includes": "FeatureNet (HIP) + path; h2d adds the "
                      "pinned-host -> HBM copy of the N images on the same stream; K maps after "
                      f"{prewarm_ms} ms of the same calls (untimed)"}
        # like the path-only figure: forwards round-robin over the S streams (the module keeps one workspace per
        # stream); in h2d mode every forward first copies its own images on its stream
        def e2e_forward(i, mode):
            with torch.cuda.stream(streams[i % S]):
                x_d = imgs_h.to(dev, non_blocking=True) if mode == "h2d" else imgs_d
                model(x_d, proj_i, dv_i)

        for mode in ("resident", "h2d"):
            imgs_d = imgs_h.to(dev)
            for i in range(3 * S):
                e2e_forward(i, mode)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            while (time.perf_counter() - tp) * 1e3 < prewarm_ms:   # the CPU baseline above left the device idle
                for i in range(10):
                    e2e_forward(i, mode)
                torch.cuda.synchronize()
            te = time.perf_counter()
            for i in range(K):
                e2e_forward(i, mode)
            torch.cuda.synchronize()
            end_to_end[mode] = round(K / (time.perf_counter() - te), 2)
        end_to_end["streams"] = S

    # live PMC measurement of the dominant kernel's HBM traffic (rank 0 of a one-GPU cfg2 run; after every timing)
    if (rank == 0 and world == 1 and roofline is not None and args.config == "cfg2" and storage == "f32"
            and not args.no_live_traffic and roofline["kernel"] in ("conv0", "warp_variance")):
        torch.cuda.synchronize()
        sub = {"conv0": "conv0_w43_mfma_kernel" if wino == "4" else "conv0_4x4_mfma_kernel",
               "warp_variance": "warp_variance_tc2_kernel"}[roofline["kernel"]]
        t_live = time.perf_counter()
        live = live_traffic(sub, "conv0" if roofline["kernel"] == "conv0" else "warp")
        if live:
            roofline["traffic"] = live
            roofline["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child "
                                          "passes, --kernel-trace only) over tools/prof_stage.py, per launch; FETCH x2 gfx950 "
                                          f"correction; {time.perf_counter() - t_live:.0f} s")

    if rank == 0:
        line = {
            "metric": "depth maps/sec at N=5 views, 640x512, D=192; achieved HBM GB/s"
                      if args.config == "cfg2" else f"depth maps/sec ({args.config})",
            "value": round(maps_per_s, 3), "unit": "depth maps/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "effective_warmup_steps": effective_warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if storage == "f32" else (
                f"{storage} storage, f32 MFMA arithmetic" if os.environ.get("MVS_MFMA16") == "0"
                else f"{storage} storage and MFMA operands, f32 accumulation"),
            "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N} views, {cfg['H']}x{cfg['W']} image -> "
                                   f"{h}x{w} features, D={D}, C=32, {storage} volumes; path-only (features "
                                   "resident in HBM -> depth+confidence)",
                       "maps_per_rank": K, "sharding": "independent ref views per rank, one RCCL "
                                                       "all-gather of results at the end",
                       "call": ("staged C-ABI calls with HIP events" if step_one is step_staged else
                                "one mvs_depth_infer call per map") +
                               (f"; per-kernel durations from {KS} further maps through the per-stage C-ABI "
                                "calls with HIP events, after the timed region" if KS else ""),
                       "in_image_frac": in_image_frac,
                       "prewarm_ms": prewarm_ms,
                       "streams": S},
            "first_pass": {"value": round(world * K / first_elapsed, 3), "ms_per_step": round(first_elapsed / K * 1e3, 4),
                           "note": f"the same {K} steps timed right after the {Wm} warm-up steps, before the device "
                                   f"clocks had settled; `value` is the same pass repeated after {prewarm_ms} ms of "
                                   "the same workload (untimed)"},
            "hbm_GBps_algorithmic": round(path_bytes * maps_per_s / 1e9, 1),
            "hbm_frac_of_peak": round(path_bytes * maps_per_s / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "path": {"algorithmic_bytes": path_bytes, "algorithmic_flops": path_flops,
                     "stagewise_roofline_ms": round(stagewise_floor_s * 1e3, 4),
                     "frac_of_stagewise_roofline": round(stagewise_floor_s / (elapsed / K), 4)},
            "per_rank": {"maps_per_s": [round(K / t, 2) for t in per_rank], "host_cores": len(pinned) or None,
                         "note": "each rank's K maps / its own clock around the timed region (barriers and the "
                                 "gather included); `value` = world*K / the slowest"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "end_to_end": end_to_end, "stages": stages,
            "parity_rel_l1_vs_oracle": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
