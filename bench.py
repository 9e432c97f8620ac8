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


def executed_costs(costs, storage, N, D, h, w, env=None):
    """What THIS BUILD executes / moves per stage, next to the algorithmic d3 figures of `stage_costs`:
    * fp32 storage: conv0 runs Winograd F(4,3) along z (6 transformed planes x 9 taps per 4 output planes = 1/2 of the
      27 multiply-adds per output), conv2 / conv4 F(2,3) along z on the 16x16x4 MFMA (4 planes x 10 tap slots per 2
      output planes = 20/27) -- unless MVS_CONV0_WINO=0 / MVS_CONV_WINO=0 (csrc/conv3d_direct.hip:460-480);
    * 16-bit storage: the logits stay fp32 (conv11+prob writes, softargmin reads 4 bytes per voxel).
    A roofline fraction computed from these never exceeds 1."""
    env = os.environ if env is None else env
    ex = {k: dict(v) for k, v in costs.items()}
    V0 = D * h * w
    if storage == "f32":
        if env.get("MVS_FORCE_DIRECT") != "1":
            if env.get("MVS_CONV0_WINO") != "0" and D % 4 == 0 and V0 * 32 < (1 << 31):
                ex["conv0"]["flops"] = costs["conv0"]["flops"] * 0.5
                if conv0_split_enabled(env):
                    # split operands (csrc/conv0_split.hip): six bf16 cross products per fp32 product, Toeplitz-pair
                    # form (4 x-taps issued per 3 real ones), on the bf16 matrix cores: 1/2 x 6 x 4/3 = 4x the
                    # algorithmic multiply-adds, priced against the bf16 MFMA peak
                    ex["conv0"]["flops"] = costs["conv0"]["flops"] * 4.0
                    ex["conv0"]["mfma_peak"] = MFMA_16BIT_PEAK_TFLOPS
                    ex["conv0"]["arith"] = "3xbf16 split operands, six cross products, fp32 accumulate"
            if env.get("MVS_SPLIT_LAYERS") != "0":
                # conv2 .. conv4: split operands on the bf16 matrix cores, six cross products per product, 27 taps padded
                # to 28 (csrc/conv3d_mfma16.hip convgs)
                for n in ("conv2", "conv3", "conv4"):
                    ex[n]["flops"] = costs[n]["flops"] * 6.0 * 28.0 / 27.0
                    ex[n]["mfma_peak"] = MFMA_16BIT_PEAK_TFLOPS
                    ex[n]["arith"] = "3xbf16 split operands, six cross products, fp32 accumulate"
            elif env.get("MVS_CONV_WINO") != "0":
                for n in ("conv2", "conv4"):
                    ex[n]["flops"] = costs[n]["flops"] * 20.0 / 27.0
            sd = int(env.get("MVS_SPLIT_DECONV", "2"))
            for bit, n in ((1, "conv7"), (2, "conv9")):
                if sd & bit:
                    # transposed layers with split operands (csrc/conv3d_mfma16.hip deconvgs): six cross products,
                    # 9 (z, y)-tap combos in 10 k-slots, x parity folded into N (a quarter of the panel is zero)
                    ex[n]["flops"] = costs[n]["flops"] * 6.0 * (10.0 / 9.0) * (4.0 / 3.0)
                    ex[n]["mfma_peak"] = MFMA_16BIT_PEAK_TFLOPS
                    ex[n]["arith"] = "3xbf16 split operands, six cross products, fp32 accumulate"
            if env.get("MVS_TAIL_SPLIT") != "0" and env.get("MVS_FUSE_PROB") != "0":
                # the fused tail (csrc/conv11_prob.hip, conv11_prob_split_kernel): conv11 with split operands on the bf16
                # matrix cores -- six cross products, 9 (z, y)-tap combos in 10 k-slots, x parity folded into N (one of
                # four (dx, px) blocks of the panel is zero): 6 x 10/9 x 4/3 = 8.9x its multiply-adds -- and the prob
                # stencil as packed fp32 FMAs on the vector units: two units that can work side by side, so the
                # compute floor is the larger of the two times
                f11 = costs["conv11"]["flops"] * 6.0 * (10.0 / 9.0) * (4.0 / 3.0)
                ex["conv11_prob"]["parts"] = [(f11, MFMA_16BIT_PEAK_TFLOPS), (costs["prob"]["flops"], MFMA_F32_PEAK_TFLOPS)]
                ex["conv11_prob"]["flops"] = f11 + costs["prob"]["flops"]
                ex["conv11_prob"]["arith"] = "conv11: 3xbf16 split operands, six cross products, fp32 accumulate; prob: fp32 vector FMAs"
    else:
        for n in ("prob", "conv11_prob", "softargmin"):
            ex[n]["bytes"] = costs[n]["bytes"] + V0 * 2
        if env.get("MVS_FEAT16") != "1":     # the warp gathers from the fp32 feature copy
            ex["warp_variance"]["bytes"] = costs["warp_variance"]["bytes"] + N * 32 * h * w * 2
    return ex


def conv0_split_enabled(env=None):
    """conv0 of the fp32 path runs with split bf16 operands unless MVS_CONV0_SPLIT=0 (csrc/conv3d_direct.hip)."""
    env = os.environ if env is None else env
    return env.get("MVS_CONV0_SPLIT") != "0"


def compute_floor_s(c_ex, mfma_peak):
    """Seconds the executed arithmetic needs at the peak of the unit(s) it runs on: `parts` = [(flops, peak TFLOP/s)]
    for a stage whose work sits on two units that can overlap (the larger time), else flops / the stage's peak."""
    if "parts" in c_ex:
        return max(f / (p * 1e12) for f, p in c_ex["parts"])
    return c_ex["flops"] / (c_ex.get("mfma_peak", mfma_peak) * 1e12)


def stage_entry(ms, c_alg, c_ex, mfma_peak):
    """One `stages` entry: measured ms, the rates, and two roofline fractions -- `frac` from what the kernel really
    executes / moves (<= 1 by construction), `frac_algorithmic` from SURVEY d3's algorithmic bytes / FLOPs (a Winograd
    kernel can exceed 1 there: it issues fewer multiply-adds than the algorithm counts)."""
    ent = {"ms": round(ms, 4)}
    if not c_alg or ms <= 0:
        return ent
    ent["GBps"] = round(c_alg["bytes"] / ms / 1e6, 1)
    if c_alg["flops"]:
        ent["TFLOPs"] = round(c_alg["flops"] / ms / 1e9, 2)
    t_hbm, t_mfma = c_ex["bytes"] / (HBM_PEAK_GBPS * 1e6), compute_floor_s(c_ex, mfma_peak) * 1e3
    ent["bound"] = "mfma" if t_mfma > t_hbm else "hbm"
    ent["frac"] = round(max(t_hbm, t_mfma) / ms, 3)
    if "arith" in c_ex:
        ent["arith"] = c_ex["arith"]
    alg_ms = max(c_alg["bytes"] / (HBM_PEAK_GBPS * 1e6), c_alg["flops"] / (mfma_peak * 1e9))
    ent["frac_algorithmic"] = round(alg_ms / ms, 3)
    return ent


def roofline_entry(dom, ms, c_alg, c_ex, mfma_peak):
    """The `roofline` object of the dominant kernel.  `achieved` / `frac` are the EXECUTED rate (what the matrix pipe
    or HBM really did per second); the algorithmic rate of SURVEY d3 travels as `achieved_algorithmic` /
    `algorithmic_ratio` (not a fraction of peak: > 1 is possible for a Winograd kernel)."""
    t_hbm = c_ex["bytes"] / (HBM_PEAK_GBPS * 1e9)
    ex_peak = c_ex.get("mfma_peak", mfma_peak)     # peak of the matrix unit the kernel really runs on
    t_mfma = compute_floor_s(c_ex, mfma_peak) if "parts" not in c_ex else 0.0   # a two-unit stage is priced on its bytes here
    if t_mfma > t_hbm:
        ach, alg = c_ex["flops"] / ms / 1e9, c_alg["flops"] / ms / 1e9
        r = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": ex_peak, "unit": "TFLOP/s",
             "frac": round(ach / ex_peak, 4), "traffic": None, "avg_launch_ms": ms,
             "achieved_algorithmic": round(alg, 3), "algorithmic_ratio": round(alg / mfma_peak, 4),
             "algorithmic_flops": c_alg["flops"], "executed_flops": c_ex["flops"],
             "algorithmic_bytes": c_alg["bytes"]}
        if "arith" in c_ex:
            r["note"] = (f"{dom}: {c_ex['arith']} on the bf16 matrix cores (Winograd F(4,3) along z, Toeplitz-pair form): "
                         f"it issues {c_ex['flops'] / c_alg['flops']:.2f}x the algorithmic multiply-adds as bf16 MFMA work; "
                         f"`frac` = executed bf16 MFMA flops / time / the bf16 peak ({ex_peak:g} TF); `algorithmic_ratio` = "
                         f"algorithmic fp32 flops / time / the fp32 MFMA peak ({mfma_peak:g} TF; it exceeds 1: the fp32 "
                         "matrix pipe is no longer what bounds this layer).  HBM floor of the layer: "
                         f"{c_ex['bytes'] / HBM_PEAK_GBPS / 1e6:.3f} ms")
        elif c_ex["flops"] != c_alg["flops"]:
            r["note"] = (f"{dom} runs a Winograd transform along z: it issues {c_ex['flops'] / c_alg['flops']:.3f} of "
                         "the algorithmic multiply-adds; `frac` = executed MFMA flops / time / peak")
    else:
        ach, alg = c_ex["bytes"] / ms / 1e6, c_alg["bytes"] / ms / 1e6
        r = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "avg_launch_ms": ms,
             "achieved_algorithmic": round(alg, 1), "algorithmic_ratio": round(alg / HBM_PEAK_GBPS, 4),
             "algorithmic_bytes": c_alg["bytes"], "executed_bytes": c_ex["bytes"]}
    return r


# stage -> (kernel-name substring in the rocprofv3 trace, tools/prof_stage.py selector, FETCH_SIZE factor).
# gfx950: FETCH_SIZE under-reports wide coalesced 16 B/lane streaming reads by 2x (MI355X_MICROARCH.md, HBM section):
# valid for the staging loads of the conv kernels, NOT for the scattered tap gathers of the warp kernel (x1).
TRAFFIC_KERNELS = {
    "conv0": ("conv0", "conv0", 2.0),
    "warp_variance": ("warp_variance_tc2_kernel", "warp", 1.0),
    "conv11_prob": ("conv11_prob", "all", 2.0),
}


def committed_traffic(stage, config):
    """HBM bytes per launch of `stage`'s kernel from the newest committed PMC profile of that config
    (profiles/rNN_traffic.json = cfg2, profiles/rNN_traffic_<cfg>.json otherwise) -> (bytes, source) or (None, None)."""
    import glob
    if stage not in TRAFFIC_KERNELS:
        return None, None
    sub, _, factor = TRAFFIC_KERNELS[stage]
    pat = "r*_traffic.json" if config == "cfg2" else f"r*_traffic_{config}.json"
    try:
        newest = sorted(glob.glob(os.path.join(REPO, "profiles", pat)))[-1]
        with open(newest) as f:
            prof = json.load(f)["kernels"]
        ents = [v for k, v in prof.items() if sub in k and "FETCH_SIZE_KB_avg" in v]
        if not ents:
            return None, None
        ent = max(ents, key=lambda v: v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"])
        val = int((factor * ent["FETCH_SIZE_KB_avg"] + ent["WRITE_SIZE_KB_avg"]) * 1024)
        return val, (f"profiles/{os.path.basename(newest)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same "
                     f"kernel, per launch; FETCH x{factor:g} -- the gfx950 correction of this kernel's load pattern)")
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def live_traffic(kernel_substr, what, reps=3, timeout=150, cfg="cfg2", storage="f32", fetch_factor=2.0):
    """HBM bytes per launch of one kernel, measured NOW: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE;
    each with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over `tools/prof_stage.py <what> <reps> <cfg>
    <dtype>` -- the same kernel on the same inputs -- run as child processes of this one.  `fetch_factor`: the gfx950
    FETCH_SIZE correction for that kernel's load pattern (TRAFFIC_KERNELS).  Returns None when rocprofv3 is missing,
    fails or times out."""
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
                       sys.executable, os.path.join(REPO, "tools", "prof_stage.py"), what, str(reps), cfg, storage]
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
    return int((fetch_factor * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)


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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the path-only runs of BASELINE configs 3 (bf16 1600x1184x256) and 5 (fp16 N=4) that the "
                         "default cfg2 one-GPU run adds as `other_configs`")
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


class Ctx:
    """Process-wide state shared by the per-config measurements."""
    pass


def measure(ctx, config, storage, K, Wm, prewarm_ms, S, KS, staged_timed=False):
    """Time the hot path on one BASELINE config: W warm-up steps, the K steps right after them (`first`), the same K
    steps again after `prewarm_ms` of the same workload (`sustained`), then KS maps through the per-stage C-ABI calls
    with a HIP event after each kernel.  Returns a dict; every rank takes part (barriers + gather inside the timed
    region when world > 1)."""
    torch, dist, _lib, sharding, synthetic = ctx.torch, ctx.dist, ctx._lib, ctx.sharding, ctx.synthetic
    dev, rank, world, rehearsal = ctx.dev, ctx.rank, ctx.world, ctx.rehearsal
    cfg = synthetic.CONFIGS[config]
    N, D, h, w = cfg["nviews"], cfg["D"], cfg["H"] // 4, cfg["W"] // 4
    dt = _lib.dtype_code(storage)
    es = 4 if storage == "f32" else 2
    lib = _lib.load()

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
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(max(K, KS))]

    # Pre-allocated per-stream buffers and pre-bound C calls keep the host ahead of the GPU in the
    # staged mode (no torch allocations or shape logic inside the timed loop).
    def lvl(c, l):
        return torch.empty((c // 8, D >> l, h >> l, w >> l, 8), dtype=_lib.TORCH_DTYPES[dt], device=dev)

    bufs = []
    for si in range(S if staged_timed else 1):
        bufs.append(dict(rt=torch.empty((max(N - 1, 1), 12), dtype=torch.float32, device=dev),
                         var=lvl(32, 0),
                         act=[lvl(8, 0), lvl(16, 1), lvl(16, 1), lvl(32, 2), lvl(32, 2), lvl(64, 3),
                              lvl(64, 3), lvl(32, 2), lvl(16, 1), lvl(8, 0)],
                         cost=torch.empty((D, h, w), dtype=torch.float32, device=dev)))
    skips = {7: 4, 8: 2, 9: 0}

    def step_staged(k, ev=None):
        si = k % len(bufs)
        ws, B = wss[k % S], bufs[si]
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
    step_one = step_staged if staged_timed else step_fused

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

    # ---- per-kernel durations: the same kernels on the same inputs through the per-stage C-ABI
    # calls, a HIP event (on the launch stream) after each, one stream, right after the timed region
    costs = stage_costs(N, D, h, w, es)
    ex_costs = executed_costs(costs, storage, N, D, h, w)
    mfma_peak = mfma_peak_tflops(storage, os.environ.get("MVS_MFMA16") != "0")
    stages = {}
    if KS:
        step_staged(0)
        torch.cuda.synchronize()
        for k in range(KS):
            step_staged(k % K, events[k])
        torch.cuda.synchronize()
        for si, name in enumerate(stage_names):
            ms = float(np.mean([events[k][si].elapsed_time(events[k][si + 1]) for k in range(KS)]))
            stages[name] = stage_entry(ms, costs.get(name), ex_costs.get(name), mfma_peak)
    roofline = None
    if stages:
        dom = max((n for n in stages if n in costs), key=lambda n: stages[n]["ms"])
        roofline = roofline_entry(dom, stages[dom]["ms"], costs[dom], ex_costs[dom], mfma_peak)
        # HBM traffic of the dominant kernel: the committed PMC profile of this config (separate rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE passes); main() replaces it by a live measurement at the very end when it can
        roofline["traffic"], src = committed_traffic(dom, config)
        if src:
            roofline["traffic_source"] = src

    # whole-path totals follow SURVEY.md §8 d3 (layer-by-layer, no fusion credited), independent of which kernels
    # ran; the executed floor prices what this build's kernels really issue / move (fused tail counted once)
    path_bytes, path_flops, stagewise_floor_s = path_totals(costs, mfma_peak)
    ran = [n for n in stage_names if n in ex_costs]
    executed_floor_s = sum(max(ex_costs[n]["bytes"] / (HBM_PEAK_GBPS * 1e9), compute_floor_s(ex_costs[n], mfma_peak))
                           for n in ran)
    res = dict(config=config, storage=storage, N=N, D=D, h=h, w=w, H=cfg["H"], W=cfg["W"], K=K, Wm=Wm, S=S, KS=KS,
               elapsed=elapsed, first_elapsed=first_elapsed, effective_warmup=effective_warmup,
               maps_per_s=world * K / elapsed, ms_per_step=elapsed / K * 1e3, per_rank=list(per_rank),
               stages=stages, roofline=roofline, path_bytes=path_bytes, path_flops=path_flops,
               stagewise_floor_s=stagewise_floor_s, executed_floor_s=executed_floor_s, in_image_frac=in_image_frac,
               call=("staged C-ABI calls with HIP events" if staged_timed else "one mvs_depth_infer call per map"),
               last_depth=out[K - 1, 0], inputs=(feats_np, proj_np, dv_np, sd), streams=streams)
    return res


def brief(res, world=1):
    """The `other_configs` entry of one further BASELINE config (same two-pass timing as `value`, path-only)."""
    st = res["stages"]
    dom = res["roofline"]
    out = {"value": round(res["maps_per_s"], 2), "unit": "depth maps/s", "ms_per_step": round(res["ms_per_step"], 4),
           "first_pass": round(world * res["K"] / res["first_elapsed"], 2), "steps": res["K"],
           "effective_warmup_steps": res["effective_warmup"],
           "dtype": f"{res['storage']} storage and MFMA operands, f32 accumulation" if res["storage"] != "f32" else "f32",
           "workload": f"{res['config']}: N={res['N']} views, {res['H']}x{res['W']} image -> {res['h']}x{res['w']} "
                       f"features, D={res['D']}, {res['storage']} volumes; path-only",
           "frac_of_stagewise_roofline": round(res["stagewise_floor_s"] / (res["elapsed"] / res["K"]), 4),
           "frac_of_executed_roofline": round(res["executed_floor_s"] / (res["elapsed"] / res["K"]), 4),
           "hbm_GBps_algorithmic": round(res["path_bytes"] * res["maps_per_s"] / 1e9, 1),
           "in_image_frac": res["in_image_frac"]}
    if dom:
        out["dominant"] = {k: dom[k] for k in ("kernel", "bound", "achieved", "unit", "frac", "avg_launch_ms", "traffic")
                           if k in dom}
    out["stages_ms"] = {n: e["ms"] for n, e in st.items()}
    out["stages_frac"] = {n: e["frac"] for n, e in st.items() if "frac" in e}
    return out


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

    import torch
    import torch.distributed as dist
    from scene_3dreconstruction_mvsnet_amd import _lib, sharding, synthetic
    # one process per GPU: keep this rank's host threads on the cores next to its GPU (before any GPU call)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    pinned = sharding.pin_rank(local_rank, local_world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # MVS_BENCH_REHEARSAL=1: run every rank on cuda:0 with the gloo backend -- a way to exercise
    # the N>1 code path on a one-GPU box (numbers are meaningless; the driver never sets it)
    rehearsal = os.environ.get("MVS_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pin_check = None
    if pinned and not rehearsal:
        # is the GPU HIP gave this rank the one whose cores were chosen from sysfs before HIP was up?
        pr = torch.cuda.get_device_properties(dev_index)
        pin_check = sharding.verify_pinning(local_rank, f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
        if not pin_check["match"]:
            print(f"[bench] rank {rank}: pinned to the cores of {pin_check['assumed_bus_id']} but HIP device "
                  f"{dev_index} is {pin_check['hip_bus_id']} (affinity only; results unaffected)", file=sys.stderr)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx = Ctx()
    ctx.torch, ctx.dist, ctx._lib, ctx.sharding, ctx.synthetic = torch, dist, _lib, sharding, synthetic
    ctx.dev, ctx.rank, ctx.world, ctx.rehearsal = dev, rank, world, rehearsal
    _lib.load()

    K, Wm = args.steps, args.warmup
    storage = args.dtype or {"cfg3": "bf16", "cfg5": "f16"}.get(args.config, "f32")
    S = max(1, args.streams)
    prewarm_ms = max(0, args.prewarm_ms)
    KS = max(0, args.staged_steps)
    res = measure(ctx, args.config, storage, K, Wm, prewarm_ms, S, KS, staged_timed=args.staged_timed)
    N, D, h, w = res["N"], res["D"], res["h"], res["w"]
    cfg = synthetic.CONFIGS[args.config]
    maps_per_s, ms_per_step, elapsed, first_elapsed = res["maps_per_s"], res["ms_per_step"], res["elapsed"], res["first_elapsed"]
    stages, roofline, streams = res["stages"], res["roofline"], res["streams"]
    feats_np, proj_np, dv_np, sd = res["inputs"]

    # ---- the other single-GPU BASELINE configs (cfg3: bf16 1600x1184x256; cfg5: fp16 N=4), path-only, same two-pass
    # timing, in this process right after the metric's own timed region.  They are parity-test cases and a report,
    # never `value`.  One-GPU runs of the default config only; K scaled so each costs about a second of GPU time.
    other_configs = None
    if args.config == "cfg2" and storage == "f32" and world == 1 and not args.no_other_configs:
        other_configs = {}
        for oc, ost, ok in (("cfg3", "bf16", max(4, min(K, 20))), ("cfg5", "f16", max(4, min(2 * K, 60)))):
            torch.cuda.empty_cache()
            try:
                r = measure(ctx, oc, ost, ok, Wm, prewarm_ms, S, min(KS, 5))
                other_configs[oc] = brief(r, world)
                del r
            except RuntimeError as e:      # report, never hide: the default line must still print
                other_configs[oc] = {"error": str(e)[:300]}
        torch.cuda.empty_cache()

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
        got = res["last_depth"].cpu().numpy()
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
        # the reference's loader makes these floats from 8-bit pixels (np.array(img, float32) / 255,
        # datasets/data_io.py:143): the uint8 form of the same images is what `h2d_uint8` copies
        imgs_u8_np = np.clip(np.rint(imgs_np * 255.0), 0, 255).astype(np.uint8)
        imgs_np = imgs_u8_np.astype(np.float32) / np.float32(255.0)
        imgs_h = torch.from_numpy(imgs_np).pin_memory()
        imgs_u8_h = torch.from_numpy(imgs_u8_np).pin_memory()
        proj_i, dv_i = torch.from_numpy(proj_i).to(dev), torch.from_numpy(dv_i).to(dev)
        end_to_end = {"unit": "depth maps/s", "includes": "FeatureNet (HIP) + path; h2d adds the "
                      "pinned-host -> HBM copy of the N float32 images on the same stream, h2d_uint8 copies the same "
                      "images as uint8 (4x fewer bytes; the division by 255 happens in FeatureNet's first kernel, "
                      f"bit-equal to the reference's loader); K maps after {prewarm_ms} ms of the same calls (untimed)"}
        # like the path-only figure: forwards round-robin over the S streams (the module keeps one workspace per
        # stream); in h2d mode every forward first copies its own images on its stream
        S_ = len(streams)
        accepts_u8 = getattr(MVSNet, "ACCEPTS_UINT8_IMAGES", False)

        def e2e_forward(i, mode):
            with torch.cuda.stream(streams[i % S_]):
                if mode == "h2d":
                    x_d = imgs_h.to(dev, non_blocking=True)
                elif mode == "h2d_uint8":
                    x_d = imgs_u8_h.to(dev, non_blocking=True)
                else:
                    x_d = imgs_d
                model(x_d, proj_i, dv_i)

        for mode in ("resident", "h2d") + (("h2d_uint8",) if accepts_u8 else ()):
            imgs_d = imgs_h.to(dev)
            for i in range(3 * S_):
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
        end_to_end["streams"] = S_

    # live PMC measurement of the dominant kernel's HBM traffic (rank 0 of a one-GPU run; after every timing)
    if (rank == 0 and world == 1 and roofline is not None and not args.no_live_traffic
            and roofline["kernel"] in TRAFFIC_KERNELS):
        torch.cuda.synchronize()
        sub, what, factor = TRAFFIC_KERNELS[roofline["kernel"]]
        t_live = time.perf_counter()
        live = live_traffic(sub, what, cfg=args.config, storage=storage, fetch_factor=factor)
        if live:
            roofline["traffic"] = live
            roofline["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child "
                                          f"passes, --kernel-trace only) over tools/prof_stage.py {what} 3 {args.config} {storage}, "
                                          f"per launch; FETCH x{factor:g} (gfx950 correction for this kernel's load pattern); "
                                          f"{time.perf_counter() - t_live:.0f} s")

    if rank == 0:
        conv0_split = (storage == "f32" and conv0_split_enabled() and os.environ.get("MVS_FORCE_DIRECT") != "1"
                       and os.environ.get("MVS_CONV0_WINO") != "0" and D % 4 == 0)
        line = {
            "metric": "depth maps/sec at N=5 views, 640x512, D=192; achieved HBM GB/s"
                      if args.config == "cfg2" else f"depth maps/sec ({args.config})",
            "value": round(maps_per_s, 3), "unit": "depth maps/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "effective_warmup_steps": res["effective_warmup"],
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (conv0, conv2-4, conv9, conv11: 3xbf16 split operands on the bf16 matrix cores, fp32 accumulate)"
                      if conv0_split and os.environ.get("MVS_SPLIT_LAYERS") != "0" else
                      "f32 (conv0: 3xbf16 split operands, fp32 accumulate)" if conv0_split else "f32") if storage == "f32" else (
                f"{storage} storage, f32 MFMA arithmetic" if os.environ.get("MVS_MFMA16") == "0"
                else f"{storage} storage and MFMA operands, f32 accumulation"),
            "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N} views, {cfg['H']}x{cfg['W']} image -> "
                                   f"{h}x{w} features, D={D}, C=32, {storage} volumes; path-only (features "
                                   "resident in HBM -> depth+confidence)",
                       "maps_per_rank": K, "sharding": "independent ref views per rank, one RCCL "
                                                       "all-gather of results at the end",
                       "call": res["call"] +
                               (f"; per-kernel durations from {KS} further maps through the per-stage C-ABI "
                                "calls with HIP events, after the timed region" if KS else ""),
                       "in_image_frac": res["in_image_frac"],
                       "prewarm_ms": prewarm_ms,
                       "streams": S},
            "first_pass": {"value": round(world * K / first_elapsed, 3), "ms_per_step": round(first_elapsed / K * 1e3, 4),
                           "note": f"the same {K} steps timed right after the {Wm} warm-up steps, before the device "
                                   f"clocks had settled; `value` is the same pass repeated after {prewarm_ms} ms of "
                                   "the same workload (untimed)"},
            "hbm_GBps_algorithmic": round(res["path_bytes"] * maps_per_s / 1e9, 1),
            "hbm_frac_of_peak": round(res["path_bytes"] * maps_per_s / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "path": {"algorithmic_bytes": res["path_bytes"], "algorithmic_flops": res["path_flops"],
                     "stagewise_roofline_ms": round(res["stagewise_floor_s"] * 1e3, 4),
                     "frac_of_stagewise_roofline": round(res["stagewise_floor_s"] / (elapsed / K), 4),
                     "executed_roofline_ms": round(res["executed_floor_s"] * 1e3, 4),
                     "frac_of_executed_roofline": round(res["executed_floor_s"] / (elapsed / K), 4),
                     "note": "stagewise = SURVEY d3 algorithmic bytes / FLOPs per stage; executed = what this build's "
                             "kernels issue and move (Winograd layers fewer multiply-adds, fused tail once, fp32 logits "
                             "in the 16-bit modes)"},
            "per_rank": {"maps_per_s": [round(K / t, 2) for t in res["per_rank"]], "host_cores": len(pinned) or None,
                         "pin_check": pin_check,
                         "note": "each rank's K maps / its own clock around the timed region (barriers and the "
                                 "gather included); `value` = world*K / the slowest"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "end_to_end": end_to_end, "stages": stages,
            "other_configs": other_configs,
            "parity_rel_l1_vs_oracle": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
