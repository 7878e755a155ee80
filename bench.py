#!/usr/bin/env python3
"""bench.py - cost-volume voxels/s (CVC+CVF+WTA) of the HIP DispEst hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic stereo pair already resident in HBM:
CostConst (gray/gradient + both cost volumes) -> CostFilter (guidance precompute + guided filter
of both volumes) -> DispSel (WTA; for N > 1: the fused kernel's packed per-pixel minima over the local slices -> ONE RCCL
all-reduce(MIN) per frame (or all-gather + device-side minimum) -> final maps).  Workload at N=1: BASELINE.json configs[3], 1920x1080, D=256,
float32 - the configuration the metric is quoted on; for N > 1 the D slices of that same job are
sharded over the ranks (total work fixed -> "scaling": "strong").

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - dominant kernel: algorithmic HBM bytes per launch / its mean hipEvent duration
  cpu_baseline - the CPU oracle (restatement of the reference pthreads path) timed on a bounded
                 sample on this box's host cores (rank 0, N=1 only); a reported baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
CONFIGS = {
    # name: (W, H, D, description)
    "c4": (1920, 1080, 256, "synthetic 1920x1080 pair, D=256, float32 (BASELINE configs[3])"),
    "c3": (1280, 720, 128, "synthetic 1280x720 pair, D=128, float32 (BASELINE configs[2])"),
    "c5": (3840, 2160, 256, "synthetic 3840x2160 pair, D=256, float32 (BASELINE configs[4])"),
    "c2": (450, 375, 64, "synthetic 450x375 pair, D=64, float32 (size of BASELINE configs[1])"),
    # 8-bit char mode (BASELINE configs[0]): the shipped Cones size and the 384x288 the json quotes
    "c1": (450, 375, 64, "synthetic 450x375 pair, D=64, 8-bit char mode (size of the shipped Cones pair, BASELINE configs[0])"),
    "c1x": (384, 288, 64, "synthetic 384x288 pair, D=64, 8-bit char mode (the size BASELINE configs[0] quotes)"),
}
# algorithmic bytes per voxel of the staged 8-bit pipeline: CVC 1 W; CVF stage A 1 R + 16 W, stage B 16 R + 1 W; WTA 1 R
ALG_BYTES_U8 = {"cvf_fused": 34.0, "cvc": 1.0, "wta": 1.0, "pipeline": 36.0}
# algorithmic HBM bytes per voxel (SURVEY.md 8d / DESIGN.md): stage A 4 R + 16 W, stage B 16 R + 4 W
ALG_BYTES = {"cvf_fused": 40.0, "cvf_a": 20.0, "cvf_b": 20.0, "cvc": 4.0, "wta": 4.0, "box8": 8.0, "pipeline": 48.0}


def spawn_ranks(n):
    """Re-run this command as n ranks (one per GPU) under torch.distributed.run; returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="", choices=["", "f32", "u8"], help="volume element type (default: f32, u8 for the c1 configs)")
    ap.add_argument("--seg-rows", type=int, default=-1, help="marching-kernel y segment (-1: library default)")
    ap.add_argument("--waves", type=int, default=0, help="waves per workgroup (0: library default)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flags", type=int, default=-1, help="PSM_OPT_FLAGS tuning bits (-1: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-d", type=int, default=0,
                    help="disparities in the CPU-baseline sample (0 = auto: the whole D at 1080p and below (~10 s on 8 threads), "
                         "proportionally fewer for larger images)")
    ap.add_argument("--cpu-wide", action="store_true", help="also time the CPU baseline on min(64, host cores) threads (doubles its run time)")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even for N=1")
    ap.add_argument("--exchange", default="allreduce", choices=["allreduce", "allgather", "none"],
                    help="N>1 exchange step: one all_reduce(MIN) of the packed keys (default; ~2x33 MB per rank "
                         "at 1080p whatever N) or one all_gather ((N-1)x33 MB per rank) + device-side minimum")
    ap.add_argument("--shard", default="rows", choices=["rows", "disp"],
                    help="N > 1: what a rank owns - 'rows': a stripe of H/N output rows of both volumes, all D slices (no minima "
                         "exchanged, one all-gather of the finished map rows per frame); 'disp': D/N slices of both volumes, whole "
                         "image (one collective on packed per-pixel minima per frame, --exchange)")
    ap.add_argument("--lr-check", type=int, default=-1, choices=[-1, 0, 1],
                    help="PP left-right check on the GPU inside the step (BASELINE configs[4]); -1: on for config c5 only")
    ap.add_argument("--box-bench", action="store_true", help="also time the plain box-filter pass")
    ap.add_argument("--fgf", type=int, default=0, choices=[0, 2, 4, 8],
                    help="diagnostic: aggregate with the Fast Guided Filter variant (CostFilter_FGF) at this subsample "
                         "rate instead of the full guided filter; not the north-star metric")
    ap.add_argument("--no-overlap", action="store_true", help="(kept for old command lines; same as --no-frame-pipeline)")
    ap.add_argument("--no-frame-pipeline", action="store_true",
                    help="N>1: finish the exchange + merge of a frame inside its own step instead of one step later "
                         "(default: a frame's all-reduce overlaps the next frame's filter; two key tensors)")
    ap.add_argument("--verify", action="store_true",
                    help="N=1: also compare the maps of the timed path with a fresh single-context run (always done for N>1)")
    ap.add_argument("--shard-sim", type=int, default=0,
                    help="diagnostic: time only rank 0's disparity shard of a G-rank job on this GPU (no exchange); "
                         "the JSON line is then NOT the headline metric")
    args = ap.parse_args()

    if args.no_overlap:
        args.no_frame_pipeline = True
    N = args.gpus
    if N < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.shard == "disp" and N > CONFIGS[args.config][2]:
        raise SystemExit(f"bench.py: --gpus {N} exceeds the {CONFIGS[args.config][2]} disparity slices of config {args.config} (one shard per rank)")
    if args.shard == "rows" and (N > CONFIGS[args.config][1] or (N - 1) * -(-CONFIGS[args.config][1] // N) >= CONFIGS[args.config][1]):
        # (every rank decides this the same way, before any rendezvous: stripes of ceil(H / N) rows must leave the last rank some)
        raise SystemExit(f"bench.py: --gpus {N}: stripes of {-(-CONFIGS[args.config][1] // N)} rows leave a rank without rows of the "
                         f"{CONFIGS[args.config][1]}-row image of config {args.config} (one stripe per rank)")
    if N > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher - one rank per GPU under torch.distributed.run on
        # 127.0.0.1 - and pass rank 0's single JSON line through
        return spawn_ranks(N)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = (N > 1) or args.force_dist
    torch = dist = None
    json_fd = None
    if use_dist:
        # RCCL prints a version banner on the C-level stdout (flushed at exit, i.e. after our line): keep stdout for the ONE
        # JSON line only - everything else written to fd 1 by this process goes to stderr
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        # torch first: libprimesm_hip.so then binds to the HIP runtime torch already loaded
        # (same SONAME libamdhip64.so.7), so device pointers are interchangeable.
        import torch
        import torch.distributed as dist
        if world != N:
            raise SystemExit(f"bench.py --gpus {N}: WORLD_SIZE={world} in the environment does not match")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import numpy as np
    import primestereomatch_amd as P
    from primestereomatch_amd import capi, synth

    W, H, D, desc = CONFIGS[args.config]
    dtype = args.dtype or ("u8" if args.config.startswith("c1") else "f32")
    if dtype == "u8":
        ALG_BYTES.update(ALG_BYTES_U8)
        if not args.config.startswith("c1"):
            desc = desc.replace("float32", "8-bit char mode")
    rows_mode = args.shard == "rows" and not args.fgf and (use_dist or args.shard_sim > 1)
    from primestereomatch_amd import stripes
    parts = world if use_dist else max(args.shard_sim, 1)
    rows_max, y0, y1 = H, 0, H
    if rows_mode:
        # stripes aligned at multiples of ceil(H / parts): the gathered tensor is the image
        rows_max, y0, y1 = stripes.stripe_bounds(H, parts, rank if use_dist else 0)
        if y1 <= y0:
            raise SystemExit(f"bench.py: rank {rank} of {world} has no rows of the {H}-row image")
        d0, d1 = 0, D
    elif use_dist:
        d0, d1 = D * rank // world, D * (rank + 1) // world
    elif args.shard_sim > 1:
        d0, d1 = 0, D // args.shard_sim
    else:
        d0, d1 = 0, D
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    de = P.DispEst(l, r, D, 8, True, device=local_rank, d_range=(d0, d1), dtype=dtype)
    if args.seg_rows >= 0:
        de.set_option(capi.PSM_OPT_SEG_ROWS, args.seg_rows)
    if args.waves:
        de.set_option(capi.PSM_OPT_WAVES, args.waves)
    de.set_option(capi.PSM_OPT_KERNEL_VARIANT, args.variant)
    if args.flags >= 0:
        de.set_option(capi.PSM_OPT_FLAGS, args.flags)
    de.set_option(capi.PSM_OPT_ASYNC, 1)
    if rows_mode:
        de.set_rows(y0, y1)

    keys_local = keys_all = None
    kbuf = []
    pending = []                 # (work handles, key buffer) of the frame whose merge is still outstanding
    frame = [0]
    if use_dist:
        HW2 = 2 * H * W
        kbuf = [torch.empty(HW2, dtype=torch.int64, device="cuda") for _ in range(2)]
        keys_local = kbuf[0]
        keys_all = torch.empty(world * HW2 if args.exchange == "allgather" else 1, dtype=torch.int64, device="cuda")
        if rows_mode:
            # the maps of a frame are written by the library straight into one of two torch tensors (psm_set_map_buffer);
            # the stripe rows go through one all_gather per frame: [world][2][rows_max][W] uint8 = the two whole maps
            mbuf = [torch.zeros(HW2 + 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
            send = torch.zeros(2 * rows_max * W, dtype=torch.uint8, device="cuda")
            recv = torch.zeros(world * 2 * rows_max * W, dtype=torch.uint8, device="cuda")
        # one non-default torch stream carries both our kernels and the RCCL collective, so the
        # exchange is ordered against the kernels without host synchronisation
        side_stream = torch.cuda.Stream()
        torch.cuda.set_stream(side_stream)
        de.set_stream(side_stream.cuda_stream)

    if args.fgf:
        de.setSubsampleRate(args.fgf)

    pipelined = use_dist and (args.exchange == "allreduce" or rows_mode) and not args.no_frame_pipeline and not args.fgf
    # BASELINE configs[4]: "+ PP left-right check on-GPU" - lrCheck (src/PP.cpp:17-50) on the finished maps, part of the step
    lrc = (args.lr_check == 1 or (args.lr_check < 0 and args.config == "c5")) and not (args.shard_sim > 1)

    def finish_pending():
        # exchange of an earlier frame -> final maps (the collective ran on RCCL's stream meanwhile)
        while pending:
            works, kb = pending.pop(0)
            for w_ in works:
                w_.wait()
            if rows_mode:
                stripes.assemble(recv, world, H, W, rows_max, kb)    # [rank][side][row][x] -> [side][y][x]
                de.set_map_buffer(kb.data_ptr(), whole=True)
            else:
                de.DispSelect_merge(kb.data_ptr(), 1, download=False)
            if lrc:
                de.LRCheck_device()

    def stripe_exchange(mb, async_op):
        stripes.pack_stripe(mb, y0, y1, send, H, W, rows_max)
        return dist.all_gather_into_tensor(recv, send, async_op=async_op)      # the one exchange step (RCCL)

    def step():
        de.CostConst_GPU()
        if args.fgf:
            de.CostFilter_FGF_GPU()
            if use_dist:
                de.DispSelect_partial(keys_local.data_ptr())
                dist.all_reduce(keys_local, op=dist.ReduceOp.MIN)
                de.DispSelect_merge(keys_local.data_ptr(), 1, download=False)
            elif args.shard_sim > 1:
                de.DispSelect_partial()
            else:
                de.DispSelect_device()
            return
        if rows_mode and use_dist:
            # Row stripes: nothing of the cost volumes or their minima leaves the rank; the finished rows of both maps are
            # gathered - asynchronously, behind the next frame's filter when pipelined (maps alternate between two tensors).
            mb = mbuf[frame[0] & 1]
            frame[0] += 1
            de.set_map_buffer(mb.data_ptr())
            de.CostFilter_GPU()
            de.DispSelect_device()
            finish_pending()
            if pipelined:
                pending.append(((stripe_exchange(mb, True),), mb))
            else:
                stripe_exchange(mb, False)
                pending.append(((), mb))
                finish_pending()
            return
        if pipelined:
            # Frame pipeline, ONE collective per frame: the fused kernel leaves the packed minima of both volumes directly
            # in this frame's key tensor; the all-reduce(MIN) over the ranks is asynchronous and has the whole next
            # frame's filter to complete - its merge is issued after that filter, so no kernel of ours ever waits for a
            # collective that is still running.  Keys alternate between two tensors.
            kb = kbuf[frame[0] & 1]
            frame[0] += 1
            de.set_key_buffer(kb.data_ptr())
            de.CostFilter_GPU()
            finish_pending()
            w_ = dist.all_reduce(kb, op=dist.ReduceOp.MIN, async_op=True)
            pending.append(((w_,), kb))
            return
        if use_dist:
            de.set_key_buffer(keys_local.data_ptr())
        de.CostFilter_GPU()
        if use_dist:
            if args.exchange == "none":       # diagnostic only: cost of the torch collective call itself
                de.DispSelect_merge(keys_local.data_ptr(), 1, download=False)
            elif args.exchange == "allreduce":
                dist.all_reduce(keys_local, op=dist.ReduceOp.MIN)     # the one exchange step (RCCL)
                de.DispSelect_merge(keys_local.data_ptr(), 1, download=False)
            else:
                dist.all_gather_into_tensor(keys_all, keys_local)     # the one exchange step (RCCL)
                de.DispSelect_merge(keys_all.data_ptr(), world, download=False)
        elif args.shard_sim > 1 and not rows_mode:
            de.DispSelect_partial()
        else:
            de.DispSelect_device()
        if lrc and not (use_dist and args.exchange == "none"):
            de.LRCheck_device()

    def sync():
        finish_pending()
        if use_dist:
            torch.cuda.synchronize()
        de.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync(); barrier(); sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    voxels_per_step = 2.0 * W * H * D           # both volumes, all ranks
    value = voxels_per_step / (elapsed / args.steps)

    # ---- per-step times (SURVEY.md 8d asks for the median): each step bracketed by its own synchronisation; the
    # frame-pipelined N>1 path finishes a frame's exchange one step later, so its steps are only meaningful in bulk ----
    step_ms = []
    for _ in range(max(3, min(args.steps, 20))):
        sync()
        ts = time.perf_counter()
        step()
        sync()
        step_ms.append(1e3 * (time.perf_counter() - ts))
    step_ms.sort()
    median_ms = step_ms[len(step_ms) // 2]

    # ---- PCIe legs the reference's stage timers include (src/StereoMatch.cpp:227-237), never part of `value` ----
    pcie = None
    if rank == 0 and args.shard_sim <= 1:
        sync()
        ts = time.perf_counter()
        de.setInputImages(l, r)              # H2D of the u8 pair (blocking)
        h2d = 1e3 * (time.perf_counter() - ts)
        step(); sync()                       # maps of this pair on the device again
        de.download_maps()                   # (first call allocates the library's page-locked bounce buffer)
        ts = time.perf_counter()
        de.download_maps()                   # D2H of the two u8 maps (blocking)
        d2h = 1e3 * (time.perf_counter() - ts)
        pcie = {"h2d_ms": round(h2d, 3), "d2h_ms": round(d2h, 3), "h2d_bytes": int(l.nbytes + r.nbytes), "d2h_bytes": 2 * W * H,
                "note": "u8 pair in, two u8 maps out; excluded from value"}

    # ---- per-kernel device time (hipEvents on the launch stream), separate pass -----------
    de.set_option(capi.PSM_OPT_PROFILE, 1)
    de.reset_kernel_times()
    prof_steps = max(2, min(args.steps, 5))
    for _ in range(prof_steps):
        step()
    sync()
    names = {capi.PSM_K_PREP: "prep", capi.PSM_K_CVC: "cvc", capi.PSM_K_GUIDE: "guidance", capi.PSM_K_CVF_F: "cvf_fused",
             capi.PSM_K_CVF_A: "cvf_a", capi.PSM_K_CVF_B: "cvf_b", capi.PSM_K_WTA: "wta",
             capi.PSM_K_MERGE: "merge", capi.PSM_K_FGF: "cvf_fgf", capi.PSM_K_LRC: "lr_check"}
    kern = {}
    for k, nm in names.items():
        tot, n = de.kernel_time_ms(k)
        if n:
            kern[nm] = {"avg_ms": tot / n, "launches_per_step": n / prof_steps}
    de.set_option(capi.PSM_OPT_PROFILE, 0)
    # voxels per launch of the filter kernel = the step's 2*W*H*Dloc over its launches per step: 1 (both volumes in one
    # launch), 2 (one launch per volume - or, from 112 local slices up, the two phases of the select form: every 5th slice
    # of both volumes through the minima planes, then the other slices of both volumes against the key plane; the two
    # are instantiations of the same kernel, so avg_launch_ms is their mean and alg bytes / launch the mean as well)
    lps = max(1, round(kern.get("cvf_fused", {}).get("launches_per_step", 2)))   # (other filter forms: one side per launch)
    vox_per_launch = 2.0 * W * (y1 - y0) * (d1 - d0) / lps                        # (this rank's rows and slices)
    if args.fgf:   # per launch pair (setup + model + smooth + apply of one side): cost read at 1/s^2, model planes, q write
        ALG_BYTES["cvf_fgf"] = 4.0 + 68.0 / (args.fgf * args.fgf)
    # the default fused kernel also builds the costs and runs the WTA over its slices ("select" mode): it is credited
    # with the whole staged pipeline's algorithmic bytes (CVC 4 + CVF 40 + WTA 4); --flags 8192 is the storing form (CVF only)
    select_mode = not args.fgf and args.variant == 0 and not (max(args.flags, 0) & (16 | 512 | 8192))
    if select_mode:
        ALG_BYTES["cvf_fused"] = ALG_BYTES["pipeline"]
    fl = max(args.flags, 0)
    two_phase = select_mode and not (fl & (2097152 | 524288 | 262144 | 65536 | 16384)) and ((d1 - d0) >= 112 or (fl & 1048576))
    dom = max(("cvf_fgf",) if args.fgf else ("cvf_fused", "cvf_a", "cvf_b"), key=lambda k: kern.get(k, {"avg_ms": 0})["avg_ms"])
    dom_ms = kern[dom]["avg_ms"]
    achieved = ALG_BYTES[dom] * vox_per_launch / (dom_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": ("k_cvf_pc (select mode: CVC+CVF+WTA fused" + (", two phases = two launches per step)" if two_phase else ")"))
                if (select_mode and dom == "cvf_fused") else "k_" + dom,
                "alg_bytes_per_voxel": ALG_BYTES[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": ALG_BYTES[dom] * vox_per_launch, "avg_launch_ms": round(dom_ms, 4),
                "note": "achieved/frac credit the fused kernel with the staged pipeline's algorithmic bytes (SURVEY.md 8d); "
                        "its physical HBM rate is traffic_GBs - the kernel is VALU-issue bound, not HBM bound",
                "pipeline_alg_GBs": round(ALG_BYTES["pipeline"] * value / 1e9, 1),
                "pipeline_frac": round(ALG_BYTES["pipeline"] * value / 1e9 / HBM_PEAK_GBS, 4)}
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            key = f"{args.config}:k_{dom}"
            if key in tr and world == 1:
                roofline["traffic"] = tr[key]    # HBM bytes per launch from rocprofv3 PMC passes (not measured in this run)
                roofline["traffic_source"] = tr.get("_source", "profiles/traffic.json (rocprofv3 --pmc passes of scripts/gpu_run.sh)")
                roofline["traffic_GBs"] = round(tr[key] / (dom_ms * 1e-3) / 1e9, 1)   # physical HBM rate of the kernel
                vi = tr.get(key + "_valu_insts")
                if vi:
                    # what bounds it: VALU issue.  256 CUs x 4 SIMDs, one wave-instruction per 4 cycles and SIMD, 2.4 GHz peak
                    # engine clock (MI355X_MICROARCH.md) -> fraction of the issue slots the launch filled
                    roofline["valu"] = {"wave_insts_per_launch": vi, "issue_slot_frac_at_2.4GHz": round(vi * 4.0 / (1024 * 2.4e9 * dom_ms * 1e-3), 4),
                                        "source": "SQ_INSTS_VALU, " + roofline["traffic_source"].replace("{rd,wr}", "sq")}
        except Exception:
            pass
    for nm, v in kern.items():
        if nm in ALG_BYTES:
            v["alg_GBs"] = round(ALG_BYTES[nm] * vox_per_launch / (v["avg_ms"] * 1e-3) / 1e9, 1)
        v["avg_ms"] = round(v["avg_ms"], 4)

    # ---- result check (outside the timed region): the maps the timed path left on the device ------------------
    # against an unsharded single-context run of the same pair on this GPU.  With N > 1 this covers the RCCL
    # exchange itself: rank 0's merged maps must equal the one-GPU maps bit for bit.
    verified = None
    if rank == 0 and args.shard_sim <= 1 and (use_dist or args.verify):
        got_l, got_r = (m.copy() for m in de.download_maps())
        de.set_option(capi.PSM_OPT_ASYNC, 0)
        with P.DispEst(l, r, D, 8, True, device=local_rank, dtype=dtype) as ref:
            if args.fgf:
                ref.setSubsampleRate(args.fgf)
            ref.CostConst_GPU()
            ref.CostFilter_FGF_GPU() if args.fgf else ref.CostFilter_GPU()
            ref.DispSelect_GPU()
            verified = bool(np.array_equal(ref.lDisMap, got_l) and np.array_equal(ref.rDisMap, got_r))
            if lrc:      # the validity masks of the timed path against those of the one-GPU run
                got_lv, got_rv = (m.copy() for m in de.download_valid())
                ref.LRCheck_GPU()
                verified = verified and bool(np.array_equal(ref.lValid, got_lv) and np.array_equal(ref.rValid, got_rv))
        if not verified:
            print("bench.py: MAPS OF THE TIMED PATH DIFFER FROM THE ONE-GPU RUN", file=sys.stderr)

    box = None
    if args.box_bench and not use_dist:
        de.set_option(capi.PSM_OPT_PROFILE, 1)
        de.reset_kernel_times()
        for _ in range(5):
            de.box8_volume(0, download=False)
        de.synchronize()
        tot, n = de.kernel_time_ms(capi.PSM_K_BOX)
        de.set_option(capi.PSM_OPT_PROFILE, 0)
        bms = tot / n
        box = {"avg_ms": round(bms, 4), "alg_GBs": round(8.0 * vox_per_launch / (bms * 1e-3) / 1e9, 1),
               "read_GBs": round(4.0 * vox_per_launch / (bms * 1e-3) / 1e9, 1),
               "read_frac_of_peak": round(4.0 * vox_per_launch / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline: the oracle, driven like the reference pthreads path ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import psm_oracle_py as O   # checker / baseline only - never on the GPU path
        cores = os.cpu_count() or 1
        threads = min(8, cores)                 # MAX_CPU_THREADS (include/ComFunc.h:52)
        sd = args.cpu_sample_d if args.cpu_sample_d > 0 else int(256.0 * (1920 * 1080) / (W * H))
        sd = max(2, min(sd, D))
        tcpu = time.perf_counter()
        res = (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, sd, threads=threads)
        tcpu = time.perf_counter() - tcpu
        stage_s = (res["cvc_ms"] + res["cvf_ms"] + res["dispsel_ms"]) * 1e-3
        cpu = {"value": round(2.0 * W * H * sd / stage_s, 1), "unit": "voxels/s", "cores": threads,
               "kind": "port", "host_cores": cores,
               "sample": f"same {W}x{H} pair, first {sd} of {D} disparities (2*W*H*{sd} voxels), "
                         f"{threads} pthreads in the reference's per-d block pattern; "
                         f"cvc {res['cvc_ms']:.0f} ms, cvf {res['cvf_ms']:.0f} ms, dispsel {res['dispsel_ms']:.0f} ms "
                         f"(wall {tcpu:.1f} s)"}
        # the same restatement on more host cores (SURVEY.md 8d asks for 8 threads and for the box's core count):
        # context only, the contract's cpu_baseline is the 8-thread figure above
        wide = min(64, cores)
        if args.cpu_wide and wide > threads:
            resw = (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, sd, threads=wide)
            sw = (resw["cvc_ms"] + resw["cvf_ms"] + resw["dispsel_ms"]) * 1e-3
            cpu["wide"] = {"value": round(2.0 * W * H * sd / sw, 1), "cores": wide}

    seed_stride = de.seed_stride()        # what the library's in-place tuner chose for this geometry (0: not in use)
    if rank == 0:
        out = {
            "metric": "cost-volume voxels/s (CVC+CVF+WTA)" if not args.fgf else f"cost-volume voxels/s (CVC+CVF_FGF s={args.fgf}+WTA)",
            "value": value, "unit": "voxels/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": desc, "W": W, "H": H, "D": D, "voxels_per_step": voxels_per_step,
                       "parallelism": "1 GPU" if world == 1 and not use_dist else
                                      (f"{world} row stripes of {rows_max} rows (all {D} slices each) + 1 RCCL all_gather of the map rows per frame"
                                       if rows_mode else f"D sharded over {world} ranks + 1 RCCL {args.exchange} of packed minima"),
                       "kernel_variant": args.variant, "shard_sim": args.shard_sim, "seed_stride": seed_stride, "lr_check_on_gpu": bool(lrc), "shard": (args.shard if (use_dist or args.shard_sim > 1) else None)},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern,
            "median_ms_per_step": round(median_ms, 4), "pcie": pcie,
        }
        if verified is not None:
            out["verified_vs_single_gpu"] = verified
        if box:
            out["box_filter_pass"] = box
        if json_fd is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))
            sys.stdout.flush()
    de.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
