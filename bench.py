#!/usr/bin/env python3
"""bench.py - cost-volume voxels/s (CVC+CVF+WTA) of the HIP DispEst hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic stereo pair already resident in HBM:
CostConst (gray/gradient + both cost volumes) -> CostFilter (guidance precompute + guided filter
of both volumes) -> DispSel (WTA).  Workload at N=1: BASELINE.json configs[3], 1920x1080, D=256, float32 - the
configuration the metric is quoted on.  For N > 1 the same job is sharded over the ranks (total work fixed -> "scaling":
"strong") and ONE invocation times both sharding axes:
  value      - row stripes: rank g owns H/N output rows of both maps, all D slices; no minima leave the rank, the one
               exchange per frame is an RCCL all-gather of the finished map rows (0.5 MB per rank at 1080p / 8)
  alt_shard  - the configuration BASELINE configs[3] / the north star name literally: D/N slices per rank, one RCCL
               all-gather of the packed per-pixel minima per frame (33 MB per rank), device-side minimum
Both produce the same maps bit for bit (each record carries verified_vs_single_gpu and oracle_maps_equal); the row axis is
the headline because it is the faster way to run the same job (no 33 MB collective, two-phase selection at full efficiency).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - dominant kernel (k_cvf_pc): algorithmic HBM bytes per launch / its mean duration over the launches OF THE
                 TIMED REGION (time stamps the kernel takes itself, PSM_OPT_PROFILE 2 - no events, no separate pass)
  cpu_baseline - the CPU oracle (restatement of the reference pthreads path) timed on a bounded
                 sample on this box's host cores (rank 0, N=1 only); a reported baseline
  oracle_maps_equal - the maps the TIMED path left on the device == the oracle's maps of the same pair (whole D)
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
# algorithmic HBM bytes per voxel of the staged pipeline (SURVEY.md 8d / DESIGN.md): CVC 4 W; CVF stage A 4 R + 16 W, stage B
# 16 R + 4 W; WTA 4 R.  8-bit mode: CVC 1 W; CVF 1 R + 16 W, 16 R + 1 W; WTA 1 R.
ALG = {"f32": {"cvf_fused": 40.0, "cvf_a": 20.0, "cvc": 4.0, "wta": 4.0, "box8": 8.0, "pipeline": 48.0},
       "u8": {"cvf_fused": 34.0, "cvf_a": 17.0, "cvc": 1.0, "wta": 1.0, "box8": 8.0, "pipeline": 36.0}}
FORM_NAME = {0: "store", 1: "planes", 2: "keys"}


def spawn_ranks(n, args):
    """Re-run this command as n ranks (one per GPU) under torch.distributed.run; returns its exit code.
    --same-device (all ranks on GPU 0: the N > 1 protocol on a one-GPU box) with --backend auto: RCCL is tried first; if it
    refuses the duplicate device (or does not come up within two minutes) the run is repeated with the exchange staged through
    host memory over gloo (primestereomatch_amd/exchange.py) - the line then says which transport carried it and why."""
    import signal
    import socket
    import subprocess

    def run(extra, timeout=None):
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
        env.setdefault("OMP_NUM_THREADS", "8")
        for attempt in range(3):
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
            p = subprocess.Popen(cmd, env=env, start_new_session=True, stderr=subprocess.PIPE, text=True)
            try:
                _, err = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)           # (our own process group: the launcher and its ranks)
                p.wait()
                return 124, "no communicator within %d s" % timeout
            # the port was free when we asked and taken when the rendezvous store tried to listen (seen once in a round's sessions:
            # the line of that run was lost): nothing has run yet - ask for another port
            if p.returncode != 0 and "EADDRINUSE" in (err or "") and attempt < 2:
                print("bench.py: rendezvous port %d was taken before the launcher could listen on it; retrying with another" % port, file=sys.stderr)
                continue
            if not timeout and err:
                sys.stderr.write(err)                      # (a measurement run: its ranks' messages belong on our stderr)
            return p.returncode, err

    if not args.same_device:
        # one rank per GPU: more ranks than devices cannot work - say so before anything is launched
        try:
            from primestereomatch_amd import capi
            ndev = capi.device_count()
        except Exception as e:      # (no library / no runtime: the ranks will say why)
            ndev = None
            print(f"bench.py: could not count the devices ({type(e).__name__}: {e}); launching anyway", file=sys.stderr)
        if ndev is not None and ndev < n:
            print(f"bench.py: --gpus {n} needs {n} devices, this node shows {ndev} (one rank per GPU; --same-device runs the N > 1 "
                  "protocol on one GPU as a correctness check); nothing measured", file=sys.stderr)
            return 3
    if args.backend in ("auto", "nccl") and not args.nccl_probe:
        # Does an RCCL communicator of these ranks come up at all?  One short probe run first (init + one collective, two minutes
        # at most), so that a transport problem is ONE clear line instead of N tracebacks in the middle of a measurement.
        rc, err = run(["--backend", "nccl", "--nccl-probe"], timeout=120)
        if rc == 0:
            return run(["--backend", "nccl"])[0]
        lines = [ln.strip() for ln in (err or "").splitlines() if ln.strip()]
        why = "rc %d" % rc
        for pat in ("bench.py: RCCL", "uplicate GPU", "ncclInvalid", "NCCL error", "NCCL WARN", "DistBackendError", "HIP error", "Error"):
            hit = [ln for ln in lines if pat in ln and "traceback" not in ln]
            if hit:
                why = hit[-1]
                break
        if args.same_device and args.backend == "auto":
            print("bench.py: RCCL with %d ranks on one device failed (%s) - staging the exchange over gloo" % (n, why[:200]), file=sys.stderr)
            return run(["--backend", "gloo", "--backend-note", "RCCL refused %d ranks on one device: %s" % (n, why[:200])])[0]
        print("bench.py: no RCCL communicator of %d ranks on this node (%s); nothing measured.  (--backend gloo stages the exchange "
              "through host memory: a correctness run, not a transport)" % (n, why[:300]), file=sys.stderr)
        return 3
    return run([])[0]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="", choices=["", "f32", "u8"], help="volume element type (default: f32, u8 for the c1 configs)")
    ap.add_argument("--seg-rows", type=int, default=-1, help="marching-kernel y segment (-1: library default)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--flags", type=int, default=-1, help="PSM_OPT_FLAGS bits (-1: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true", help="N > 1: skip the oracle run that checks the maps (N = 1: it is the cpu_baseline run)")
    ap.add_argument("--cpu-sample-d", type=int, default=0,
                    help="disparities in the CPU-baseline sample (0 = auto: the whole D at 1080p and below (~10 s on 8 threads), "
                         "proportionally fewer for larger images)")
    ap.add_argument("--no-cpu-wide", action="store_true",
                    help="skip the second CPU-baseline run on min(64, host cores) threads (SURVEY.md 8d asks for 8 threads and for the "
                         "box's core count; the second run adds a few seconds)")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even for N=1")
    ap.add_argument("--exchange", default="", choices=["", "allreduce", "allgather", "none"],
                    help="--shard disp, N>1: the one exchange step - all_gather of the packed keys ((N-1)x33 MB per rank at 1080p; "
                         "the north star's wording, default of the alt_shard record) or one all_reduce(MIN) (~2x33 MB whatever N; "
                         "default when --shard disp is the headline)")
    ap.add_argument("--shard", default="rows", choices=["rows", "disp"],
                    help="N > 1: what a rank owns in the headline measurement - 'rows': a stripe of H/N output rows of both volumes, "
                         "all D slices; 'disp': D/N slices of both volumes, whole image.  The other axis is timed as alt_shard")
    ap.add_argument("--strided", action="store_true",
                    help="--shard disp (and the alt_shard record of a rows run): rank g owns the slices d = g (mod N) instead of the contiguous "
                         "range [D g / N, D (g + 1) / N) - psm_create_shard_strided; its slices span the whole disparity range, so the two-phase "
                         "selection's seeds bound every pixel")
    ap.add_argument("--no-alt-shard", action="store_true", help="N > 1: do not time the other sharding axis")
    ap.add_argument("--lr-check", type=int, default=-1, choices=[-1, 0, 1],
                    help="PP left-right check on the GPU inside the step (BASELINE configs[4]); -1: on for config c5 only")
    ap.add_argument("--pp", action="store_true", help="also time the post-processing stages (lrCheck, fillInv, wgtMedian: PP::processDM) "
                                                      "on the finished maps and verify them against the oracle; never part of value")
    ap.add_argument("--frame-loop", type=int, default=20, help="N=1: frames of the PCIe-inclusive frame loop (0: skip)")
    ap.add_argument("--box-bench", action="store_true", help="also time the plain box-filter pass")
    ap.add_argument("--fgf", type=int, default=0, choices=[0, 2, 4, 8],
                    help="diagnostic: aggregate with the Fast Guided Filter variant (CostFilter_FGF) at this subsample "
                         "rate instead of the full guided filter; not the north-star metric")
    ap.add_argument("--no-frame-pipeline", action="store_true",
                    help="N>1: finish the exchange + merge of a frame inside its own step instead of one step later "
                         "(default: a frame's collective overlaps the next frame's filter; two buffers)")
    ap.add_argument("--verify", action="store_true",
                    help="N=1: also compare the maps of the timed path with a fresh single-context run (always done for N>1)")
    ap.add_argument("--batch", type=int, default=1,
                    help="N = 1: B different stereo pairs of the configuration per step through ONE set of launches (psm_compute_batch: the "
                         "reference's loop over pairs / datasets, src/main.cpp:64-73); value counts the voxels of all B pairs")
    ap.add_argument("--same-device", action="store_true",
                    help="N > 1: every rank uses GPU 0 - the N > 1 protocol (two buffers, pending exchange, gather / merge at world N) "
                         "on a box with one GPU; a correctness run, not a scaling measurement")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="transport of the one exchange per frame: nccl = RCCL on device tensors (default); gloo = the same collective "
                         "staged through page-locked host tensors (for --same-device when RCCL refuses a duplicate device)")
    ap.add_argument("--nccl-probe", action="store_true", help=argparse.SUPPRESS)     # internal: init RCCL, one collective, exit
    ap.add_argument("--backend-note", default="", help=argparse.SUPPRESS)
    ap.add_argument("--pair", default="synthetic", choices=["synthetic", "fixture"],
                    help="c1 / c1x / c2: 'fixture' times the Middlebury pair BASELINE.json names instead of a synthetic pair of its size - Cones "
                         "(c1; c1x: its 384 x 288 crop) and Teddy (c2) from tests/golden/*_pair.npz, the images the reference ships")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="(0 = default: 1 at N = 1; 2 per rank for row stripes from N = 4 and for disparity shards from N = 2)  F contexts of the configuration, each on its own stream, take the steps in turn - frame i + 1 is queued while "
                         "frame i runs, so a frame's short kernels (prep, guidance, reduction) and the half-empty last round of its fused "
                         "launch run beside the next frame's fused kernel.  The reference's use is a frame loop (src/main.cpp:64-73); this is "
                         "that loop with two frames in the device's queues.  Measured: -9 % at 720p x 128, -19 % at 450 x 375 x 64, nothing at "
                         "1080p x 256 (profiles/r05/exp_frames_in_flight.txt); a rank's 1/8 share of 1080p x 256: -6 %")
    ap.add_argument("--shard-sim", type=int, default=0,
                    help="diagnostic: time only rank 0's share of a G-rank job on this GPU (no exchange); "
                         "the JSON line is then NOT the headline metric")
    return ap.parse_args()


def main():
    args = parse_args()
    # (ranks launched by a driver's own torch.distributed.run come here directly: the host driver only supports dmabuf IPC, and
    # RCCL's device-memory sharing fails with the legacy mode - hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    N = args.gpus
    if N < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    Wc, Hc, Dc, _ = CONFIGS[args.config]
    if args.shard == "disp" and N > Dc:
        raise SystemExit(f"bench.py: --gpus {N} exceeds the {Dc} disparity slices of config {args.config} (one shard per rank)")
    if args.shard == "rows" and (N > Hc or (N - 1) * -(-Hc // N) >= Hc):
        # (every rank decides this the same way, before any rendezvous: stripes of ceil(H / N) rows must leave the last rank some)
        raise SystemExit(f"bench.py: --gpus {N}: stripes of {-(-Hc // N)} rows leave a rank without rows of the "
                         f"{Hc}-row image of config {args.config} (one stripe per rank)")
    if N > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher - one rank per GPU under torch.distributed.run on
        # 127.0.0.1 - and pass rank 0's single JSON line through
        return spawn_ranks(N, args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = (N > 1) or args.force_dist
    torch = dist = ex = None
    json_fd = None
    dev_index = 0 if args.same_device else local_rank
    xname = "RCCL" if args.backend in ("auto", "nccl") else "gloo (host-staged)"
    backend = "nccl" if args.backend == "auto" else args.backend
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
        if dev_index >= torch.cuda.device_count():
            print(f"bench.py: rank {rank} wants device {dev_index}, this node shows {torch.cuda.device_count()} (one rank per GPU)", file=sys.stderr)
            os._exit(3)
        torch.cuda.set_device(dev_index)
        from primestereomatch_amd.exchange import Exchange
        try:
            import datetime
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index),
                                        timeout=datetime.timedelta(seconds=900))      # (rank 0 checks maps against the CPU oracle between collectives: seconds at 1080p, a minute at 4K)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            ex = Exchange(torch, dist, backend)
            # the transport's first collective (communicator set-up happens here at the latest): outside every timed region
            t = torch.ones(4, dtype=torch.int64, device="cuda")
            ex.all_reduce_min(t)
            torch.cuda.synchronize()
        except Exception as e:      # one clear line per rank instead of a traceback in the middle of a measurement
            msg = " ".join(str(e).split())[:400]
            print(f"bench.py: {xname} communicator of {world} ranks failed on rank {rank} (device {dev_index}): {type(e).__name__}: {msg}", file=sys.stderr)
            os._exit(3)
        if args.nccl_probe:
            dist.destroy_process_group()
            return 0

    import numpy as np
    import primestereomatch_amd as P
    from primestereomatch_amd import capi, stripes, synth
    from primestereomatch_amd.dispest import compute_batch, share_streams

    W, H, D, desc = CONFIGS[args.config]
    dtype = args.dtype or ("u8" if args.config.startswith("c1") else "f32")
    alg = ALG[dtype]
    if dtype == "u8" and not args.config.startswith("c1"):
        desc = desc.replace("float32", "8-bit char mode")
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    data_note = "synthetic"
    if args.pair == "fixture":
        if args.config not in ("c1", "c1x", "c2"):
            raise SystemExit("bench.py: --pair fixture exists for the Middlebury-size configs c1, c1x, c2")
        which = "teddy" if args.config == "c2" else "cones"
        g_ = np.load(os.path.join(ROOT, "tests", "golden", which + "_pair.npz"))
        l, r = np.ascontiguousarray(g_["l_bgr"][:H, :W]), np.ascontiguousarray(g_["r_bgr"][:H, :W])
        assert l.shape == (H, W, 3), l.shape
        data_note = f"Middlebury {which.capitalize()} (tests/golden/{which}_pair.npz" + (", top-left 384 x 288 crop)" if args.config == "c1x" else ")")
        desc = desc.replace("synthetic", f"Middlebury {which.capitalize()}")
    use_batch = (args.batch > 1 or args.batch == -1) and N == 1 and not args.force_dist and args.shard_sim <= 1 and not args.fgf
    B = abs(args.batch) if use_batch else 1      # (--batch -1: ONE pair through the batch entry - one call instead of three)
    batch_pairs = [(l, r)] + [synth.make_pair(W, H, D, seed=b)[:2] for b in range(1, B)]    # B different pairs
    voxels_per_step = 2.0 * W * H * D * B       # both volumes, all ranks, all pairs of a batch
    # BASELINE configs[4]: "+ PP left-right check on-GPU" - lrCheck (src/PP.cpp:17-50) on the finished maps, part of the step
    lrc = (args.lr_check == 1 or (args.lr_check < 0 and args.config == "c5")) and not (args.shard_sim > 1)
    side_stream = None
    if use_dist:
        # one non-default torch stream carries both our kernels and the RCCL collective, so the
        # exchange is ordered against the kernels without host synchronisation
        side_stream = torch.cuda.Stream()
        torch.cuda.set_stream(side_stream)

    def single_gpu_maps():
        """The same pair through a fresh unsharded context on this GPU: what every sharded run must reproduce bit for bit."""
        with P.DispEst(l, r, D, 8, True, device=dev_index, dtype=dtype) as ref:
            if args.fgf:
                ref.setSubsampleRate(args.fgf)
            ref.CostConst_GPU()
            ref.CostFilter_FGF_GPU() if args.fgf else ref.CostFilter_GPU()
            ref.DispSelect_GPU()
            out = [ref.lDisMap.copy(), ref.rDisMap.copy()]
            if lrc:
                ref.LRCheck_GPU()
                out += [ref.lValid.copy(), ref.rValid.copy()]
        return out

    def measure(shard, exchange):
        """One timed measurement of the step under a sharding axis; returns the record (with the live context in it)."""
        rows_mode = shard == "rows" and not args.fgf and (use_dist or args.shard_sim > 1)
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
        strided = None
        if args.strided and not rows_mode and (use_dist or args.shard_sim > 1):
            strided = (rank, world) if use_dist else (0, args.shard_sim)
            d0, d1 = strided[0], D

        def new_ctx(pl_=l, pr_=r, dr=(d0, d1)):
            o_ = P.DispEst(pl_, pr_, D, 8, True, device=dev_index, d_range=dr, dtype=dtype, d_stride=(strided if dr == (d0, d1) else None))
            if args.seg_rows >= 0:
                o_.set_option(capi.PSM_OPT_SEG_ROWS, args.seg_rows)
            o_.set_option(capi.PSM_OPT_KERNEL_VARIANT, args.variant)
            if args.flags >= 0:
                o_.set_option(capi.PSM_OPT_FLAGS, args.flags)
            o_.set_option(capi.PSM_OPT_ASYNC, 1)
            return o_

        de = new_ctx()
        batch_all = [de] + [new_ctx(pl_, pr_, (0, D)) for pl_, pr_ in batch_pairs[1:]]
        if len(batch_all) > 1:
            share_streams(batch_all)         # one compute stream and one copy stream each way for the whole batch
        if rows_mode:
            de.set_rows(y0, y1)
        if args.fgf:
            de.setSubsampleRate(args.fgf)
        # --frames-in-flight F: F - 1 more contexts with the same pair, geometry and options, each on its own stream; step i runs on
        # context i % F.  Default (0): one frame at a time at N = 1 (the headline configuration does not gain), TWO per rank where a
        # rank's share of the job is a short launch chain whose tails and small kernels hide under the next frame's fused kernel:
        # row stripes from N = 4 (rank-local, one GPU, 1080p x 256: 1/2 stripe 3.40 -> 3.51 ms, 1/4 1.84 -> 1.78, 1/8 1.01 -> 0.92),
        # disparity shards from N = 2 (1/2: 3.74 -> 3.57, 1/4: 2.09 -> 2.06); profiles/r06/exp_frames_in_flight_by_share.txt, DESIGN.md 6.
        FIF = args.frames_in_flight if args.frames_in_flight > 0 else (2 if use_dist and (world >= 4 or not rows_mode) else 1)
        if use_batch or args.fgf:
            FIF = 1
        ring = [de]
        for _ in range(FIF - 1):
            o_ = new_ctx()
            if rows_mode:
                o_.set_rows(y0, y1)
            ring.append(o_)
        if FIF > 1:
            for o_ in ring:                  # what FrameRing tells its contexts: the planner cuts the launches for FIF pairs at a time
                o_.set_option(capi.PSM_OPT_FRAMES_IN_FLIGHT, FIF)
        # ---- one slot per frame in flight: its context, (N > 1) its torch stream - kernels AND the frame's collective are ordered
        # on it, so two slots overlap freely - and its own exchange buffers (two key / map tensors alternate within a slot: a frame's
        # exchange is still pending when the slot's next frame writes) ----
        import contextlib

        class Slot:
            pass
        slots = []
        HW2 = 2 * H * W
        for k_, o_ in enumerate(ring):
            sl = Slot()
            sl.de, sl.pending, sl.n, sl.stream = o_, [], 0, None
            if use_dist:
                sl.stream = side_stream if k_ == 0 else torch.cuda.Stream()
                with torch.cuda.stream(sl.stream):
                    sl.kbuf = [torch.empty(HW2, dtype=torch.int64, device="cuda") for _ in range(2)]
                    sl.keys_all = torch.empty(world * HW2 if exchange == "allgather" and not rows_mode else 1, dtype=torch.int64, device="cuda")
                    if rows_mode:
                        # the maps of a frame are written by the library straight into one of two torch tensors (psm_set_map_buffer);
                        # the stripe rows go through one all_gather per frame: [world][2][rows_max][W] uint8 = the two whole maps
                        sl.mbuf = [torch.zeros(HW2 + 4, dtype=torch.uint8, device="cuda") for _ in range(2)]
                        sl.send = torch.zeros(2 * rows_max * W, dtype=torch.uint8, device="cuda")
                        sl.recv = torch.zeros(world * 2 * rows_max * W, dtype=torch.uint8, device="cuda")
                o_.set_stream(sl.stream.cuda_stream)
            slots.append(sl)
        if use_dist:
            torch.cuda.synchronize()
        frame = [0]
        pipelined = use_dist and (exchange == "allreduce" or rows_mode) and not args.no_frame_pipeline and not args.fgf

        def on(sl):          # everything torch issues for a slot's frame (packing, the collective, waits, assembling) goes to ITS stream
            return torch.cuda.stream(sl.stream) if use_dist else contextlib.nullcontext()

        def finish_pending(sl):
            # exchange of an earlier frame of this slot -> final maps (the collective ran on RCCL's stream meanwhile)
            while sl.pending:
                works, kb = sl.pending.pop(0)
                for w_ in works:
                    w_.wait()
                if rows_mode:
                    stripes.assemble(sl.recv, world, H, W, rows_max, kb)    # [rank][side][row][x] -> [side][y][x]
                    sl.de.set_map_buffer(kb.data_ptr(), whole=True)
                else:
                    sl.de.DispSelect_merge(kb.data_ptr(), 1, download=False)
                if lrc:
                    sl.de.LRCheck_device()

        def stripe_exchange(sl, mb, async_op):
            stripes.pack_stripe(mb, y0, y1, sl.send, H, W, rows_max)
            return ex.all_gather(sl.recv, sl.send, async_op=async_op)      # the one exchange step (RCCL; or staged over gloo)

        def step():
            if use_batch:                    # all B pairs: one prep, one guidance, one fused grid (blockIdx.z = pair), one reduction
                compute_batch(batch_all)
                return
            sl = slots[frame[0] % FIF]       # frames in flight: this step's frame goes to the next slot of the ring (its own stream)
            frame[0] += 1
            with on(sl):
                step_on(sl)

        def step_on(sl):
            cur = sl.de
            cur.CostConst_GPU()
            if args.fgf:
                cur.CostFilter_FGF_GPU()
                if use_dist:
                    cur.DispSelect_partial(sl.kbuf[0].data_ptr())
                    ex.all_reduce_min(sl.kbuf[0])
                    cur.DispSelect_merge(sl.kbuf[0].data_ptr(), 1, download=False)
                elif args.shard_sim > 1:
                    cur.DispSelect_partial()
                else:
                    cur.DispSelect_device()
                return
            if rows_mode and use_dist:
                # Row stripes: nothing of the cost volumes or their minima leaves the rank; the finished rows of both maps are
                # gathered - asynchronously, behind the next frame's filter when pipelined (maps alternate between two tensors).
                mb = sl.mbuf[sl.n & 1]
                sl.n += 1
                cur.set_map_buffer(mb.data_ptr())
                cur.CostFilter_GPU()
                cur.DispSelect_device()
                finish_pending(sl)
                if pipelined:
                    sl.pending.append(((stripe_exchange(sl, mb, True),), mb))
                else:
                    stripe_exchange(sl, mb, False)
                    sl.pending.append(((), mb))
                    finish_pending(sl)
                return
            if pipelined:
                # Frame pipeline, ONE collective per frame: the fused kernel leaves the packed minima of both volumes directly
                # in this frame's key tensor; the all-reduce(MIN) over the ranks is asynchronous and has the whole next
                # frame's filter to complete - its merge is issued after that filter, so no kernel of ours ever waits for a
                # collective that is still running.  Keys alternate between two tensors.
                kb = sl.kbuf[sl.n & 1]
                sl.n += 1
                cur.set_key_buffer(kb.data_ptr())
                cur.CostFilter_GPU()
                finish_pending(sl)
                w_ = ex.all_reduce_min(kb, async_op=True)
                sl.pending.append(((w_,), kb))
                return
            if use_dist:
                cur.set_key_buffer(sl.kbuf[0].data_ptr())
            cur.CostFilter_GPU()
            if use_dist:
                if exchange == "none":       # diagnostic only: cost of the torch collective call itself
                    cur.DispSelect_merge(sl.kbuf[0].data_ptr(), 1, download=False)
                elif exchange == "allreduce":
                    ex.all_reduce_min(sl.kbuf[0])                         # the one exchange step (RCCL)
                    cur.DispSelect_merge(sl.kbuf[0].data_ptr(), 1, download=False)
                else:
                    ex.all_gather(sl.keys_all, sl.kbuf[0])                # the one exchange step (RCCL)
                    cur.DispSelect_merge(sl.keys_all.data_ptr(), world, download=False)
            elif args.shard_sim > 1 and not rows_mode:
                cur.DispSelect_partial()
            else:
                cur.DispSelect_device()
            if lrc and not (use_dist and exchange == "none"):
                cur.LRCheck_device()

        def sync():
            for sl in slots:
                with on(sl):
                    finish_pending(sl)
            if use_dist:
                torch.cuda.synchronize()
            for o_ in ring:
                o_.synchronize()

        def barrier():
            if use_dist:
                ex.barrier()

        for _ in range(args.warmup):
            step()
        # the fused filter kernel stamps its own start / end from here on: two atomics per workgroup, no events between the
        # kernels - the launches of the timed region itself are what roofline reports
        for o_ in ring:
            o_.set_option(capi.PSM_OPT_PROFILE, 2)
        sync()
        for o_ in ring:
            o_.filter_launch_times()
        sync(); barrier(); sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync(); barrier(); sync()
        elapsed = time.perf_counter() - t0
        launch_times = [lt for o_ in ring for lt in o_.filter_launch_times()]
        for o_ in ring:
            o_.set_option(capi.PSM_OPT_PROFILE, 0)
        if use_dist:
            elapsed = ex.max_float(elapsed)
        rec = {"shard": shard if (use_dist or args.shard_sim > 1) else None,
               "exchange": (("all_gather of map rows" if rows_mode else exchange) if use_dist else None),
               "ms_per_step": 1e3 * elapsed / args.steps, "value": voxels_per_step / (elapsed / args.steps)}
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
        rec["median_ms_per_step"] = round(step_ms[len(step_ms) // 2], 4)
        # ---- the launches of the fused kernel inside the timed region, by form ----
        by_form = {}
        for ms, f in launch_times:
            by_form.setdefault(f, []).append(ms)
        rec["filter_launches"] = {FORM_NAME.get(f, str(f)): {"avg_ms": round(sum(v) / len(v), 4), "min_ms": round(min(v), 4),
                                                             "max_ms": round(max(v), 4), "per_step": len(v) / args.steps}
                                  for f, v in sorted(by_form.items())}
        rec["filter_ms_per_step"] = sum(ms for ms, _ in launch_times) / args.steps if launch_times else None
        # ---- N > 1: what a step is made of on every rank - the rank-local kernels alone and the frame's ONE collective alone (each
        # bracketed by its own synchronisation: their sum exceeds a pipelined step, which overlaps the two) ----
        if use_dist and not args.fgf:
            s0 = slots[0]

            def local_only():
                de.CostConst_GPU()
                if rows_mode:
                    de.set_map_buffer(s0.mbuf[0].data_ptr())
                    de.CostFilter_GPU()
                    de.DispSelect_device()
                else:
                    de.set_key_buffer(s0.kbuf[0].data_ptr())
                    de.CostFilter_GPU()

            def exchange_only():
                if rows_mode:
                    stripe_exchange(s0, s0.mbuf[0], False)
                    stripes.assemble(s0.recv, world, H, W, rows_max, s0.mbuf[0])
                elif exchange == "allgather":
                    ex.all_gather(s0.keys_all, s0.kbuf[0])
                    de.DispSelect_merge(s0.keys_all.data_ptr(), world, download=False)
                elif exchange == "allreduce":
                    ex.all_reduce_min(s0.kbuf[0])
                    de.DispSelect_merge(s0.kbuf[0].data_ptr(), 1, download=False)

            comp, coll = [], []
            for _ in range(5):
                sync(); barrier()
                ts = time.perf_counter()
                local_only()
                sync()
                comp.append(1e3 * (time.perf_counter() - ts))
                barrier()
                ts = time.perf_counter()
                exchange_only()
                sync()
                coll.append(1e3 * (time.perf_counter() - ts))
            comp.sort(); coll.sort()
            per = ex.gather_floats([comp[len(comp) // 2], coll[len(coll) // 2]])
            rec["per_rank"] = {"compute_ms": [round(v[0], 4) for v in per], "collective_ms": [round(v[1], 4) for v in per],
                               "note": "medians of 5: the rank-local kernels of one frame (prep, guidance, fused filter, reduction) and the "
                                       "frame's one exchange (+ gather / merge of its result), each alone between synchronisations; a "
                                       "pipelined step overlaps the exchange of frame i with the filter of frame i + 1"}
            for _ in ring:
                step()
            sync()                               # (leave every context as a complete step does: final maps of this pair)
        nsl = len(range(strided[0], D, strided[1])) if strided else d1 - d0
        rec["geometry"] = {"rows": [y0, y1], "slices": [d0, d1], "n_slices": nsl, "strided": list(strided) if strided else None,
                           "rows_max": rows_max, "parts": parts, "rows_mode": rows_mode}
        rec["sync"], rec["step"], rec["de"], rec["batch_all"] = sync, step, de, batch_all + ring[1:]
        rec["ring"] = ring
        return rec

    def check_maps(rec, ref_maps, oracle_maps):
        """The maps the timed path left on the device against the one-GPU run and against the oracle (rank 0)."""
        de = rec["de"]
        got = [m.copy() for m in de.download_maps()]
        if lrc:
            got += [m.copy() for m in de.download_valid()]
        out = {}
        if ref_maps is not None:
            out["verified_vs_single_gpu"] = bool(all(np.array_equal(a, b) for a, b in zip(ref_maps, got)))
            if not out["verified_vs_single_gpu"]:
                print("bench.py: MAPS OF THE TIMED PATH DIFFER FROM THE ONE-GPU RUN", file=sys.stderr)
        if oracle_maps is not None:
            nl, nr = int(np.count_nonzero(got[0] != oracle_maps[0])), int(np.count_nonzero(got[1] != oracle_maps[1]))
            out["oracle_maps_equal"] = nl == 0 and nr == 0
            out["oracle_map_mismatches"] = [nl, nr]
            if nl or nr:
                print(f"bench.py: MAPS OF THE TIMED PATH DIFFER FROM THE ORACLE'S ({nl} + {nr} pixels)", file=sys.stderr)
        return out

    # ================= headline measurement =================
    exchange = args.exchange or ("allreduce" if args.shard == "disp" else "allgather")
    head = measure(args.shard, exchange)
    de, sync, step, batch_all = head["de"], head["sync"], head["step"], head["batch_all"]

    def step_all():      # one step on every context of the frame ring (one step when there is no ring): every context holds current maps
        for _ in head["ring"]:
            step()
    geo = head["geometry"]
    (y0, y1), (d0, d1), rows_mode, n_slices = geo["rows"], geo["slices"], geo["rows_mode"], geo["n_slices"]
    ms_per_step, value = head["ms_per_step"], head["value"]

    # ---- the maps the timed region left on the device (before anything else runs on this context) ----
    timed_maps = None
    if rank == 0 and args.shard_sim <= 1:
        timed_maps = [m.copy() for m in de.download_maps()]

    # ---- PCIe legs the reference's stage timers include (src/StereoMatch.cpp:227-237), never part of `value` ----
    # (N > 1: step() contains the frame's collective - EVERY rank runs the same sequence of steps here and below; only the
    # measuring and the downloads are rank 0's)
    pcie = None
    if args.shard_sim <= 1:
        sync()
        ts = time.perf_counter()
        de.setInputImages(l, r)              # H2D of the u8 pair (blocking)
        h2d = 1e3 * (time.perf_counter() - ts)
        step_all(); sync()                       # maps of this pair on the device again
    if rank == 0 and args.shard_sim <= 1:
        de.download_maps()                   # (first call allocates the library's page-locked bounce buffer)
        ts = time.perf_counter()
        de.download_maps()                   # D2H of the two u8 maps (blocking)
        d2h = 1e3 * (time.perf_counter() - ts)
        pcie = {"h2d_ms": round(h2d, 3), "d2h_ms": round(d2h, 3), "h2d_bytes": int(l.nbytes + r.nbytes), "d2h_bytes": 2 * W * H,
                "note": "u8 pair in, two u8 maps out; excluded from value"}
        if use_batch and args.frame_loop > 0:
            # the batch as a frame loop: every frame's B pairs travel while the previous batch computes, its 2 B maps return
            # while the next one computes (per-context psm_upload_pair_async / psm_download_maps_async around psm_compute_batch)
            nf = args.frame_loop

            def bframe(i, last):
                compute_batch(batch_all)               # adopts the pairs staged during the previous frame
                for o_, (pl_, pr_) in zip(batch_all, batch_pairs):
                    if not last:
                        o_.setInputImages_async(pl_, pr_)
                for o_ in batch_all:
                    if i > 0:
                        o_.download_maps_wait()
                    o_.download_maps_async()

            for i in range(3):
                bframe(i, False)
            for o_ in batch_all:
                o_.download_maps_wait()
            sync()
            compute_batch(batch_all); sync()
            ts = time.perf_counter()
            for i in range(nf):
                bframe(i, i + 1 == nf)
            bl = [tuple(m.copy() for m in o_.download_maps_wait()) for o_ in batch_all]
            sync()
            loop_ms = 1e3 * (time.perf_counter() - ts) / nf
            pcie["frame_loop"] = {"frames": nf, "pairs_per_frame": B, "ms_per_frame": round(loop_ms, 4), "over_step_ms": round(loop_ms - ms_per_step, 4),
                                  "maps_equal_timed_path": bool(np.array_equal(bl[0][0], timed_maps[0]) and np.array_equal(bl[0][1], timed_maps[1])),
                                  "note": "H2D of every frame's B pairs + D2H of its 2 B maps inside the loop, overlapped with the kernels"}
            step_all(); sync()
        elif not use_dist and args.frame_loop > 0 and not args.fgf:
            # the reference's use is a frame loop (src/main.cpp:64-73) whose stage timers include the copies: pair i+1
            # travels (psm_upload_pair_async) and the maps of frame i-1 return (psm_download_maps_async) while frame i computes
            nf = args.frame_loop

            def frame(i, last):
                de.CostConst_GPU()                     # adopts the pair staged during the previous frame
                if not last:
                    de.setInputImages_async(l, r)      # next frame's pair: staged + H2D on the copy stream
                de.CostFilter_GPU()
                de.DispSelect_device()
                if lrc:
                    de.LRCheck_device()
                if i > 0:
                    de.download_maps_wait()            # the previous frame's maps have arrived
                de.download_maps_async()

            de.setInputImages(l, r)
            for i in range(3):
                frame(i, False)
            de.download_maps_wait(); sync()
            de.CostConst_GPU(); sync()                 # (consume the staged pair: the timed loop starts with a resident one)
            ts = time.perf_counter()
            for i in range(nf):
                frame(i, i + 1 == nf)
            lm, rm = (m.copy() for m in de.download_maps_wait())
            sync()
            loop_ms = 1e3 * (time.perf_counter() - ts) / nf
            pcie["frame_loop"] = {"frames": nf, "ms_per_frame": round(loop_ms, 4), "over_step_ms": round(loop_ms - ms_per_step, 4),
                                  "maps_equal_timed_path": bool(np.array_equal(lm, timed_maps[0]) and np.array_equal(rm, timed_maps[1])),
                                  "note": "H2D of every frame's pair + D2H of every frame's maps inside the loop, overlapped with the "
                                          "kernels (psm_upload_pair_async / psm_download_maps_async); unpipelined it would be "
                                          f"{ms_per_step + h2d + d2h:.3f} ms"}
            step_all(); sync()

    # ---- per-kernel device time of the OTHER kernels (hipEvents on the launch stream), separate pass: it perturbs the step
    # by a few %, so the fused filter's times are NOT taken from it (filter_launches are the timed region's own) ----
    ring = head["ring"]
    for o_ in ring:
        o_.set_option(capi.PSM_OPT_PROFILE, 1)
        o_.reset_kernel_times()
    prof_steps = max(2, min(args.steps, 5)) * len(ring)
    for _ in range(prof_steps):
        step()
    sync()
    names = {capi.PSM_K_PREP: "prep", capi.PSM_K_CVC: "cvc", capi.PSM_K_GUIDE: "guidance", capi.PSM_K_CVF_F: "cvf_fused",
             capi.PSM_K_CVF_A: "cvf_a", capi.PSM_K_CVF_B: "cvf_b", capi.PSM_K_WTA: "wta",
             capi.PSM_K_MERGE: "merge", capi.PSM_K_FGF: "cvf_fgf", capi.PSM_K_LRC: "lr_check"}
    kern = {}
    for k, nm in names.items():
        tot, n = (sum(v) for v in zip(*[o_.kernel_time_ms(k) for o_ in ring]))
        if n:
            kern[nm] = {"avg_ms": tot / n, "launches_per_step": n / prof_steps, "source": "hipEvent pass (separate; perturbs the step)"}
    for o_ in ring:
        o_.set_option(capi.PSM_OPT_PROFILE, 0)
    fl = head["filter_launches"]
    if fl and not args.fgf:
        # the dominant kernel's entry comes from the timed region: mean over all its launches (planes and key phase)
        n_l = sum(v["per_step"] for v in fl.values())
        kern["cvf_fused"] = {"avg_ms": head["filter_ms_per_step"] / n_l, "launches_per_step": n_l,
                             "source": "time stamps taken by the kernel inside the timed region (PSM_OPT_PROFILE 2)", "by_form": fl}
    other_ms = sum(v["avg_ms"] * v["launches_per_step"] for nm, v in kern.items() if nm != "cvf_fused")
    kernels_sum = (head["filter_ms_per_step"] or 0.0) + other_ms
    # select form: the fused kernel also builds the costs and runs the WTA over its slices - it is credited with the whole
    # staged pipeline's algorithmic bytes (CVC + CVF + WTA); PSM_FLAG_STORE_FILTERED is the storing form (CVF only)
    fl_bits = max(args.flags, 0)
    select_mode = not args.fgf and args.variant == 0 and not (fl_bits & capi.PSM_FLAG_STORE_FILTERED)
    algb = dict(alg)
    if select_mode:
        algb["cvf_fused"] = alg["pipeline"]
    pipe_alg = alg["pipeline"]
    if args.fgf:
        # Algorithmic bytes of the STAGED Fast Guided Filter pipeline per full-resolution voxel, as SURVEY.md 8d counts the main
        # path's (every stage reads its input and writes its output once; FastGuidedFilterColor, src/fastguidedfilter.cpp:124-209,
        # between CostConst and DispSelect): cost volume 4 W; sub-sampling 4/s^2 R + 4/s^2 W; model stage 4/s^2 R + 16/s^2 W; smoothing
        # 16/s^2 R + 16/s^2 W; up-sampling + q = a.I + b: 16/s^2 R + 4 W; WTA 4 R  =  12 + 76/s^2.  Per kernel class of this
        # implementation: "cvf_fgf" (setup, sub-sampled costs, models, smoothing of one side) 60/s^2; "wta" (up-sample + apply +
        # argmin of one side: the q write and the WTA read of the staged form) 8 + 16/s^2.  (Measured 0.4 - 0.8 per kernel class; the
        # whole step on the staged bytes - pipeline_frac - can exceed 1 at s = 8: the path never builds a full-resolution volume.)
        s2_ = float(args.fgf * args.fgf)
        algb["cvf_fgf"], algb["wta"], pipe_alg = 60.0 / s2_, 8.0 + 16.0 / s2_, 12.0 + 76.0 / s2_
    dom = max(("cvf_fgf", "wta") if args.fgf else ("cvf_fused", "cvf_a"), key=lambda k: kern.get(k, {"avg_ms": 0})["avg_ms"])
    lps = max(1, round(kern[dom]["launches_per_step"]))
    vox_per_launch = 2.0 * W * (y1 - y0) * n_slices * B / lps                    # (this rank's rows and slices; all pairs of a batch)
    dom_ms = kern[dom]["avg_ms"]
    achieved = algb[dom] * vox_per_launch / (dom_ms * 1e-3) / 1e9
    two_phase = select_mode and "keys" in fl and "planes" in fl
    roofline = {"bound": "hbm", "kernel": ("k_cvf_pc (select mode: CVC+CVF+WTA fused" + (", two phases = two launches per step)" if two_phase else ")"))
                if (select_mode and dom == "cvf_fused") else "k_" + dom,
                "alg_bytes_per_voxel": algb[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": algb[dom] * vox_per_launch, "avg_launch_ms": round(dom_ms, 4),
                "launch_time_source": kern[dom]["source"],
                "note": "achieved / frac credit the fused kernel with the staged pipeline's ALGORITHMIC bytes (SURVEY.md 8d); they are "
                        "an algorithmic-equivalent rate, not bandwidth utilisation: the kernel's physical HBM rate is traffic_GBs "
                        "(traffic_frac of peak) and what bounds it is VALU issue (valu)",
                "pipeline_alg_bytes_per_voxel": pipe_alg,
                "pipeline_alg_GBs": round(pipe_alg * value / 1e9, 1),
                "pipeline_frac": round(pipe_alg * value / 1e9 / HBM_PEAK_GBS, 4)}
    if args.shard_sim > 1:      # (value of a --shard-sim line is the whole job over ONE share's time: not a throughput)
        roofline["pipeline_alg_GBs"] = roofline["pipeline_frac"] = None
    if args.fgf:
        roofline["note"] = ("Fast Guided Filter row: algorithmic bytes of the staged pipeline (12 + 76/s^2 per voxel; per kernel class "
                            "60/s^2 and 8 + 16/s^2 - bench.py); the sub-sampled costs are built on the fly and the filtered volume stays "
                            "virtual, so the physical traffic is far below these figures")
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    variant_flags = fl_bits & (capi.PSM_FLAG_F32_TOL | capi.PSM_FLAG_FMA_SOLVE)      # (the PMC figures are those of the default form, one pair per launch)
    if os.path.exists(traffic_file) and world == 1 and not args.shard_sim and select_mode and B == 1 and not variant_flags:
        try:
            tr = json.load(open(traffic_file))
            key = f"{args.config}:{dtype}:k_{dom}"
            if key in tr:
                roofline["traffic"] = tr[key]    # HBM bytes per launch from rocprofv3 PMC passes (not measured in this run)
                roofline["traffic_source"] = tr.get("_source", "profiles/traffic.json")
                roofline["traffic_session"] = tr.get(key + "_session", tr.get("_session"))      # box / date of the PMC session the figure comes from
                roofline["traffic_GBs"] = round(tr[key] / (dom_ms * 1e-3) / 1e9, 1)   # physical HBM rate of the kernel
                roofline["traffic_frac"] = round(tr[key] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                vi, s4 = tr.get(key + "_valu_insts"), tr.get(key + "_four_cycle_share")
                if vi and s4:
                    # What bounds the kernel.  Wave-instructions per launch (SQ_INSTS_VALU) split into the part's two VALU rate
                    # classes by the static instruction mix of the kernel's loop bodies (scripts/isa_mix.py): fp64 adds, fp64<->fp32
                    # conversions and DPP moves take 4 cycles per wave64, plain fp32 / integer ops 2; sustained issue rates
                    # measured on this part with the whole chip busy (scripts/exp/rate.hip): 0.50 G/s and 0.86 G/s per SIMD.
                    simds = 1024
                    bound_ms = 1e3 * vi * (s4 / 0.50e9 + (1.0 - s4) / 0.86e9) / simds
                    roofline["valu"] = {"wave_insts_per_launch": vi, "four_cycle_share": s4,
                                        "bound_ms_at_measured_issue_rates": round(bound_ms, 4),
                                        "frac_of_valu_bound": round(bound_ms / dom_ms, 4),
                                        "source": "SQ_INSTS_VALU (rocprofv3 PMC pass, " + str(tr.get(key + "_session", tr.get("_session"))) + "), mix from scripts/isa_mix.py"}
        except Exception:
            pass
    # ---- lead with the binding resource.  The fused select kernel keeps the staged pipeline's intermediates on chip: its physical
    # HBM rate is a small fraction of the peak (traffic_frac) and what bounds it is VALU issue - binding says so, next to the HBM
    # fraction on the algorithmic bytes that SURVEY.md 8d defines (48 B / voxel, unchanged) ----
    if select_mode and dom == "cvf_fused":
        # `bound` names the resource that binds (the contract lists hbm | mfma; for this kernel neither is true and the line says so
        # with its first key); achieved / peak / unit / frac stay the HBM figures on SURVEY.md 8d's algorithmic bytes (frac_of)
        roofline["bound"] = roofline["binding"] = "valu"
        roofline["frac_of"] = "hbm peak, algorithmic bytes (SURVEY.md 8d: 48 B / voxel for the fused CVC + CVF + WTA kernel)"
        if "valu" in roofline:
            roofline["binding_frac"] = roofline["valu"]["frac_of_valu_bound"]
            roofline["binding_note"] = ("VALU issue: SQ_INSTS_VALU per launch at the part's measured issue rates (0.50 / 0.86 G wave-instructions/s per SIMD "
                                        "for 4- / 2-cycle ops) over the launch time; the HBM figures (frac, traffic_frac) are context")
        else:
            roofline["binding_frac"] = None
            roofline["binding_note"] = "VALU issue (no PMC pass of this configuration in profiles/traffic.json: measured for c4 / c3 / c2 / c5 f32 and c4 u8)"
    else:
        roofline["binding"] = "hbm"
    roofline["frac_basis"] = "dominant kernel: algorithmic bytes per launch / launch time"
    if roofline["frac"] > 1.0 and roofline.get("pipeline_frac"):
        # An algorithmic-equivalent rate above the physical peak is not a utilisation: a fused kernel that never moves the staged
        # pipeline's intermediates can outrun that byte count.  The line then reports the whole step on the same bytes instead.
        roofline["kernel_alg_equiv_frac"] = roofline["frac"]
        roofline["kernel_alg_equiv_GBs"] = roofline["achieved"]
        roofline["achieved"], roofline["frac"] = roofline["pipeline_alg_GBs"], roofline["pipeline_frac"]
        roofline["frac_basis"] = ("whole step: pipeline algorithmic bytes / ms_per_step (the dominant kernel's own algorithmic-equivalent rate "
                                  "exceeds the HBM peak - kernel_alg_equiv_frac - which no bandwidth figure can)")
    if len(ring) > 1:
        # frames in flight: the launches of consecutive frames overlap, so a launch's own duration says little - the line reports the
        # whole step (frames / time) on the pipeline's algorithmic bytes
        roofline["kernel_alg_equiv_frac"], roofline["kernel_alg_equiv_GBs"] = roofline["frac"], roofline["achieved"]
        roofline["achieved"], roofline["frac"] = roofline["pipeline_alg_GBs"], roofline["pipeline_frac"]
        roofline["frac_basis"] = f"whole step: pipeline algorithmic bytes / ms_per_step ({len(ring)} frames in flight: launches of consecutive frames overlap)"
        roofline.pop("valu", None); roofline["binding_frac"] = None
    if roofline["frac"] is not None and roofline["frac"] > 1.0:
        # even the whole step outruns what the staged pipeline's bytes would need at the HBM peak (the opt-in tolerance form on the
        # round's fastest boxes: 1.00x): the figure stays what was measured - it is an algorithmic-equivalent rate, not a
        # utilisation - and the line says so instead of clamping it
        roofline["exceeds_hbm_peak"] = True
        roofline["exceeds_hbm_peak_note"] = ("frac > 1: the fused step finishes sooner than the staged pipeline's algorithmic bytes (48 B / voxel) could cross "
                                             "HBM at its peak; the kernel's physical HBM rate is traffic_frac of the peak and VALU issue binds it")
    for nm, v in kern.items():
        if nm in algb:
            v["alg_GBs"] = round(algb[nm] * vox_per_launch / (v["avg_ms"] * 1e-3) / 1e9, 1)
        v["avg_ms"] = round(v["avg_ms"], 4)

    # ---- CPU baseline: the oracle, driven like the reference pthreads path; its maps also check the timed path's ----
    cpu = None
    tol_model_maps = None
    fma_model_maps = None
    oracle_maps = None
    O = None
    sim = args.shard_sim > 1
    want_oracle = rank == 0 and not (args.fgf and dtype == "u8") and \
        ((world == 1 and not sim and not args.no_cpu_baseline) or ((world > 1 or sim) and not args.no_oracle_check))
    if want_oracle:
        from oracle import psm_oracle_py as O   # checker / baseline only - never on the GPU path
        cores = os.cpu_count() or 1
        threads = min(8, cores)                 # MAX_CPU_THREADS (include/ComFunc.h:52)
        sd = args.cpu_sample_d if args.cpu_sample_d > 0 else int(256.0 * (1920 * 1080) / (W * H))
        sd = max(2, min(sd, D))
        if world > 1 or sim:
            sd, threads = D, min(32, cores)     # N > 1 / --shard-sim: a checker run only (no cpu_baseline in the line)
        tcpu = time.perf_counter()
        if args.fgf:      # DispEst::CostFilter_FGF, the reference's live CPU branch (src/DispEst.cpp:281-296)
            res = O.pipeline_fgf(l, r, sd, s=args.fgf, threads=threads)
        else:
            res = (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, sd, threads=threads)
        tcpu = time.perf_counter() - tcpu
        if sd == D:
            oracle_maps = [res["ldisp"], res["rdisp"]]
            if dtype == "f32" and args.flags >= 0 and (args.flags & capi.PSM_FLAG_F32_TOL):
                # the tolerance form is checked against ITS model (oracle variant F32_L1: the same one extra fp32 rounding per pair of
                # taps) - bit for bit -, and the line says how many disparities differ from the canonical oracle's
                with O.variant(O.VAR_F32_L1):
                    rm_ = O.pipeline_f32(l, r, sd, threads=min(32, cores))
                tol_model_maps = [rm_["ldisp"], rm_["rdisp"]]
            if dtype == "f32" and args.flags >= 0 and (args.flags & capi.PSM_FLAG_FMA_SOLVE):
                # the FMA reading of the solve is checked against the oracle's same reading - bit for bit
                with O.variant(O.VAR_FMA_SOLVE):
                    rm_ = O.pipeline_f32(l, r, sd, threads=min(32, cores))
                fma_model_maps = [rm_["ldisp"], rm_["rdisp"]]
        elif dtype == "f32" and not args.no_oracle_check and not args.fgf:
            # (larger than 1080p x 256: the timed cpu_baseline is a sample of the disparities; the maps come from the oracle's
            # streaming form - same jobs and arithmetic, no volumes held - on up to 32 threads, outside every timed region)
            rs = O.pipeline_f32_maps(l, r, D, threads=min(32, cores))
            oracle_maps = [rs["ldisp"], rs["rdisp"]]
        if world == 1 and not sim:
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
            if not args.no_cpu_wide and wide > threads:
                resw = O.pipeline_fgf(l, r, sd, s=args.fgf, threads=wide) if args.fgf else \
                    (O.pipeline_u8 if dtype == "u8" else O.pipeline_f32)(l, r, sd, threads=wide)
                sw = (resw["cvc_ms"] + resw["cvf_ms"] + resw["dispsel_ms"]) * 1e-3
                cpu["wide"] = {"value": round(2.0 * W * H * sd / sw, 1), "cores": wide}

    # ---- result checks (outside the timed region): the maps the timed path left on the device against an unsharded
    # single-context run of the same pair on this GPU (with N > 1 this covers the RCCL exchange itself) and the oracle ----
    checks = {}
    ref_maps = None
    if rank == 0 and args.shard_sim <= 1:
        de.set_option(capi.PSM_OPT_ASYNC, 0)
        if use_dist or args.verify:
            ref_maps = single_gpu_maps()
        checks = check_maps(head, ref_maps, oracle_maps)
        if len(ring) > 1:           # every context of the ring left the same maps (same pair): frames in flight change no result
            same = True
            for o_ in ring[1:]:
                o_.set_option(capi.PSM_OPT_ASYNC, 0)
                got_ = o_.download_maps()
                same = same and bool(np.array_equal(got_[0], timed_maps[0]) and np.array_equal(got_[1], timed_maps[1]))
                o_.set_option(capi.PSM_OPT_ASYNC, 1)
            checks["frames_in_flight_maps_equal"] = same
        if tol_model_maps is not None and timed_maps is not None:
            checks["tolerance_form"] = {"flag": "PSM_FLAG_F32_TOL", "maps_equal_its_oracle_model": bool(np.array_equal(timed_maps[0], tol_model_maps[0]) and
                                                                                                   np.array_equal(timed_maps[1], tol_model_maps[1])),
                                        "pixels_differing_from_the_canonical_oracle": checks.get("oracle_map_mismatches")}
        if fma_model_maps is not None and timed_maps is not None:
            checks["fma_solve_form"] = {"flag": "PSM_FLAG_FMA_SOLVE", "maps_equal_its_oracle_reading": bool(np.array_equal(timed_maps[0], fma_model_maps[0]) and
                                                                                                        np.array_equal(timed_maps[1], fma_model_maps[1])),
                                        "pixels_differing_from_the_canonical_oracle": checks.get("oracle_map_mismatches")}
        if B > 1:      # every pair of the batch against its own single-pair run through the three reference entry points
            okb = True
            for o_, (pl_, pr_) in zip(batch_all, batch_pairs):
                got_ = [m.copy() for m in o_.download_maps()]
                with P.DispEst(pl_, pr_, D, 8, True, device=dev_index, dtype=dtype) as one:
                    if args.flags >= 0:
                        one.set_option(capi.PSM_OPT_FLAGS, args.flags)
                    one.CostConst_GPU(); one.CostFilter_GPU(); one.DispSelect_GPU()
                    okb = okb and bool(np.array_equal(one.lDisMap, got_[0]) and np.array_equal(one.rDisMap, got_[1]))
            checks["verified_vs_single_gpu"] = okb
            checks["batch_check"] = f"each of the {B} pairs of the batch == its own CostConst_GPU / CostFilter_GPU / DispSelect_GPU run"
        if timed_maps is not None and oracle_maps is not None:      # (the maps downloaded right after the timed region, too)
            checks["oracle_maps_equal"] = bool(checks["oracle_maps_equal"] and np.array_equal(timed_maps[0], oracle_maps[0])
                                               and np.array_equal(timed_maps[1], oracle_maps[1]))
        de.set_option(capi.PSM_OPT_ASYNC, 1)

    # ---- --shard-sim G: the timed context holds share 0 of a G-part job.  The other G - 1 shares run once, untimed, in fresh
    # contexts on this GPU; the whole is put together by the library's single-process exchange (psm_gather_rows_ctx /
    # psm_disp_merge_ctx) with the TIMED context as the root, and compared with the unsharded run and the oracle ----
    if rank == 0 and sim and not args.fgf:
        G = args.shard_sim
        de.set_option(capi.PSM_OPT_ASYNC, 0)
        step_all(); sync()
        parts_ctx = [de]
        for g_ in range(1, G):
            if rows_mode:
                _, ya, yb = stripes.stripe_bounds(H, G, g_)
                if yb <= ya:
                    continue
                o_ = P.DispEst(l, r, D, 8, True, device=dev_index, dtype=dtype)
                o_.set_rows(ya, yb)
            else:
                o_ = P.DispEst(l, r, D, 8, True, device=dev_index, d_range=(D * g_ // G, D * (g_ + 1) // G), dtype=dtype,
                               d_stride=((g_, G) if args.strided else None))
            if args.flags >= 0:
                o_.set_option(capi.PSM_OPT_FLAGS, args.flags)
            o_.CostConst_GPU()
            o_.CostFilter_GPU()
            o_.DispSelect_device() if rows_mode else o_.DispSelect_partial()
            parts_ctx.append(o_)
        if rows_mode:
            de.gather_rows_ctx(parts_ctx)
        else:
            de.DispSelect_merge_ctx(parts_ctx)
        whole = [de.lDisMap.copy(), de.rDisMap.copy()]
        for o_ in parts_ctx[1:]:
            o_.close()
        ref_maps = single_gpu_maps()[:2]
        checks["verified_vs_single_gpu"] = bool(all(np.array_equal(a, b) for a, b in zip(ref_maps, whole)))
        if oracle_maps is not None:
            nl, nr = int(np.count_nonzero(whole[0] != oracle_maps[0])), int(np.count_nonzero(whole[1] != oracle_maps[1]))
            checks["oracle_maps_equal"], checks["oracle_map_mismatches"] = nl == 0 and nr == 0, [nl, nr]
        checks["shard_sim_check"] = (f"share 0 (timed context) + {G - 1} untimed shares on this GPU, put together by "
                                     + ("psm_gather_rows_ctx" if rows_mode else "psm_disp_merge_ctx"))
        de.set_option(capi.PSM_OPT_ASYNC, 1)

    # ---- post-processing stages on the finished maps (PP::processDM: lrCheck, fillInv, wgtMedian; src/PP.cpp:405-410) ----
    pp = None
    if args.pp and rank == 0 and not use_dist and args.shard_sim <= 1:
        de.set_option(capi.PSM_OPT_ASYNC, 0)
        pp = {}
        for it in range(2):                      # second pass: scratch allocated, clocks up
            step_all(); sync()
            raw_l, raw_r = (m.copy() for m in de.download_maps())
            t = time.perf_counter(); de.LRCheck_device(); de.synchronize(); pp["lr_check_ms"] = round(1e3 * (time.perf_counter() - t), 4)
            lv, rv = (m.copy() for m in de.download_valid())
            t = time.perf_counter(); de._ck(de._lib.psm_fill_invalid(de._h, None, None, 0), "fill"); de.synchronize()
            pp["fill_inv_ms"] = round(1e3 * (time.perf_counter() - t), 4)
            t = time.perf_counter(); de._ck(de._lib.psm_wgt_median(de._h, None, None, 0), "wmf"); de.synchronize()
            pp["wgt_median_ms"] = round(1e3 * (time.perf_counter() - t), 4)
        out_l, out_r = (m.copy() for m in de.download_maps())
        sw, ev = de.wgt_median_stats()
        pp["invalid_frac"] = [round(float(1.0 - lv.mean()), 4), round(float(1.0 - rv.mean()), 4)]
        pp["wgt_median_sweeps"], pp["wgt_median_evals"] = sw, ev
        if O is None:
            from oracle import psm_oracle_py as O
        elv, erv = O.lr_check(raw_l, raw_r)
        ok = bool(np.array_equal(lv, elv) and np.array_equal(rv, erv))
        if ok:
            t = time.perf_counter()
            fl_, fr_ = O.fill_inv(raw_l, elv), O.fill_inv(raw_r, erv)
            lf, rf = O.u8_to_f32(l), O.u8_to_f32(r)
            el = O.wgt_median(lf, fl_, elv, D, right=False)
            er = O.wgt_median(rf, fr_, erv, D, right=True)
            pp["oracle_s"] = round(time.perf_counter() - t, 2)
            ok = bool(np.array_equal(out_l, el) and np.array_equal(out_r, er))
        pp["verified_vs_oracle"] = ok
        pp["note"] = "stage wall times incl. launch + synchronisation, maps resident on the device; not part of value"
        de.set_option(capi.PSM_OPT_ASYNC, 1)

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
        bv = 1.0 * W * H * (d1 - d0)             # one volume
        box = {"avg_ms": round(bms, 4), "alg_GBs": round(8.0 * bv / (bms * 1e-3) / 1e9, 1),
               "read_GBs": round(4.0 * bv / (bms * 1e-3) / 1e9, 1),
               "read_frac_of_peak": round(4.0 * bv / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "fused_pass_equivalent_ms": round(ms_per_step / 16.0, 4),
               "fused_pass_equivalent_read_frac": round(4.0 * bv / (ms_per_step / 16.0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "note": "the north star's '>= 60 % of HBM-read roofline on the CVF box-filter pass': a stand-alone pass that writes as much as "
                       "it reads cannot reach it (read_frac_of_peak); inside the fused kernel the 16 box passes of a frame take "
                       "ms_per_step / 16 each (fused_pass_equivalent_*) - only in that accounting is it met"}
    for o_ in batch_all:
        o_.close()

    # ================= the other sharding axis (N > 1) =================
    alt = None
    if use_dist and not args.no_alt_shard and not args.fgf:
        other = "disp" if args.shard == "rows" else "rows"
        ok_axis = (other == "disp" and world <= D) or (other == "rows" and (world - 1) * -(-H // world) < H)
        if ok_axis:
            a = measure(other, args.exchange or "allgather")
            alt = {"shard": other, "exchange": a["exchange"], "ms_per_step": a["ms_per_step"], "value": a["value"],
                   "median_ms_per_step": a["median_ms_per_step"], "filter_launches": a["filter_launches"], "per_rank": a.get("per_rank"),
                   "frames_in_flight": len(a["ring"]),
                   "parallelism": (f"D sharded over {world} ranks + 1 {xname} {a['exchange']} of packed minima per frame" if other == "disp"
                                   else f"{world} row stripes + 1 {xname} all_gather of the map rows per frame"),
                   "note": ("the configuration BASELINE configs[3] / the north star name (D slices sharded, one all-gather of per-pixel minima); "
                            "same maps as the headline axis" if other == "disp" else "row stripes; same maps as the headline axis")}
            if rank == 0:
                a["sync"]()
                a["de"].set_option(capi.PSM_OPT_ASYNC, 0)
                alt.update(check_maps(a, ref_maps, oracle_maps))
            for o_ in a["batch_all"]:
                o_.close()

    if rank == 0:
        out = {
            "metric": "cost-volume voxels/s (CVC+CVF+WTA)" if not args.fgf else f"cost-volume voxels/s (CVC+CVF_FGF s={args.fgf}+WTA)",
            "value": value, "unit": "voxels/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": data_note,
            "config": {"workload": desc, "W": W, "H": H, "D": D, "voxels_per_step": voxels_per_step,
                       "parallelism": "1 GPU" if world == 1 and not use_dist else
                                      (f"{world} row stripes of {geo['rows_max']} rows (all {D} slices each) + 1 {xname} all_gather of the map rows per frame"
                                       if rows_mode else f"D sharded over {world} ranks + 1 {xname} {exchange} of packed minima"),
                       "kernel_variant": args.variant, "shard_sim": args.shard_sim, "lr_check_on_gpu": bool(lrc),
                       "shard": head["shard"], "batch": B, "ranks": world, "same_device": bool(args.same_device),
                       "frames_in_flight": len(head["ring"]), "strided_disparity_shards": bool(geo["strided"]),
                       "exchange_backend": (backend if use_dist else None)},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kern,
            "kernels_sum_ms_per_step": round(kernels_sum, 4), "kernels_sum_le_step": bool(kernels_sum <= ms_per_step * 1.005),
            "median_ms_per_step": head["median_ms_per_step"], "pcie": pcie,
        }
        if B > 1:
            out["ms_per_pair"] = ms_per_step / B
            out["config"]["workload"] = f"{B} x " + desc + " per step (psm_compute_batch: one set of launches for all pairs)"
        out.update(checks)
        if args.same_device and use_dist:
            out["same_device_note"] = (f"{world} ranks share GPU 0: the N > 1 protocol (alternating buffers, pending exchange, gather / merge at world "
                                       f"{world}) run for correctness - value is NOT a scaling measurement"
                                       + (f"; {args.backend_note}" if args.backend_note else ""))
        if use_dist:
            out["ranks"] = world
            out["exchange_backend"] = backend
            out["shard"] = head["shard"]
            out["exchange"] = head["exchange"]
            out["per_rank"] = head.get("per_rank")
            out["frame_pipeline"] = bool(not args.no_frame_pipeline)
        if alt:
            out["alt_shard"] = alt
        if pp:
            out["pp"] = pp
        if box:
            out["box_filter_pass"] = box
        if json_fd is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))
            sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
