"""-m gpu: bench.py end to end - the JSON contract on a live run, and the torch.distributed (RCCL) path of the N > 1
bench inside the driver's test run: world size 1 always, world size 2 when the box has two GPUs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]       # ONE JSON line
    return json.loads(lines[0])


def test_bench_line_live_small_config():
    j = _bench("--config", "c2", "--steps", "5", "--warmup", "2", "--cpu-sample-d", "8", "--verify")
    assert j["n_gpus"] == 1 and j["unit"] == "voxels/s" and j["dtype"] == "f32"
    assert abs(j["value"] - j["config"]["voxels_per_step"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    r = j["roofline"]
    assert r["bound"] == r["binding"] == "valu" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3     # (bound agrees with binding)
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    assert j["verified_vs_single_gpu"] is True
    assert j["median_ms_per_step"] > 0 and j["pcie"]["h2d_ms"] > 0 and j["pcie"]["d2h_ms"] > 0
    # the fused kernel's time comes from the timed region itself (in-kernel stamps): it fits inside the step it describes
    k = j["kernels"]["cvf_fused"]
    assert "timed region" in k["source"] and k["avg_ms"] * k["launches_per_step"] <= j["ms_per_step"]
    assert j["pcie"]["frame_loop"]["maps_equal_timed_path"] is True and j["pcie"]["frame_loop"]["ms_per_frame"] > 0


def test_bench_line_oracle_checks_the_timed_maps():
    """The oracle run that provides cpu_baseline also checks the maps the TIMED path left on the device (whole D)."""
    j = _bench("--config", "c3", "--steps", "3", "--warmup", "1", "--pp")
    assert j["oracle_maps_equal"] is True and j["oracle_map_mismatches"] == [0, 0]
    assert j["kernels_sum_ms_per_step"] <= 1.03 * j["ms_per_step"]      # (the small kernels come from a perturbed event pass)
    assert j["pp"]["verified_vs_oracle"] is True and j["pp"]["wgt_median_ms"] > 0


@pytest.mark.parametrize("shard,exchange,extra", [("rows", "allreduce", ()), ("rows", "allreduce", ("--no-frame-pipeline",)),
                                                  ("disp", "allreduce", ()), ("disp", "allgather", ())])
def test_distributed_path_world1_verified(shard, exchange, extra):
    """The RCCL call sequence of the N > 1 bench with one rank, maps checked against the plain single-context run:
    row stripes (all_gather of the finished map rows into torch-owned map tensors) and disparity shards (packed-key local
    WTA, collective, merge)."""
    j = _bench("--gpus", "1", "--force-dist", "--shard", shard, "--exchange", exchange, "--config", "c3", "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", *extra)
    assert j["verified_vs_single_gpu"] is True and j["scaling"] == "strong" and j["config"]["shard"] == shard
    # one invocation times both axes: the other one sits in alt_shard, verified the same way
    a = j["alt_shard"]
    assert a["shard"] == ("disp" if shard == "rows" else "rows") and a["verified_vs_single_gpu"] is True and a["ms_per_step"] > 0


@pytest.mark.parametrize("shard,extra", [("rows", ()), ("disp", ()), ("rows", ("--no-frame-pipeline",)), ("disp", ("--exchange", "allgather"))])
def test_distributed_path_world2(shard, extra):
    """bench.py's real step() at world size 2, both axes per invocation, frame-pipelined and not: on two GPUs over RCCL when
    the box has them; on a one-GPU box both ranks share device 0 (--same-device; RCCL is tried first and, when it refuses the
    duplicate device, the same collectives are staged through host memory over gloo behind the same pending / finish_pending /
    alternating-buffer code).  Everything that only exists at world >= 2 runs here: stripes.assemble on gathered tensors,
    DispSelect_merge(world = 2) on a real all-gather result, set_map_buffer(whole) / set_key_buffer alternation."""
    from primestereomatch_amd import capi
    same = () if capi.device_count() >= 2 else ("--same-device",)
    j = _bench("--gpus", "2", "--shard", shard, "--config", "c3", "--steps", "4", "--warmup", "2", "--frames-in-flight", "2", *same, *extra, timeout=900)   # bare command: self-launch
    assert j["n_gpus"] == 2 and j["config"]["ranks"] == 2 and j["config"]["shard"] == shard
    assert j["verified_vs_single_gpu"] is True and j["oracle_maps_equal"] is True
    a = j["alt_shard"]
    assert a["shard"] != shard and a["verified_vs_single_gpu"] is True and a["oracle_maps_equal"] is True
    assert j["config"]["exchange_backend"] in ("nccl", "gloo") and j["config"]["same_device"] is bool(same)
    assert j["config"]["frames_in_flight"] == 2 and a["frames_in_flight"] == 2 and j["frames_in_flight_maps_equal"] is True


def test_shard_sim_lines_are_verified():
    """--shard-sim G: the timed share + the other G - 1 shares (untimed) put together by the library's single-process exchange
    equal the unsharded maps and the oracle's."""
    for shard in ("rows", "disp"):
        j = _bench("--shard-sim", "4", "--shard", shard, "--config", "c3", "--steps", "3", "--warmup", "1")
        assert j["verified_vs_single_gpu"] is True and j["oracle_maps_equal"] is True, shard
    # ... and with two frames in flight on the share (the line has no per-launch roofline then: frac is null, not an error)
    j = _bench("--shard-sim", "4", "--shard", "rows", "--frames-in-flight", "2", "--config", "c3", "--steps", "4", "--warmup", "2", "--no-cpu-baseline")
    assert j["verified_vs_single_gpu"] is True and j["config"]["frames_in_flight"] == 2 and j["roofline"]["frac"] is None


@pytest.mark.parametrize("parts", [2, 8])
def test_one_stripe_of_n_live(parts):
    """--shard-sim N --shard rows: the rank-local work of an N-way row-stripe run (stripe 0) on this GPU."""
    j = _bench("--shard-sim", str(parts), "--shard", "rows", "--config", "c3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["config"]["shard"] == "rows" and j["ms_per_step"] > 0


@pytest.mark.parametrize("extra", [(), ("--gpus", "1", "--force-dist", "--shard", "rows"), ("--gpus", "1", "--force-dist", "--shard", "disp")])
def test_lr_check_inside_the_step(extra):
    """BASELINE configs[4]: the PP left-right check runs on the GPU as part of the step (default for c5, --lr-check 1 elsewhere);
    the validity masks of the timed path equal those of the plain one-GPU run."""
    j = _bench("--config", "c3", "--lr-check", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--verify", *extra)
    assert j["config"]["lr_check_on_gpu"] is True and j["verified_vs_single_gpu"] is True


@pytest.mark.parametrize("cfg,extra", [("c2", ("--pair", "fixture", "--seg-rows", "375")), ("c1x", ("--pair", "fixture")), ("c3", ("--pp",))])
def test_frames_in_flight_and_the_middlebury_fixtures(cfg, extra):
    """Round 5: --frames-in-flight 2 (two contexts on two streams take the steps in turn) and --pair fixture (Teddy for c2, Cones /
    its 384 x 288 crop for the c1 configs): same maps on every context of the ring, equal to the single-context run and the oracle;
    the line says what its roofline fraction refers to; N > 1 fields appear on distributed lines only."""
    j = _bench("--config", cfg, "--frames-in-flight", "2", "--steps", "6", "--warmup", "2", "--cpu-sample-d", "0", "--no-cpu-wide", "--verify", *extra)
    assert j["config"]["frames_in_flight"] == 2 and j["frames_in_flight_maps_equal"] is True
    assert j["verified_vs_single_gpu"] is True and j["oracle_maps_equal"] is True
    r = j["roofline"]
    assert "whole step" in r["frac_basis"] and r["frac"] <= 1.0 and r["binding"] == "valu" and "valu" not in r
    assert "ranks" not in j and "per_rank" not in j
    if "fixture" in extra:
        assert "Middlebury" in j["data"]
    if "--pp" in extra:
        assert j["pp"]["verified_vs_oracle"] is True


def test_n_gt_1_line_explains_itself():
    """Round 5: ranks, backend, shard, exchange, frame pipeline and per-rank compute / collective ms for both axes."""
    j = _bench("--gpus", "1", "--force-dist", "--config", "c3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["ranks"] == 1 and j["exchange_backend"] == "nccl" and j["shard"] == "rows" and j["frame_pipeline"] is True
    # round 6: two frames in flight per rank are the distributed path's default where a rank's share is short enough to gain -
    # row stripes from 4 ranks, disparity shards from 2 (measured per share) - so at world 1 only the other axis ...
    assert j["config"]["frames_in_flight"] == 1 and j["alt_shard"]["frames_in_flight"] == 2
    # ... and on request on both; every context of the ring ends with the same (verified) maps
    j = _bench("--gpus", "1", "--force-dist", "--frames-in-flight", "2", "--config", "c3", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert j["config"]["frames_in_flight"] == 2 and j["alt_shard"]["frames_in_flight"] == 2 and j["frames_in_flight_maps_equal"] is True
    assert j["verified_vs_single_gpu"] is True and j["alt_shard"]["verified_vs_single_gpu"] is True
    for rec in (j, j["alt_shard"]):
        pr = rec["per_rank"]
        assert len(pr["compute_ms"]) == 1 and len(pr["collective_ms"]) == 1 and pr["compute_ms"][0] > pr["collective_ms"][0] > 0

