"""profiles/traffic.json from the rocprofv3 PMC summaries of a scripts/gpu_round.sh session (passes rd, wr, sq) and the static
instruction mix (scripts/isa_mix.py -> profiles/<tag>/isa_mix.json).
    python scripts/make_traffic.py gpurun_out/<tag> <tag>
Fabric-side bytes per launch = TCC_EA0_RDREQ_sum*128 (these kernels issue no 32-byte requests; equals
2*FETCH_SIZE*1024, the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE*1024.  Keys: "<config>:<dtype>:k_cvf_fused" =
mean over the launches of a step (planes phase + key phase), like bench.py's roofline.avg_launch_ms."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, kernel):
    """counters of the first kernel whose name matches the regex `kernel`"""
    out, cur = {}, None
    for line in open(path):
        if line.startswith("### counters:"):
            cur = line
        elif cur and re.search(kernel, cur):
            m = re.match(r"\s+(\S+)\s+avg/dispatch = (\S+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
    return out


def fabric_bytes(d, pre, kernel):
    rd = counters(os.path.join(d, f"{pre}rd.summary.txt"), kernel)
    wr = counters(os.path.join(d, f"{pre}wr.summary.txt"), kernel)
    r32 = rd.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    return (rd["TCC_EA0_RDREQ_sum"] - r32) * 128 + r32 * 32, wr["WRITE_SIZE"] * 1024


def main():
    d, tag = sys.argv[1], sys.argv[2]
    session = "?"
    for name in ("device.txt", "device_small.txt"):          # (device_small.txt: a partial session, scripts/gpu_round_small.sh)
        if os.path.exists(os.path.join(d, name)):
            session = open(os.path.join(d, name)).read().strip().replace("\n", "; ")
            break
    mixf = os.path.join(ROOT, "profiles", tag, "isa_mix.json")
    if not os.path.exists(mixf):              # (scripts/isa_mix.py of this round not run yet: the latest committed mix)
        have = sorted(t for t in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", t, "isa_mix.json")))
        mixf = os.path.join(ROOT, "profiles", have[-1], "isa_mix.json")
    mix = json.load(open(mixf))
    out = {
        "_comment": "Fabric-side (L2 <-> Infinity Fabric) bytes per launch from rocprofv3 PMC passes (scripts/gpu_round.sh, "
                    "scripts/make_traffic.py): TCC_EA0_RDREQ_sum*128 (+32 B per 32-byte request; equals 2*FETCH_SIZE*1024, the "
                    "gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE*1024. Requests served by the Infinity cache (MALL) "
                    "are included, so this is an upper bound of the HBM bytes.  <config>:<dtype>:k_cvf_fused is the mean over the "
                    "two launches of a step (planes phase k_cvf_pc<false,3,1>, key phase k_cvf_pc<false,3,2>), like bench.py's "
                    "avg_launch_ms; *_valu_insts = SQ_INSTS_VALU per launch (same mean), *_four_cycle_share = share of 4-cycle "
                    "VALU ops in the loop bodies (scripts/isa_mix.py), weighted by the two launches' instruction counts.",
        "_source": f"profiles/{tag}/rocprofv3_pmc_*{{rd,wr,sq}}.summary.txt (c4: pmc_{{rd,wr,sq}} / pmc_u8_*; other configs pmc_<config>_*)",
        "_session": session,
    }
    # (config, dtype, file prefix of its passes): c4 from scripts/gpu_round.sh (pmc_ / pmc_u8_), the other configs pmc_<config>_
    cases = [("c4", "f32", "pmc_"), ("c4", "u8", "pmc_u8_"), ("c3", "f32", "pmc_c3_"), ("c2", "f32", "pmc_c2_"), ("c5", "f32", "pmc_c5_")]
    prev = {}
    if os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")):
        prev = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for cfg, dt, pre in cases:
        u8 = "true" if dt == "u8" else "false"
        # (the sixth template argument was a bool before round 5's VAR, and round 5 added the layout argument NARROW: all spellings)
        ks = {"planes": rf"k_cvf_pc<false, 3, 1, {u8}, false, (false|0)(, (false|true))?>", "keys": rf"k_cvf_pc<false, 3, 2, {u8}, false, (false|0)(, (false|true))?>"}
        key = f"{cfg}:{dt}:k_cvf_fused"
        try:
            parts, insts = {}, {}
            for n, k in ks.items():
                try:
                    parts[n] = fabric_bytes(d, pre, k)
                    insts[n] = counters(os.path.join(d, f"{pre}sq.summary.txt"), k)["SQ_INSTS_VALU"]
                except KeyError:
                    pass                      # (a configuration below 112 slices runs the plane form only)
            if not parts:
                raise FileNotFoundError("no k_cvf_pc counters")
        except Exception as e:
            print(f"({cfg} {dt}: no PMC summaries in this session: {e}; keeping the committed figures)")
            for k_, v_ in prev.items():
                if k_.startswith(key) or k_ == f"{cfg}:{dt}:k_chunk_min":
                    out[k_] = v_
            if any(k_.startswith(key) for k_ in prev):
                out[key + "_session"] = prev.get(key + "_session", prev.get("_session"))
            continue
        nl = len(parts)
        out[key] = round(sum(r + w for r, w in parts.values()) / nl)
        out[key + "_read"] = round(sum(r for r, _ in parts.values()) / nl)
        out[key + "_write"] = round(sum(w for _, w in parts.values()) / nl)
        out[key + "_launches_per_step"] = nl
        out[key + "_by_form"] = {n: {"read": round(r), "write": round(w), "valu_insts": round(insts[n])} for n, (r, w) in parts.items()}
        out[key + "_valu_insts"] = round(sum(insts.values()) / nl)
        s4 = sum(insts[n] * mix[f"{n}_{dt}"]["four_cycle_share"] for n in parts) / sum(insts.values())
        out[key + "_four_cycle_share"] = round(s4, 4)
        out[key + "_session"] = session
        try:
            r2, w2 = fabric_bytes(d, pre, "k_chunk_min")
            out[f"{cfg}:{dt}:k_chunk_min"] = round(r2 + w2)
        except Exception:
            pass
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
