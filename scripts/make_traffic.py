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
    out, cur = {}, None
    for line in open(path):
        if line.startswith("### counters:"):
            cur = line
        elif cur and kernel in cur:
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
    session = open(os.path.join(d, "device.txt")).read().strip().replace("\n", "; ") if os.path.exists(os.path.join(d, "device.txt")) else "?"
    mix = json.load(open(os.path.join(ROOT, "profiles", tag, "isa_mix.json")))
    out = {
        "_comment": "Fabric-side (L2 <-> Infinity Fabric) bytes per launch from rocprofv3 PMC passes (scripts/gpu_round.sh, "
                    "scripts/make_traffic.py): TCC_EA0_RDREQ_sum*128 (+32 B per 32-byte request; equals 2*FETCH_SIZE*1024, the "
                    "gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE*1024. Requests served by the Infinity cache (MALL) "
                    "are included, so this is an upper bound of the HBM bytes.  <config>:<dtype>:k_cvf_fused is the mean over the "
                    "two launches of a step (planes phase k_cvf_pc<false,3,1>, key phase k_cvf_pc<false,3,2>), like bench.py's "
                    "avg_launch_ms; *_valu_insts = SQ_INSTS_VALU per launch (same mean), *_four_cycle_share = share of 4-cycle "
                    "VALU ops in the loop bodies (scripts/isa_mix.py), weighted by the two launches' instruction counts.",
        "_source": f"profiles/{tag}/rocprofv3_pmc_{{rd,wr,sq}}*.summary.txt",
        "_session": session,
    }
    for dt, u8 in (("f32", "false"), ("u8", "true")):
        pre = "pmc_" if dt == "f32" else "pmc_u8_"
        ks = {"planes": f"k_cvf_pc<false, 3, 1, {u8}, false, false>", "keys": f"k_cvf_pc<false, 3, 2, {u8}, false, false>"}
        try:
            parts = {n: fabric_bytes(d, pre, k) for n, k in ks.items()}
            insts = {n: counters(os.path.join(d, f"{pre}sq.summary.txt"), k)["SQ_INSTS_VALU"] for n, k in ks.items()}
        except Exception as e:
            print(f"({dt}: no PMC summaries: {e})")
            continue
        key = f"c4:{dt}:k_cvf_fused"
        out[key] = round(sum(r + w for r, w in parts.values()) / 2)
        out[key + "_read"] = round(sum(r for r, _ in parts.values()) / 2)
        out[key + "_write"] = round(sum(w for _, w in parts.values()) / 2)
        out[key + "_launches_per_step"] = 2
        out[key + "_by_form"] = {n: {"read": round(r), "write": round(w), "valu_insts": round(insts[n])} for n, (r, w) in parts.items()}
        out[key + "_valu_insts"] = round(sum(insts.values()) / 2)
        s4 = sum(insts[n] * mix[f"{n}_{dt}"]["four_cycle_share"] for n in ks) / sum(insts.values())
        out[key + "_four_cycle_share"] = round(s4, 4)
        try:
            r2, w2 = fabric_bytes(d, pre, "k_chunk_min")
            out[f"c4:{dt}:k_chunk_min"] = round(r2 + w2)
        except Exception:
            pass
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
