"""profiles/traffic.json from the rocprofv3 PMC summaries of scripts/gpu_round.sh (passes rd, wr).
    python scripts/make_traffic.py gpurun_out/<tag> [kernel substring] [source label]
Fabric-side bytes per launch = TCC_EA0_RDREQ_sum*128 (these kernels issue no 32-byte requests; equals
2*FETCH_SIZE*1024, the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE*1024."""
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


def fabric_bytes(d, kernel):
    rd = counters(os.path.join(d, "pmc_rd.summary.txt"), kernel)
    wr = counters(os.path.join(d, "pmc_wr.summary.txt"), kernel)
    r32 = rd.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    return (rd["TCC_EA0_RDREQ_sum"] - r32) * 128 + r32 * 32, wr["WRITE_SIZE"] * 1024


def main():
    d = sys.argv[1]
    # the select form of psm_cost_filter is two launches per step from 160 local slices up (planes phase MODE 1, key phase
    # MODE 2): bench.py's "per launch" figures are the mean over the launches of a step, so is the traffic here
    kernels = sys.argv[2].split("|") if len(sys.argv) > 2 else ["k_cvf_pc<false, 3, 1", "k_cvf_pc<false, 3, 2"]
    source = sys.argv[3] if len(sys.argv) > 3 else "profiles/ (rocprofv3 --pmc passes)"
    parts = {}
    for k in kernels:
        try:
            parts[k] = fabric_bytes(d, k)
        except Exception:
            pass
    if not parts:
        raise SystemExit("no PMC counters for " + repr(kernels))
    n = len(parts)
    rbytes = sum(v[0] for v in parts.values()) / n
    wbytes = sum(v[1] for v in parts.values()) / n
    out = {
        "_comment": "Fabric-side (L2 <-> Infinity Fabric) bytes per launch from rocprofv3 PMC passes (scripts/gpu_round.sh, "
                    "scripts/make_traffic.py): TCC_EA0_RDREQ_sum*128 (+32 B per 32-byte request; equals 2*FETCH_SIZE*1024, the "
                    "gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE*1024. Requests served by the Infinity cache (MALL) "
                    "are included, so this is an upper bound of the HBM bytes. "
                    f"Kernel(s): {', '.join(parts)} (select form, both volumes per launch, costs built on the fly); "
                    f"c4:k_cvf_fused is the mean over the {n} launch(es) of a step, like bench.py's avg_launch_ms.",
        "_source": source,
        "c4:k_cvf_fused": round(rbytes + wbytes),
        "c4:k_cvf_fused_read": round(rbytes),
        "c4:k_cvf_fused_write": round(wbytes),
        "c4:k_cvf_fused_launches_per_step": n,
    }
    for k, v in parts.items():
        out["c4:" + k] = {"read": round(v[0]), "write": round(v[1])}
    # VALU wave-instructions per launch (SQ_INSTS_VALU of the sq pass), same mean: what actually bounds the kernel
    try:
        insts = [counters(os.path.join(d, "pmc_sq.summary.txt"), k)["SQ_INSTS_VALU"] for k in parts]
        out["c4:k_cvf_fused_valu_insts"] = round(sum(insts) / n)
    except Exception:
        pass
    try:
        r2, w2 = fabric_bytes(d, "k_chunk_min")
        out["c4:k_chunk_min"] = round(r2 + w2)
    except Exception:
        pass
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
