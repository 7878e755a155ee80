"""Static VALU instruction mix of k_cvf_pc's loop bodies (no GPU needed): compiles psm_pc.hip to gfx950 assembly and counts,
per producer / consumer batch loop of the product instantiations, the wave-instructions of the part's two VALU rate classes

    four-cycle  : fp64 adds / moves, fp64 <-> fp32 conversions, DPP moves, packed fp32 - and, measured in round 3
                  (scripts/exp/clock.hip: 0.54-0.56 G wave-instructions/s per SIMD against 0.84-0.92 for the two-cycle
                  class): v_sad_u8, v_mul_u32_u24, integer <-> fp32 conversions, v_trunc / v_rndne_f32, v_med3_f32
                  (the 8-bit cost and re-quantisation); 64-bit integer ops are counted here too (not measured)
    two-cycle   : everything else in the VALU (fp32 add / mul / fma, 32-bit integer and bit ops, compares)

plus LDS, VMEM and scalar instructions.  The rarely executed border-cost block of the producer (columns x < d) is left
out.  bench.py's roofline.valu uses `four_cycle_share` together with the measured SQ_INSTS_VALU (rocprofv3 PMC pass) to
state what the kernel's time is bounded by.

    python scripts/isa_mix.py [--json profiles/r03/isa_mix.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "primestereomatch_amd", "csrc", "psm_pc.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"]
# template arguments <VEC4, CVC, MODE, U8, BATCH, VAR, NARROW>
KERNELS = {"planes_f32": "k_cvf_pcILb0ELi3ELi1ELb0ELb0ELi0ELb0EEE", "keys_f32": "k_cvf_pcILb0ELi3ELi2ELb0ELb0ELi0ELb0EEE",
           "planes_u8": "k_cvf_pcILb0ELi3ELi1ELb1ELb0ELi0ELb0EEE", "keys_u8": "k_cvf_pcILb0ELi3ELi2ELb1ELb0ELi0ELb0EEE",
           "planes_f32_fma": "k_cvf_pcILb0ELi3ELi1ELb0ELb0ELi2ELb0EEE", "keys_f32_fma": "k_cvf_pcILb0ELi3ELi2ELb0ELb0ELi2ELb0EEE",
           "planes_f32_narrow": "k_cvf_pcILb0ELi3ELi1ELb0ELb0ELi0ELb1EEE", "keys_f32_narrow": "k_cvf_pcILb0ELi3ELi2ELb0ELb0ELi0ELb1EEE",
           "planes_u8_narrow": "k_cvf_pcILb0ELi3ELi1ELb1ELb0ELi0ELb1EEE", "keys_u8_narrow": "k_cvf_pcILb0ELi3ELi2ELb1ELb0ELi0ELb1EEE"}


def classify(op, rest):
    if op.startswith("v_"):
        if re.search(r"_f64|f64_f32|f32_f64|_b64|_u64|_i64", op) or op.startswith("v_pk_") or "dpp" in op:
            return "V4"
        if re.match(r"v_(sad_u8|mul_u32_u24|mul_i32_i24|cvt_f32_[ui]32|cvt_[ui]32_f32|cvt_f32_ubyte\d|trunc_f32|rndne_f32|med3_f32)", op):
            return "V4"
        if "row_" in rest or "wave_" in rest or "quad_perm" in rest:
            return "V4"
        return "V2"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "VMEM"
    if op.startswith("s_"):
        return "S"
    return "other"


def function_body(lines, mangled):
    cur, buf = None, []
    for ln in lines:
        m = re.match(r"^(_ZN3psm\w+):", ln)
        if m:
            cur, buf = m.group(1), []
        elif ln.startswith(".Lfunc_end") and cur:
            if mangled in cur:
                return buf
            cur = None
        elif cur:
            buf.append(ln)
    raise SystemExit("kernel not found: " + mangled)


def loops(body):
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, ln in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def count(body, a, b):
    # basic blocks; drop the border-cost block (marker comment emitted by the kernel source)
    blocks, cur = [], []
    for ln in body[a:b + 1]:
        if re.match(r"^(\.LBB\d+_\d+):|^; %bb\.", ln):
            blocks.append(cur)
            cur = []
        cur.append(ln)
    blocks.append(cur)
    c = collections.Counter()
    for blk in blocks:
        if any("; border cost" in ln for ln in blk):
            continue
        for ln in blk:
            m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", ln)
            if not m or ln.strip().startswith((".", ";")):
                continue
            c[classify(m.group(1), m.group(2))] += 1
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "pc.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-x", "hip", "-S", "--cuda-device-only", SRC, "-o", asm], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    out = {}
    for name, mangled in KERNELS.items():
        body = function_body(lines, mangled)
        # the two batch loops: the SHORTEST loops that contain two batches of sliding trees (>= 400 four-cycle ops); loops that
        # enclose them (the slice loop of the plane form, the exit-path copies) are longer
        roles = {}
        for n, a, b in sorted((b - a, a, b) for a, b in loops(body) if b - a > 400):
            c = count(body, a, b)
            if c["V4"] < 400:
                continue
            role = "producer" if c["VMEM"] > 30 else "consumer"
            if role not in roles:
                roles[role] = {k: c[k] for k in ("V4", "V2", "LDS", "VMEM", "S")}
        v4 = sum(r["V4"] for r in roles.values())
        v2 = sum(r["V2"] for r in roles.values())
        out[name] = {"per_two_batches": roles, "four_cycle_share": round(v4 / (v4 + v2), 4),
                     "valu_per_row_step_pair": round((v4 + v2) / 8.0, 1)}
        print(name, json.dumps(out[name]))
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
