"""Round 6: alternating same-box A/B of a few cuts of the two-phase selection (experiment build; see exp_plan6.py)."""
import sys
sys.path.insert(0, 'scripts')
sys.path.insert(0, '.')
from exp_plan6 import run, CFG
from primestereomatch_amd import synth

CASES2 = {"c4": [{}, {"PSM_PC_S": 5}, {"PSM_PC_S": 6}, {"PSM_PC_S": 10}], "c4u8": [{}, {"PSM_PC_S": 5}, {"PSM_PC_S": 6}, {"PSM_PC_S": 4}], "c3": [{}, {"PSM_PC_S": 5}],
          "c5": [{}, {"PSM_PC_S": 4}], "rows8": [{}, {"PSM_PC_S": 5}, {"PSM_PC_S": 4}], "rows4": [{}, {"PSM_PC_S": 5}], "c3u8": [{}, {"PSM_PC_S": 5}]}
CASES = {"c4": [{}, {"PSM_PC_S": 8}, {"PSM_PC_S": 8, "PSM_PC_SEGK": 360}, {"PSM_PC_SEGK": 360}, {"PSM_PC_S": 6, "PSM_PC_SEGK": 216}, {"PSM_PC_S": 10, "PSM_PC_SEGK": 360}],
         "c3": [{}, {"PSM_PC_S": 8}, {"PSM_PC_S": 8, "PSM_PC_SEGK": 240}, {"PSM_PC_S": 10}, {"PSM_PC_S": 12}, {"PSM_PC_S": 8, "PSM_PC_DC": 1}],
         "c5": [{}, {"PSM_PC_S": 5}, {"PSM_PC_S": 8}, {"PSM_PC_S": 8, "PSM_PC_SEGK": 360}, {"PSM_PC_S": 8, "PSM_PC_SEGK": 540}],
         "rows8": [{}, {"PSM_PC_S": 4}, {"PSM_PC_S": 8}, {"PSM_PC_DC": 1}]}
if sys.argv[1:2] == ["--new"]:
    CASES = CASES2
    del sys.argv[1]
CFG["c4u8"] = CFG["c4"]; CFG["c3u8"] = CFG["c3"]; CFG["rows4"] = (1920, 1080, 256, 30, (0, 270))
for cfg in (sys.argv[1:] or ["c4", "c3"]):
    W, H, D, steps, rows = CFG[cfg]
    dtype = "u8" if cfg.endswith("u8") else "f32"
    l, r, _ = synth.make_pair(W, H, D, seed=0)
    res = {i: [] for i in range(len(CASES[cfg]))}
    for rep in range(5 if cfg != "c5" else 3):
        for i, env in enumerate(CASES[cfg]):
            res[i].append(run(l, r, D, env, steps, rows, dtype))
    for i, env in enumerate(CASES[cfg]):
        v = sorted(res[i])
        print(f"{cfg}: {env or 'planner'}: median {v[len(v) // 2]:.4f} min {v[0]:.4f} max {v[-1]:.4f}", flush=True)
