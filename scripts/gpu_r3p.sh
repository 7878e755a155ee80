#!/bin/bash
# Round-3 session P: per-workgroup traces of the fused kernel (debug build, -DPSM_PC_TIMING=1) -> profiles/r03/wg_trace.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
export PRIMESM_HIP_LIB=$GRAFT_REPO_ROOT/primestereomatch_amd/lib/libprimesm_hip_dbg.so
{
timeout 300 python scripts/dbg_pc_trace.py 1920,1080,256,0,32,0,270 gpurun_out/r3p/d32.npz
timeout 300 python scripts/dbg_pc_trace.py 1920,1080,256,0,256,0,270 gpurun_out/r3p/d256.npz
timeout 300 python scripts/dbg_pc_timing.py 1920,1080,256,0,32,0,270 1920,1080,256,0,256,2097152,270
timeout 300 python scripts/dbg_pc_clock.py 1920,1080,256,0,32,0,270,4,1 1920,1080,256,0,32,0,270,30,0 1920,1080,256,0,256,0,270,30,0
} 2>&1 | tee gpurun_out/r3p/out.txt | tail -80
