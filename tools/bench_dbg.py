#!/usr/bin/env python3
"""bench.py against the profiling build of the library (tools/libbevy_terrain_amd_dbg.so, `make -C
bevy_terrain_amd/csrc debug`), whose fused kernels honour the BT_FUSED_* ablation variables.  Results of ablated
runs are NOT valid tiles; this exists only for git history, tools/experiments/ablate_sweep.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bevy_terrain_amd import _ffi

_ffi.LIB_PATH = os.environ.get("BT_LIB") or os.path.join(ROOT, "tools", "libbevy_terrain_amd_dbg.so")  # BT_LIB: another build (A/B runs)
import bench

if os.environ.get("BT_PAD_MB"):  # placement experiment: device memory taken before the bench allocates anything
    import torch

    _pad = [torch.empty(int(mb) << 20, dtype=torch.uint8, device="cuda") for mb in os.environ["BT_PAD_MB"].split(",")]
bench.main()
