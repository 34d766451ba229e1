#!/usr/bin/env python3
"""tools/config_bench.py against the profiling build (BT_FUSED_* switches, e.g. BT_FUSED_PARTS = row blocks per workgroup of fused_direct)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bevy_terrain_amd import _ffi

_ffi.LIB_PATH = os.environ.get("BT_LIB") or os.path.join(ROOT, "tools", "libbevy_terrain_amd_dbg.so")  # BT_LIB: another build (A/B runs)
import config_bench

config_bench.main()
