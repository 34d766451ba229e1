#!/bin/bash
# Register / scratch / LDS use of every kernel of bt_fused.hip as compiled for gfx950 (the metadata the assembler emits):
#   tools/kernel_resources.sh [pattern]
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/bevy_terrain_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I../../include -I. -S --cuda-device-only -o /tmp/bt_fused_gfx950.s bt_fused.hip 2>/dev/null
python3 - "$1" <<'PY'
import re, sys
text = open("/tmp/bt_fused_gfx950.s").read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name:
        continue
    def g(k):
        r = re.search(r"\." + k + r"\s+(\S+)", body)
        return r.group(1) if r else "?"
    print(name[:110])
    print("   vgpr", g("amdhsa_next_free_vgpr"), "sgpr", g("amdhsa_next_free_sgpr"), "scratch", g("amdhsa_private_segment_fixed_size"), "lds", g("amdhsa_group_segment_fixed_size"))
for m in re.finditer(r"; (ScratchSize|codeLenInByte|NumVgprs|Occupancy|VGPR spill|SGPR spill|sgpr_spill_count|vgpr_spill_count)[^\n]*", text):
    pass
# the per-function remarks
for m in re.finditer(r"^(_Z\S+|\S+):.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text, flags=re.S | re.M):
    pass
PY
grep -E "^; (Kernel|NumVgprs|ScratchSize|Occupancy|SGPRBlocks)|vgpr_spill_count|sgpr_spill_count|\.name:" /tmp/bt_fused_gfx950.s | grep -A3 "${1:-fused_main}" | head -60
