#!/bin/bash
# Per-kernel register / LDS / spill table of one HIP source, from the gfx950 code object's metadata (no GPU needed):
#   tools/kernel_stats.sh representationlearning_amd/csrc/win_attn_fwd.hip
set -e
src=$1; out=${TMPDIR:-/tmp}/ks_$(basename "$src" .hip).co
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast --cuda-device-only -c "$src" -o "$out.bundle"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$out.bundle" --output="$out"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$out" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    import subprocess
    try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()[:110]
    except Exception: pass
    agpr = re.match(r":\s*(\d+)", blk)
    print("vgpr %4s agpr %4s spill %4s sgpr %4s lds %7s scratch %6s  %s" % (g("vgpr_count"), agpr.group(1) if agpr else "?", g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), name))
'
