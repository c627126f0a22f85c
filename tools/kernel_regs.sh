#!/bin/bash
# register / scratch / LDS use of every kernel of a translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
#   tools/kernel_regs.sh acav100m_amd/csrc/acav_kmeans.hip [-D...] > profiles/rNN_train_regs.txt
cd "$(dirname "$0")/.."
src=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -Wno-unused-function -Wno-inline-asm -I include"
hipcc $F "$@" -c "$src" -o /tmp/kernel_regs_$$.o -Rpass-analysis=kernel-resource-usage 2> /tmp/kernel_regs_$$.log
python3 - /tmp/kernel_regs_$$.log "$src" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
print("# %s: hipcc -Rpass-analysis=kernel-resource-usage (gfx950); one wave per SIMD = up to 512 registers (VGPR + AGPR)" % sys.argv[2])
print("%-110s %5s %5s %8s %5s %9s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS bytes"))
seen = set()
for b in txt.split("Function Name: ")[1:]:
    name = b.split("[")[0].strip()
    if name in seen:
        continue
    seen.add(name)
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)
    dem = re.sub(r"\(.*", "", dem)[:110]
    print("%-110s %5s %5s %8s %5s %9s" % (dem, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
PY
rm -f /tmp/kernel_regs_$$.o /tmp/kernel_regs_$$.log
