"""Per-kernel registers / scratch / occupancy / LDS of one translation unit, from hipcc's kernel-resource-usage remarks:
    python tools/kernel_resources.py fc_forward.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "coevonet_amd", "csrc", sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "--cuda-device-only", "-S", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
KEYS = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]")]
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split(" ")[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn)
    vals = []
    for label, k in KEYS:
        m = re.search(k + r": (\d+)", b)
        vals.append(f"{label} {m.group(1) if m else '?'}")
    print(f"{dn:60s} " + "  ".join(vals))
