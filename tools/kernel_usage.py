"""python3 tools/kernel_usage.py [usage.txt]: registers, spills, occupancy and LDS of every kernel from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (default: the library build's pathtracing_amd/csrc/build/kernels/resource_usage.txt)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pathtracing_amd/csrc/build/kernels/resource_usage.txt")
OCC = r"Occupancy \[waves/SIMD\]"
LDS = r"LDS Size \[bytes/block\]"
for block in re.split(r"remark: Function Name: ", open(path).read())[1:]:
    name = block.split(" ")[0]

    def g(key):
        m = re.search(key + r": (\d+)", block)
        return m.group(1) if m else "?"

    pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    pretty = re.sub(r"\(.*", "", pretty)[:64]
    print(f"{pretty:64s} vgpr {g('VGPRs'):>4s} spilled {g('VGPRs Spill'):>3s} sgpr-spilled {g('SGPRs Spill'):>3s} "
          f"waves/SIMD {g(OCC)} lds {g(LDS)}")
