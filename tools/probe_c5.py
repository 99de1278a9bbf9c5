"""python3 tools/probe_c5.py: value and frame time of bench.py --config c5 with the library named by PATHTRACING_AMD_LIB."""
import json
import os
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--config", "c5", "--no-cpu-baseline", "--no-hbm-regime", "--steps", "5", "--warmup", "2"],
                     capture_output=True, text=True, timeout=600).stdout
for ln in out.splitlines():
    if ln.startswith('{"metric"'):
        j = json.loads(ln)
        print(os.path.basename(os.environ.get("PATHTRACING_AMD_LIB", "default")), "c5", round(j["value"], 1), "Mrays/s", round(j["ms_per_step"], 2), "ms", flush=True)
