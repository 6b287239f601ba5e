"""Per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py mvtracker_amd/csrc/conv_rows.hip [name filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", "/dev/null", src,
                    "-Rpass-analysis=kernel-resource-usage", *extra], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["/usr/bin/c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, d in rows.items():
    if filt in name:
        short = name.replace("(anonymous namespace)::", "").split("(")[0]
        print(f"{short:70s} VGPR {d.get('VGPRs', '?'):>4s} AGPR {d.get('AGPRs', '0'):>3s} spill {d.get('VGPRs Spill', '?'):>3s} scratch {d.get('ScratchSize', '?'):>4s} "
              f"LDS {d.get('LDS Size', '?'):>7s} waves/SIMD {d.get('Occupancy', '?')}")
