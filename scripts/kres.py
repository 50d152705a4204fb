"""Development: register / LDS / spill summary of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kres.py raster_render.hip [name filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", f"-I{root}/include", f"-I{root}/gomavatar_amd/csrc",
       "-c", f"{root}/gomavatar_amd/csrc/{src}", "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage", *extra]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur.replace("(anonymous namespace)::", "").replace("void ", ""))
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if flt in k:
        print(f"{k:60s} VGPR {v.get('VGPRs', -1):4d} SGPR {v.get('TotalSGPRs', -1):4d} scratch {v.get('ScratchSize', -1):4d} occ {v.get('Occupancy', -1):2d} LDS {v.get('LDS Size', -1):6d} spillV {v.get('VGPRs Spill', -1)}")
