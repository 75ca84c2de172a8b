"""Kernel resource table (VGPRs, SGPRs, scratch, spills, LDS, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: kernel_resources.py <source.hip> [filter-substring ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
filters = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + __import__("os").environ.get("SG_EXTRA_FLAGS", "").split() + [
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
if "tile2d" in src:
    cmd[5:5] = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = demangle(b.split()[0])
    if filters and not all(f in name for f in filters):
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [None, "?"])[1]
    print("%-70s VGPR %3s AGPR %3s SGPR %3s scratch %4s B/lane  vspill %3s sspill %3s  LDS %6s  waves/SIMD %s" % (
        name.replace("sg::", "").replace("(sg::StageArgs)", ""), g("VGPRs"), g("AGPRs"), g("TotalSGPRs"),
        g("ScratchSize [bytes/lane]"), g("VGPRs Spill"), g("SGPRs Spill"), g("LDS Size [bytes/block]"), g("Occupancy [waves/SIMD]")))
