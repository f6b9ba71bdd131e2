"""Kernel resource table (VGPR/AGPR/scratch/occupancy/LDS) for every k_* instantiation, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (device-only compile, no GPU needed).
usage: python tools/kernel_resources.py [filter-substring] [extra hipcc flags...]"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "roborugby_amd", "csrc", "rr_kernels.hip")
flt = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
extra = [a for a in sys.argv[1:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "--offload-device-only", "-c",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/rr_kres.o", SRC] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?):\s*(.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}; rows.append(cur)
    elif cur is not None: cur[k] = v
for r in rows:
    m = re.match(r"void (\w+)<rr::Cfg<(\d+), (\d+), (\d+), (\d+), (\w+), (\d+)>(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?>", r["name"])
    tag = r["name"][:60] if not m else "%s %s+%s/%s+%s %s VW%s out=%s multi=%s budget=%s" % m.groups()
    if flt and flt not in tag: continue
    print("%-72s vgpr %3s agpr %3s scratch %4s occ %s spillV %3s lds %6s" % (tag, r.get("VGPRs"), r.get("AGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
