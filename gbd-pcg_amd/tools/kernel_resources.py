#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in gbd-pcg_amd/csrc, from hipcc -Rpass-analysis=kernel-resource-usage with the
flags of the Makefile (no GPU needed):  python gbd-pcg_amd/tools/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt
DESIGN.md quotes these lines; kernels with scratch are listed first."""
import glob, os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off".split()
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
rows = []
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = None
    for ln in out.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", ln)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"file": os.path.basename(f), "name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, nm in zip(rows, names):
    r["demangled"] = re.sub(r"\(.*", "", nm).replace("void ", "").replace("gbdpcg::", "")
rows.sort(key=lambda r: (-int(r.get("ScratchSize [bytes/lane]", 0)), r["file"], r["demangled"]))
print("# hipcc -Rpass-analysis=kernel-resource-usage, flags of csrc/Makefile; scratch first")
print(f"{'kernel':78s} {'file':22s} VGPR AGPR scratch vspill sspill  LDS  waves/SIMD")
for r in rows:
    print(f"{r['demangled'][:78]:78s} {r['file']:22s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('ScratchSize [bytes/lane]','?'):>7s} "
          f"{r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} {r.get('LDS Size [bytes/block]','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
