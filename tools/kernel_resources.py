"""Per-kernel resources as the compiler reports them (hipcc -Rpass-analysis=kernel-resource-usage, gfx950): VGPRs, AGPRs,
SGPRs, scratch, static LDS, waves/SIMD by registers.  No GPU needed.

    python tools/kernel_resources.py profiles/r02/kernel_resources.csv
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssvio_amd import build as b   # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.csv")
rows = []
for src in b.sources():
    cmd = [b.hipcc(), *b.COMMON, *b.PER_FILE.get(src, []), "-c", os.path.join(b.CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: (?:\s*)(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            dem = re.sub(r"\(anonymous namespace\)::", "", dem)
            k = re.search(r"(k_\w+(?:<[^>]*>)?)", dem)
            cur = dict(file=src, kernel=k.group(1) if k else dem[:60])
            rows.append(cur)
        elif cur is not None and ":" in t:
            key, val = [x.strip() for x in t.split(":", 1)]
            cur[key] = val
cols = ["file", "kernel", "VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]
# dynamic LDS (extern __shared__, set at launch): the library knows what it launches its BA kernels with
import ctypes
lib = ctypes.CDLL(b.build())
lib.ssx_debug_kernel_dynamic_lds.restype = ctypes.c_int64
lib.ssx_debug_kernel_dynamic_lds.argtypes = [ctypes.c_char_p]
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    f.write("file,kernel,vgprs,agprs,sgprs,scratch_bytes_per_lane,static_lds_bytes_per_block,waves_per_simd_by_registers,dynamic_lds_bytes_per_block\n")
    for r in rows:
        if not r["kernel"].startswith("k_"):
            continue
        dyn = lib.ssx_debug_kernel_dynamic_lds(r["kernel"].split("<")[0].encode()) if r["file"] == "ba.hip" else 0
        f.write(",".join('"%s"' % r.get(c, "") if c == "kernel" else str(r.get(c, "")) for c in cols) + f",{dyn}\n")
print(out, len(rows), "kernels")
