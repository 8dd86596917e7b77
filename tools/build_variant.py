"""An experimental variant of libssx.so beside the real one: ba.hip recompiled with extra flags, linked with the other objects of
the last regular build, written to ssvio_amd/libssx.so.<name> (git-ignored; it travels to the GPU box).  Select it with
SSX_LIB=$PWD/ssvio_amd/libssx.so.<name>.

    python tools/build_variant.py <name> [flags ...]          e.g.  python tools/build_variant.py noslab -DSSX_EXP_SKIP_SLAB_WRITES

The -DSSX_EXP_SKIP_* switches of ba.hip leave one phase of the linearise / Schur kernels out (wrong results, meaningful times): how
profiles/r04/schur_phase_ab.md was measured, with tools/ba_persist_ab.py as the timer."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssvio_amd import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
src = "ba.hip"
if flags and flags[0].endswith(".hip"):                      # python tools/build_variant.py <name> orb.hip -D...: another source file
    src, flags = flags[0], flags[1:]
b.build()
cc = b.hipcc()
obj = f"/tmp/ssx_variant_{name}.o"
subprocess.check_call([cc, *b.COMMON, *b.PER_FILE.get(src, []), *flags, "-c", os.path.join(b.CSRC, src), "-o", obj], stderr=subprocess.DEVNULL)
objs = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.sources() if s != src] + [obj]
out = os.path.join(ROOT, "ssvio_amd", f"libssx.so.{name}")
subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", out, *objs])
print(out)
