"""An experimental variant of libssx.so beside the real one: one source file recompiled with extra flags (and, optionally, with a
patch from tools/patches/ applied to a temporary copy first), linked with the other objects of the last regular build, written to
ssvio_amd/libssx.so.<name> (git-ignored; it travels to the GPU box).  Select it with SSX_LIB=$PWD/ssvio_amd/libssx.so.<name>.

    python tools/build_variant.py <name> [source.hip] [--patch tools/patches/x.diff] [flags ...]
    e.g.  python tools/build_variant.py noslab --patch tools/patches/ba_experiment_switches.diff -DSSX_EXP_SKIP_SLAB_WRITES

The shipped kernels carry no experiment switches.  tools/patches/ba_experiment_switches.diff puts back the -DSSX_EXP_SKIP_* switches of
ba.hip (one phase of the linearise / Schur kernels left out: wrong results, meaningful times -- how profiles/r04/schur_phase_ab.md
was measured, with tools/ba_persist_ab.py as the timer) and the -DSSX_PHASE_CLOCK stamps (tools/ba_phase_clock.py);
tools/patches/orb_fast_stop_switches.diff the -DSSX_EXP_FAST_STOP=1..4 cuts of k_fast_cells (profiles/r04/fast_cells_phases.md).  A patch
was cut against the commit that added it; after later edits of the kernel it may need `patch --fuzz` or a refresh."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ssvio_amd import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
src = "ba.hip"
if flags and flags[0].endswith(".hip"):                      # python tools/build_variant.py <name> orb.hip -D...: another source file
    src, flags = flags[0], flags[1:]
patch = None
if flags and flags[0] == "--patch":
    patch, flags = os.path.abspath(flags[1]), flags[2:]
b.build()
cc = b.hipcc()
obj = f"/tmp/ssx_variant_{name}.o"
src_dir = b.CSRC
tmp = None
if patch:
    tmp = tempfile.mkdtemp(prefix="ssx_variant_")
    shutil.copytree(os.path.join(ROOT, "ssvio_amd"), os.path.join(tmp, "ssvio_amd"), ignore=shutil.ignore_patterns("*.so*", "build", "host"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    subprocess.check_call(["patch", "-p0", "--fuzz=3", "-i", patch], cwd=tmp)
    src_dir = os.path.join(tmp, "ssvio_amd", "csrc")
subprocess.check_call([cc, *b.COMMON, *b.PER_FILE.get(src, []), *flags, "-c", os.path.join(src_dir, src), "-o", obj], stderr=subprocess.DEVNULL)
objs = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.sources() if s != src] + [obj]
out = os.path.join(ROOT, "ssvio_amd", f"libssx.so.{name}")
subprocess.check_call([cc, "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-o", out, *objs])
if tmp:
    shutil.rmtree(tmp, ignore_errors=True)
print(out)
