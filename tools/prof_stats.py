"""Run a command under `rocprofv3 --kernel-trace --stats` (on the GPU box) and print the per-kernel table.

    python tools/prof_stats.py [--top N] -- python tools/ba_c4_time.py
"""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

args = sys.argv[1:]
top = 25
if args and args[0] == "--top":
    top = int(args[1]); args = args[2:]
if args and args[0] == "--":
    args = args[1:]
d = tempfile.mkdtemp(prefix="ssxprof", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", *args], cwd="/tmp", env=env,
                   capture_output=True, text=True)
for line in r.stdout.splitlines():
    if not line.startswith(("W2", "E2", "I2")):
        print(line)
fs = sorted(glob.glob(os.path.join(d, "*", "*kernel_stats.csv")))
if not fs:
    print(r.stderr[-2000:]); sys.exit(1)
rows = list(csv.DictReader(open(fs[-1])))
print(f"{'kernel':44s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
for row in rows[:top]:
    m = re.search(r"(k_\w+(?:<[^>]*>)?|__amd_rocclr_\w+)", row["Name"])
    n = m.group(1) if m else row["Name"][:44]
    print(f"{n:44s} {int(row['Calls']):7d} {float(row['TotalDurationNs']) / 1e6:10.3f} {float(row['AverageNs']) / 1e3:9.2f} "
          f"{float(row['MinNs']) / 1e3:9.2f} {float(row['MaxNs']) / 1e3:9.2f} {float(row['Percentage']):6.2f}")
