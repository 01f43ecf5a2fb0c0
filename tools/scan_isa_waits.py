"""Compiles every kernel file to gfx950 assembly (hipcc -S, no GPU needed) and lists, per kernel, how often a global / buffer load is
followed within four instructions by `s_waitcnt vmcnt(0)` (a serial load round trip) and how often such a wait follows a store
within twelve (waiting for a store's acknowledgement).  hipcc emits both behind EXEC-masked branches around per-lane accesses
("what hipcc does to a load or a store under a branch"); run this after touching an epilogue or a prologue.

    python tools/scan_isa_waits.py [file.hip ...]        (default: flair_amd/csrc/*.hip; ~1 min per file)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "flair_amd", "csrc", "*.hip")))
tmp = tempfile.mkdtemp(prefix="isa_")
for f in files:
    out = os.path.join(tmp, os.path.basename(f) + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=fast", "--offload-arch=gfx950", "-S",
                        "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), f, "-o", out],
                       capture_output=True, text=True)
    if r.returncode:
        print(f, "did not compile:", r.stderr[-400:])
        continue
    lines = open(out).read().splitlines()
    kern, res = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S+):", l)
        if m:
            kern = m.group(1)
            res[kern] = [0, 0, 0]
            continue
        if kern is None:
            continue
        t = l.strip()
        if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            back = [x.strip() for x in lines[max(0, i - 4):i]]
            if any(b.startswith(("global_load", "buffer_load")) and " lds" not in b for b in back):
                res[kern][0] += 1
            back = [x.strip() for x in lines[max(0, i - 12):i]]
            if any(b.startswith(("global_store", "buffer_store")) for b in back):
                res[kern][1] += 1
            res[kern][2] += 1
    for k, v in res.items():
        if v[0] >= 3 or v[1] >= 2:
            try:
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
            except OSError:
                name = k
            print(f"{os.path.basename(f):12s} {name[:110]:110s} load->vmcnt(0): {v[0]:3d}   store..vmcnt(0): {v[1]:3d}   vmcnt(0) total: {v[2]:3d}")
