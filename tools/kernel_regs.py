"""VGPR / SGPR / spill counts per kernel of one .hip file (hipcc -S, CPU only):  python tools/kernel_regs.py flair_amd/csrc/conv.hip [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S",
                        "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), src, "-o", out] + sys.argv[3:],
                       check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    names, rows = [], []
    for b in s.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
        names.append(g("name"))
        rows.append((b.split("\n")[0].strip(), g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for n, (ag, v, sg, sp, pv) in zip(dem, rows):
        n = n.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16").split("(")[0].replace("void ", "")
        if flt in n:
            print(f"{n:72s} vgpr {v:>4s} (agpr {ag:>3s}) sgpr {sg:>3s} spill {sp} scratch {pv}")


if __name__ == "__main__":
    main()
