"""Compressed view of one kernel's instruction stream (hipcc -S output): runs of instruction classes, waits and barriers spelled
out.  python tools/isa_stream.py <file.s> <mangled-name-substring> [start_line end_line]"""
import re
import sys


def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): return "T"
    if op.startswith("v_permlane"): return "P"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_read"): return "r"
    if op.startswith("ds_write"): return "w"
    if op.startswith("buffer_load") or op.startswith("global_load"): return "L"
    if op.startswith("buffer_store") or op.startswith("global_store"): return "S"
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "J"
    if op.startswith("s_"): return "s"
    return "?"


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r"\n(_Z\S*" + re.escape(key) + r"\S*):", s)
    i = m.end()
    j = s.index(".amdhsa_kernel", i)
    lines = s[i:j].split("\n")
    if len(sys.argv) > 4:
        lines = lines[int(sys.argv[3]):int(sys.argv[4])]
    out, run, last = [], 0, None
    for n, l in enumerate(lines):
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            if last: out.append(f"{last}{run}")
            last, run = None, 0
            out.append(f"\n[{t} @{n}]")
            continue
        if not l.startswith("\t") or t.startswith((".", ";")) or not t:
            continue
        op = t.split()[0]
        c = cls(op)
        if c in "WBJ" or "lds" in t and c == "L":
            if last: out.append(f"{last}{run}")
            last, run = None, 0
            if c == "W": out.append("<" + t.replace("s_waitcnt ", "") + ">")
            elif c == "B": out.append("|BAR|")
            elif c == "J": out.append("{" + t + "}")
            else: out.append("D")
            continue
        if c == last: run += 1
        else:
            if last: out.append(f"{last}{run}")
            last, run = c, 1
    if last: out.append(f"{last}{run}")
    print(" ".join(out))


if __name__ == "__main__":
    main()
