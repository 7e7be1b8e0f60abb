"""Static scan of the gfx950 ISA of csrc/*.hip for the instruction pattern that made bank_get fault (DESIGN.md,
"eager two-stream hazard"): a vector-memory STORE issued while older vector-memory LOADS are still outstanding,
followed by a COUNTED wait (s_waitcnt vmcnt(N), N > 0) whose consumers then read load results.

    python tools/isa_store_scan.py            # compiles every csrc/*.hip to assembly (hipcc -S) and lists the kernels
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)")
STORE = re.compile(r"^\s*(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic)")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")


def scan(path):
    """{kernel: [(line number, outstanding loads at the store, counted wait text)]}"""
    hits, kernel, pending_loads, store_seen = {}, None, 0, None
    for ln, line in enumerate(open(path), 1):
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            kernel, pending_loads, store_seen = m.group(1), 0, None
            continue
        if m:  # basic-block label: keep counting conservatively (loads may be outstanding across it)
            continue
        if kernel is None:
            continue
        if LOAD.match(line):
            pending_loads += 1
        elif STORE.match(line):
            if pending_loads > 0:
                store_seen = (ln, pending_loads)
        else:
            w = WAIT.match(line)
            if w:
                v = VMCNT.search(w.group(1))
                if v is None:
                    continue
                n = int(v.group(1))
                if n == 0:
                    pending_loads, store_seen = 0, None
                else:
                    if store_seen is not None:
                        hits.setdefault(kernel, []).append((store_seen[0], store_seen[1], line.strip()))
                        store_seen = None
                    pending_loads = min(pending_loads, n)
        if "s_endpgm" in line:
            kernel = None
    return hits


def main():
    out = tempfile.mkdtemp(prefix="isa_scan_")
    total = 0
    for src in sorted(glob.glob(os.path.join(ROOT, "simpb_amd", "csrc", "*.hip"))):
        s = os.path.join(out, os.path.basename(src)[:-4] + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", s, src] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
        for kernel, rows in scan(s).items():
            total += 1
            print(f"{os.path.basename(src)}: {kernel}: {len(rows)} site(s); first: store at line {rows[0][0]} with "
                  f"{rows[0][1]} load(s) outstanding, then `{rows[0][2]}`")
    print(f"{total} kernel(s) issue a store among outstanding loads in front of a counted vmcnt wait")


if __name__ == "__main__":
    main()
