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
    """{kernel: [(line of the counted wait, stores in flight, loads in flight, wait text)]}: sites where a counted
    `s_waitcnt vmcnt(N > 0)` releases consumers while at least one vector-memory store AND one load may still be in
    flight (issued since the last vmcnt(0), in either order: `load.. store.. wait` as bank_get had it, or
    `store.. (loop back) load.. wait` as dfa_points had it). Conservative across basic-block labels."""
    hits, kernel, loads, stores = {}, None, 0, 0
    for ln, line in enumerate(open(path), 1):
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            kernel, loads, stores = m.group(1), 0, 0
            continue
        if m or kernel is None:
            continue
        if LOAD.match(line):
            loads += 1
        elif STORE.match(line):
            stores += 1
        else:
            w = WAIT.match(line)
            if w:
                v = VMCNT.search(w.group(1))
                if v is not None:
                    n = int(v.group(1))
                    if n == 0:
                        loads = stores = 0
                    elif stores > 0 and loads > 0 and n < loads + stores:
                        hits.setdefault(kernel, []).append((ln, stores, loads, line.strip()))
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
            print(f"{os.path.basename(src)}: {kernel}: {len(rows)} site(s); first at line {rows[0][0]}: `{rows[0][3]}` with "
                  f"{rows[0][1]} store(s) and {rows[0][2]} load(s) possibly in flight")
    print(f"{total} kernel(s) release consumers on a counted vmcnt wait with stores and loads in flight")


if __name__ == "__main__":
    main()
