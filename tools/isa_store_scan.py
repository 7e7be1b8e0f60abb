"""Static scan of the gfx950 ISA of csrc/*.hip for the instruction pattern that made bank_get fault (DESIGN.md,
"eager two-stream hazard"): a vector-memory STORE issued while older vector-memory LOADS are still outstanding,
followed by a COUNTED wait (s_waitcnt vmcnt(N), N > 0) whose consumers then read load results.

    python tools/isa_store_scan.py            # compiles every csrc/*.hip to assembly (hipcc -S) and lists the kernels
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)")
STORE = re.compile(r"^\s*(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic)")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
BRANCH = re.compile(r"^\s*s_c?branch\w*\s+([.\w$]+)")

# the four double-K 16-bit matrix instructions gfx950 adds (and their 8-bit / 4-bit siblings of the same issue width):
# while one executes anywhere on the chip, other kernels' vector arithmetic goes wrong in lanes 48-63
# (profiles/r02_mfma_x16_interference/). Nothing in the product library may issue them.
BANNED_MFMA = re.compile(r"v_s?mfmac?_\w*?_(32x32x16|16x16x32|32x32x64|16x16x128|32x32x32|16x16x64)_?\w*")


def hipcc():
    """Path of hipcc ($HIPCC, /opt/rocm/bin/hipcc, PATH) or None: callers skip instead of failing without ROCm."""
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


PACKED_FP32 = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def compile_all(out_dir, extra=()):
    """hipcc -S (device only) of every csrc/*.hip into out_dir with the flags the product build gives that file
    (simpb_amd/build.py: flags_for), in parallel; {stem: path of the .s}."""
    cc = hipcc()
    if cc is None:
        raise FileNotFoundError("hipcc")
    sys.path.insert(0, ROOT)
    from simpb_amd import build
    srcs = sorted(glob.glob(os.path.join(ROOT, "simpb_amd", "csrc", "*.hip")))

    def one(src):
        s = os.path.join(out_dir, os.path.basename(src)[:-4] + ".s")
        flags = [f for f in build.flags_for(src) if f != "-fPIC"]
        subprocess.run([cc] + flags + ["-S", "--cuda-device-only", "-o", s, src] + list(extra), check=True, stderr=subprocess.DEVNULL)
        return os.path.basename(src)[:-4], s

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        return dict(pool.map(one, srcs))


def _kernels(path):
    """[(kernel name, [(line number, text)])] of every function body in an assembly listing."""
    out, name, body = [], None, []
    for ln, line in enumerate(open(path), 1):
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            if name is not None:
                out.append((name, body))
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append((ln, line))
        # up to the first s_endpgm: blocks laid out behind it are reached by branches the straight-line reading cannot
        # order (their loads would be counted as issued after the stores in front of them)
        if "s_endpgm" in line:
            out.append((name, body))
            name, body = None, []
    return out


def scan(path, loops=True):
    """{kernel: [(line of the counted wait, stores in flight, loads in flight, wait text)]}: sites where a counted
    `s_waitcnt vmcnt(N > 0)` releases consumers while at least one vector-memory store AND one load may still be in
    flight (issued since the last vmcnt(0), in either order: `load.. store.. wait` as bank_get had it, or
    `store.. (loop back) load.. wait` as dfa_points had it). The text is read straight-line, except that at a branch to
    an EARLIER local label the loop body is scanned once more with the counts that were live at the branch: a store
    issued at the end of one iteration is in flight while the next iteration's loads wait on a counted vmcnt
    (`loops=False`: the plain straight-line reading)."""
    hits = {}
    for kernel, body in _kernels(path):
        labels = {}
        for i, (_, line) in enumerate(body):
            m = LABEL.match(line)
            if m:
                labels[m.group(1)] = i
        found = []

        def walk(lo, hi, loads, stores, follow):
            i = lo
            while i < hi:
                ln, line = body[i]
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
                                found.append((ln, stores, loads, line.strip()))
                    elif follow:
                        b = BRANCH.match(line)
                        if b and b.group(1) in labels and labels[b.group(1)] <= i and (loads or stores):
                            walk(labels[b.group(1)], i, loads, stores, False)   # one more trip with the live counts
                i += 1
            return loads, stores

        walk(0, len(body), 0, 0, loops)
        if found:
            seen, uniq = set(), []
            for f in found:
                if f[0] not in seen:
                    seen.add(f[0])
                    uniq.append(f)
            hits[kernel] = uniq
    return hits


def banned_mfma(path):
    """{kernel: sorted mnemonics} of the double-K matrix instructions in an assembly listing or a disassembly."""
    hits = {}
    for kernel, body in _kernels(path):
        bad = sorted({m.group(0) for _, line in body for m in [BANNED_MFMA.search(line)] if m})
        if bad:
            hits[kernel] = bad
    return hits


def main():
    out = tempfile.mkdtemp(prefix="isa_scan_")
    total = 0
    for stem, s in sorted(compile_all(out, sys.argv[1:]).items()):
        for kernel, rows in scan(s).items():
            total += 1
            print(f"{stem}.hip: {kernel}: {len(rows)} site(s); first at line {rows[0][0]}: `{rows[0][3]}` with "
                  f"{rows[0][1]} store(s) and {rows[0][2]} load(s) possibly in flight")
        for kernel, bad in banned_mfma(s).items():
            print(f"{stem}.hip: {kernel}: BANNED matrix instruction(s) {bad}")
    print(f"{total} kernel(s) release consumers on a counted vmcnt wait with stores and loads in flight")


if __name__ == "__main__":
    main()
