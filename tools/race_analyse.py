"""Which operand did bank_get see wrong? Evaluates csrc/bank.hip:bank_get_kernel on the host from the operands
tools/pipe_race.py --trace --dump saved for the first bad frame, under each hypothesis, and reports which one
reproduces the bad output.

    python tools/race_analyse.py gpurun_out/race_dump.pt
"""
import sys

import numpy as np
import torch


def bank_get(stored, T, dt):
    """Host statement of bank_get_kernel for one stream (stored [n, 11], T [4, 4], dt scalar)."""
    a = stored.astype(np.float64)
    t = -float(dt)
    v = a[:, 8:11]
    c = a[:, 0:3] - v * t
    m = T.astype(np.float64)
    out = np.empty_like(a)
    out[:, 0:3] = c @ m[:3, :3].T + m[:3, 3]
    out[:, 3:6] = a[:, 3:6]
    s, co = a[:, 6], a[:, 7]
    out[:, 6] = m[0, 0] * co + m[0, 1] * s
    out[:, 7] = m[1, 0] * co + m[1, 1] * s
    out[:, 8:11] = v @ m[:3, :3].T
    return out


def main():
    d = torch.load(sys.argv[1], weights_only=True)
    n = lambda t: t.numpy()  # noqa: E731
    pipe, plain, pprev = d["pipe"], d["plain"], d["pipe_prev"]
    good, bad = n(plain["bank_get.out.3"])[0], n(pipe["bank_get.out.3"])[0]
    late = n(pipe["bank_get.out3_late"])[0] if "bank_get.out3_late" in pipe else None
    stored, T, dt = n(pipe["bank_get.in.stored"])[0], n(pipe["bank_get.in.T_dt.0"])[0], n(pipe["bank_get.in.T_dt.1"])[0]
    print(f"frame {d['frame']}: dt = {dt}, |T translation| = {np.linalg.norm(T[:3, 3]):.4f}")
    diff = np.abs(bad - good)
    rows = np.nonzero(diff.max(1) > 1e-6)[0]
    print(f"rows differing: {len(rows)} of {len(good)}; first {rows[:12].tolist()} last {rows[-5:].tolist()}")
    print("max |bad - good| per column:", np.array2string(diff.max(0), precision=4))
    if late is not None:
        print(f"late clone of the same tensor vs early bad clone: max diff {np.abs(late - bad).max():.3e}; vs good {np.abs(late - good).max():.3e}")
    hyp = {
        "as given (sanity, should be 0 vs good)": bank_get(stored, T, dt),
        "dt read as 0": bank_get(stored, T, 0.0),
        "T = previous frame's": bank_get(stored, n(pprev["bank_get.in.T_dt.0"])[0], dt) if "bank_get.in.T_dt.0" in pprev else None,
        "T = identity": bank_get(stored, np.eye(4, dtype=np.float32), dt),
        "stored = previous frame's cached_anchor": bank_get(n(pprev["bank_get.in.stored"])[0], T, dt) if "bank_get.in.stored" in pprev else None,
        "stored velocity columns read as 0": bank_get(np.concatenate([stored[:, :8], np.zeros_like(stored[:, 8:])], 1), T, dt),
        "output = previous frame's output (kernel did not run / clobbered by it)": n(pprev["bank_get.out.3"])[0].astype(np.float64) if "bank_get.out.3" in pprev else None,
        "output = input (copied, not warped)": stored.astype(np.float64),
    }
    for name, out in hyp.items():
        if out is None:
            continue
        print(f"  {name:75s} max|.-good| = {np.abs(out - good).max():.3e}   max|.-bad| = {np.abs(out - bad).max():.3e}")
    # row-wise: is each bad row equal to SOME row of a hypothesis output?
    for name in ("stored = previous frame's cached_anchor", "output = previous frame's output (kernel did not run / clobbered by it)"):
        out = hyp.get(name)
        if out is None:
            continue
        hit = sum(bool((np.abs(out - bad[r]).max(1) < 1e-5).any()) for r in rows)
        print(f"  bad rows found somewhere in [{name}]: {hit}/{len(rows)}")


if __name__ == "__main__":
    main()
