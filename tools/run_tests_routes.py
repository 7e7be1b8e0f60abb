"""Measurement / test helper: run pytest (arguments passed through) with route switches flipped:
python tools/run_tests_routes.py --route attention_split_fp16=1 [--route name=0|1 ...] <pytest args>"""
import contextlib
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import routes  # noqa: E402

args = sys.argv[1:]
stack = contextlib.ExitStack()
while "--route" in args:
    i = args.index("--route")
    k, v = args[i + 1].split("=")
    stack.enter_context(routes.override(**{k: bool(int(v))}))
    del args[i:i + 2]
with stack:
    print("routes:", routes.R, flush=True)
    rc = pytest.main(args or ["tests", "-q", "-m", "gpu"])
sys.exit(rc)
