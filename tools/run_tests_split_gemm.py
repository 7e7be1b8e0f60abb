"""Measurement / test helper: run pytest (arguments passed through) with routes.gemm_split_fp16 flipped to the value
of --split (default: the other branch of the shipped default). usage: python tools/run_tests_split_gemm.py [--split 0|1] <pytest args>"""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import routes  # noqa: E402

args = sys.argv[1:]
value = not routes.DEFAULTS["gemm_split_fp16"]
if "--split" in args:
    i = args.index("--split")
    value = bool(int(args[i + 1]))
    del args[i:i + 2]
with routes.override(gemm_split_fp16=value):
    print("routes:", routes.R, flush=True)
    rc = pytest.main(args or ["tests", "-q", "-m", "gpu"])
sys.exit(rc)
