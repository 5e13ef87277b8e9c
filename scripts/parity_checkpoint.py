#!/usr/bin/env python3
"""Launcher of the real-weights parity hook: ``python scripts/parity_checkpoint.py --checkpoint <local file> [...]``.
The hook itself lives in tests/parity_checkpoint.py, because it runs the CPU oracle as its checker and only code under tests/
(plus smoke() and bench.py's cpu_baseline leg) may touch oracle/.  Nothing is downloaded; the checkpoint is read with
``weights_only=True`` / safetensors."""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.argv[0] = os.path.join(root, "tests", "parity_checkpoint.py")
runpy.run_path(sys.argv[0], run_name="__main__")
