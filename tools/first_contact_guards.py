#!/usr/bin/env python3
"""`python tools/first_contact_guards.py --upstream /path/to/LongCat-Video [--checkpoint DIR] [--device cuda]`

Launcher of `tests/first_contact_guards.py` (the guards compare upstream with `oracle/`, which only test infrastructure
may use, so the implementation lives under `tests/`): one PASS / FAIL / INCONCLUSIVE line per row A1-A20 of `spec/dit.md`."""
import runpy
import sys
from pathlib import Path

sys.argv[0] = str(Path(__file__).resolve().parents[1] / "tests" / "first_contact_guards.py")
runpy.run_path(sys.argv[0], run_name="__main__")
