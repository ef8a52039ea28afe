"""Shared bootstrap of the delta runners: put the package root on sys.path the way the LoRA runner does."""
import sys
from pathlib import Path

_PKG = Path(__file__).resolve().parents[2]
if str(_PKG) not in sys.path:
    sys.path.insert(0, str(_PKG))
