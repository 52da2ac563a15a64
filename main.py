#!/usr/bin/env python3
"""`python main.py ...` -- the reference's command line (main.py:1500-1670) on the MI355X hot path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import movae_amd  # noqa: E402,F401
from movae_amd.train import cli  # noqa: E402

if __name__ == "__main__":
    cli()
