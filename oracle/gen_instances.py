#!/usr/bin/env python3
"""Shim: the generators live in lorads_amd/instances.py (the bench needs them too)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lorads_amd.instances import *  # noqa: F401,F403,E402
from lorads_amd.instances import NAMED, main, write_sdpa  # noqa: F401,E402

if __name__ == "__main__":
    sys.exit(main())
