#!/usr/bin/env python3
"""MI355X drop-in for the reference's Test/evaluation_shanghaitech_ubnormal.py: same flags, batched HIP inference.  See lstc_vad_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main("evaluation_shanghaitech_ubnormal")
