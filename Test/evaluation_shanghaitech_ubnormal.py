#!/usr/bin/env python3
"""MI355X drop-in for the reference's Test/evaluation_shanghaitech_ubnormal.py: same flags (lstc_vad_amd/cli_flags.json), same loop, HIP kernels
underneath (lstc_vad_amd/cli.py).  The module exports what the reference's module exports (evaluation :24, parser_arg :100),
so ``from Test.evaluation_shanghaitech_ubnormal import evaluation`` written against the reference keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd import cli, losses  # noqa: E402

SCRIPT = "evaluation_shanghaitech_ubnormal"


def parser_arg():
    """Test/evaluation_shanghaitech_ubnormal.py:100: the script's flags parsed from sys.argv."""
    return cli.complete_args(SCRIPT)


def evaluation(args):
    """Test/evaluation_shanghaitech_ubnormal.py:24: frame-level AUC of a trained LTN on the test videos."""
    return cli.evaluate_cli(SCRIPT, args=args)


if __name__ == "__main__":
    cli.main(SCRIPT)
