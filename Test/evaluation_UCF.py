#!/usr/bin/env python3
"""MI355X drop-in for the reference's Test/evaluation_UCF.py: same flags (lstc_vad_amd/cli_flags.json), same loop, HIP kernels
underneath (lstc_vad_amd/cli.py).  The module exports what the reference's module exports (evaluation :23, parser_arg :90),
so ``from Test.evaluation_UCF import evaluation`` written against the reference keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd import cli, losses  # noqa: E402

SCRIPT = "evaluation_UCF"


def parser_arg():
    """Test/evaluation_UCF.py:90: the script's flags parsed from sys.argv."""
    return cli.complete_args(SCRIPT)


def evaluation(args):
    """Test/evaluation_UCF.py:23: frame-level AUC of a trained LTN on the test videos."""
    return cli.evaluate_cli(SCRIPT, args=args)


if __name__ == "__main__":
    cli.main(SCRIPT)
