#!/usr/bin/env python3
"""MI355X drop-in for the reference's Train/pseudo_labels_generator_spatio.py: same flags (lstc_vad_amd/cli_flags.json), same loop, HIP kernels
underneath (lstc_vad_amd/cli.py).  The module exports what the reference's module exports (generator :22, parser_arg :92),
so ``from Train.pseudo_labels_generator_spatio import generator`` written against the reference keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd import cli, losses  # noqa: E402

SCRIPT = "pseudo_labels_generator_spatio"


def parser_arg():
    """Train/pseudo_labels_generator_spatio.py:92: the script's flags parsed from sys.argv."""
    return cli.complete_args(SCRIPT)


def generator(args):
    """Train/pseudo_labels_generator_spatio.py:22: score every training video and write the pseudo-label file."""
    return cli.generate_pseudo_labels(SCRIPT, args=args)


if __name__ == "__main__":
    cli.main(SCRIPT)
