#!/usr/bin/env python3
"""MI355X drop-in for the reference's Train/spatio_transformer_UCF.py: same flags (lstc_vad_amd/cli_flags.json), same loop, HIP kernels
underneath (lstc_vad_amd/cli.py).  The module exports what the reference's module exports (get_MIL_loss :20, train :35, parser_arg :156),
so ``from Train.spatio_transformer_UCF import get_MIL_loss`` written against the reference keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd import cli, losses  # noqa: E402

SCRIPT = "spatio_transformer_UCF"


def parser_arg():
    """Train/spatio_transformer_UCF.py:156: the script's flags parsed from sys.argv."""
    return cli.complete_args(SCRIPT)


def get_MIL_loss(args, y_pred):
    """Train/spatio_transformer_UCF.py:20-33: ``(loss, err, l1)`` of snippet scores ``y_pred`` [2*bs, part_num*part_len, 1] - bag score = max over
    parts of the part-mean, hinge over all bs x bs pairs, l1 on ``y_pred[bs:]``."""
    return losses.get_MIL_loss(args, y_pred, args.part_len)


def train(args):
    """Train/spatio_transformer_UCF.py:35: the training loop on a parsed (or caller-built) argument namespace."""
    return cli.train(SCRIPT, args=args)


if __name__ == "__main__":
    cli.main(SCRIPT)
