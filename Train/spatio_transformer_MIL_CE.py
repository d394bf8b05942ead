#!/usr/bin/env python3
"""MI355X drop-in for the reference's Train/spatio_transformer_MIL_CE.py: same flags (lstc_vad_amd/cli_flags.json), same loop, HIP kernels
underneath (lstc_vad_amd/cli.py).  The module exports what the reference's module exports (get_BCE_loss :23, get_CE_loss :28, get_MIL_loss :32, train :47, parser_arg :459),
so ``from Train.spatio_transformer_MIL_CE import get_MIL_loss`` written against the reference keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd import cli, losses  # noqa: E402

SCRIPT = "spatio_transformer_MIL_CE"


def parser_arg():
    """Train/spatio_transformer_MIL_CE.py:459: the script's flags parsed from sys.argv."""
    return cli.complete_args(SCRIPT)


def get_BCE_loss(args, outputs, labs):
    """Train/spatio_transformer_MIL_CE.py:23-26: weighted BCE of part scores [2*bs, part_num] against soft labels [2*bs, part_num, 2]."""
    return losses.get_BCE_loss(args, outputs, labs)


def get_CE_loss(args, outputs, labs):
    """Train/spatio_transformer_MIL_CE.py:28-30."""
    return losses.get_CE_loss(args, outputs, labs)


def get_MIL_loss(args, y_pred, part_len):
    """Train/spatio_transformer_MIL_CE.py:32-44: as the STN loss with the part length passed in; l1 on the first-dim slice ``y_pred[bs:]`` of the
    [2*bs*part_num*part_len, 1] scores (the co-teaching quirk)."""
    return losses.get_MIL_loss(args, y_pred, part_len)


def train(args):
    """Train/spatio_transformer_MIL_CE.py:47: the training loop on a parsed (or caller-built) argument namespace."""
    return cli.train(SCRIPT, args=args)


if __name__ == "__main__":
    cli.main(SCRIPT)
