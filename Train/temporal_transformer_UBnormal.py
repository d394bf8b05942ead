#!/usr/bin/env python3
"""MI355X drop-in for the reference's Train/temporal_transformer_UBnormal.py: same flags (lstc_vad_amd/cli_flags.json), same loop, HIP kernels
underneath (lstc_vad_amd/cli.py).  The module exports what the reference's module exports (get_CE_loss :22, get_MIL_loss :26, train :40, parser_arg :261),
so ``from Train.temporal_transformer_UBnormal import get_MIL_loss`` written against the reference keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lstc_vad_amd import cli, losses  # noqa: E402

SCRIPT = "temporal_transformer_UBnormal"


def parser_arg():
    """Train/temporal_transformer_UBnormal.py:261: the script's flags parsed from sys.argv."""
    return cli.complete_args(SCRIPT)


def get_CE_loss(args, outputs, labs):
    """Train/temporal_transformer_UBnormal.py:22-24: ``F.cross_entropy`` of the softmax OUTPUTS against soft targets."""
    return losses.get_CE_loss(args, outputs, labs)


def get_MIL_loss(args, y_pred):
    """Train/temporal_transformer_UBnormal.py:26-38: ``(loss, err, l1)`` of part scores ``y_pred`` [2*bs*part_num] - bag score = max over parts; l1 on the
    flat slice ``y_pred[bs:]`` exactly as upstream spells it."""
    return losses.get_MIL_loss(args, y_pred, 1)


def train(args):
    """Train/temporal_transformer_UBnormal.py:40: the training loop on a parsed (or caller-built) argument namespace."""
    return cli.train(SCRIPT, args=args)


if __name__ == "__main__":
    cli.main(SCRIPT)
