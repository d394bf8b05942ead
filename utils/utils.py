"""Reference import path ``utils.utils``: the helpers the Train/Test scripts actually call."""
import os
import random

import numpy as np
import torch


def set_seeds(seed):
    """utils/utils.py:107-116 (the cudnn switches have no HIP counterpart: the HIP kernels here are deterministic except for
    dropout, whose masks come from a counter-based hash seeded from ``torch.manual_seed``; split-K weight gradients add
    their partial sums in a fixed order, so runs with equal seeds are bit-identical)."""
    print('set seed {}'.format(seed))
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def mkdir(dir):
    os.makedirs(dir, exist_ok=True)


def get_video_names(txt_path, abnormal=True, normal=True):
    """utils/utils.py:25-38: names from an SHT list (``name,label[,frames]``), filtered by class."""
    names = []
    for line in open(txt_path, 'r').readlines():
        parts = line.strip().split(',')
        is_abnormal = int(parts[1]) == 1
        if (is_abnormal and abnormal) or (not is_abnormal and normal):
            names.append(parts[0])
    return names


def weights_normal_init(model, dev=0.01):
    """utils/utils.py:134-150 (imported by the reference's model files, never called by its scripts): Kaiming-normal
    Linear / Conv2d weights, zero Linear biases; accepts a module or a list of modules.  Writes go through ``.data``
    upstream, which does not bump the parameters' version counters, so the packed-weight cache is invalidated here."""
    from torch import nn
    import lstc_vad_amd
    for root in (model if isinstance(model, list) else [model]):
        for m in root.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)
                if isinstance(m, nn.Linear) and m.bias is not None:
                    with torch.no_grad():
                        m.bias.zero_()
    lstc_vad_amd.bump_weight_epoch()
