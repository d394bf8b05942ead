"""Reference import path ``utils.utils``: the helpers the Train/Test scripts actually call."""
import os
import random

import numpy as np
import torch


def set_seeds(seed):
    """utils/utils.py:107-116 (the cudnn switches have no HIP counterpart: the HIP kernels here are deterministic except for
    the documented f32-atomic accumulation order in split-K weight gradients)."""
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
