"""Command-line surface of the reference's Train/*.py scripts on top of the HIP training step.

Flag names, types and defaults come from ``cli_flags.json`` — the surface extracted from the reference's own
``parser_arg()`` functions (author-machine absolute path defaults blanked) — so every reference
command line parses unchanged.  Additions: ``--synthetic`` (default on when no ``--dataset_path`` is given: the feature
files are external downloads; a given HDF5 / .npz / directory archive is read through ``lstc_vad_amd.archive``), ``--synthetic_pairs``, ``--steps`` (stop after N
optimisation steps), ``--log_dir`` (the reference hard-codes ``/data/ssy/...`` and crashes elsewhere,
utils/utils.py:152-173).  ``--gpu`` selects the device through ``HIP_VISIBLE_DEVICES`` (reference:
``CUDA_VISIBLE_DEVICES``, e.g. Train/temporal_transformer_shanghaitech.py:328).

Loop structure follows the reference: epochs over a pair loader, one step per batch, the same log line formats, an
evaluation every ``--inter_epoch`` epochs over the test videos AND (where the script does) the training videos, and the
script's own checkpoint rule (``SELECTION``: e.g. Train/temporal_transformer_shanghaitech.py:186-252 saves when the
TRAIN-set AUC improves and exceeds ``--save_threshold``, file name ``<prefix>temporal_model_oneCrop_<type>_<str(auc)>``).
Under ``torchrun`` ``--batch_size`` stays the global pair count; rank r owns pairs [r*bs/N, (r+1)*bs/N) of every batch.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import random
import sys
from argparse import Namespace

_FLAGS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_flags.json")))
# timing record of the last train / generate_pseudo_labels / evaluate_cli call of this process (bench.py's coteach_loop reads it):
# steps and seconds between device synchronisations, so the figures are kernel time of the stage, not process start-up
LAST_RUN = {}
_TYPES = {"int": int, "float": float, "str": str}

SCRIPTS = {
    # script name -> (mode, head kind, prefix used for the model flags)
    "spatio_transformer_shanghaitech": ("STN", "regressor", ""),
    "spatio_transformer_UCF": ("STN", "regressor", ""),
    "spatio_transformer_UBnormal": ("STN", "regressor", ""),
    "temporal_transformer_shanghaitech": ("LTN", "classifier", ""),
    "temporal_transformer_UCF": ("LTN", "classifier", ""),
    "temporal_transformer_UBnormal": ("LTN", "classifier", ""),
    "spatio_transformer_MIL_CE": ("STN_MIL_CE", "regressor", "spatio_"),
}


SELECTION = {
    # script -> how it picks checkpoints (file:line of the rule).  train: flag naming the archive the TRAINING videos are
    # scored from (None: the script never computes a train AUC / sets it to 0); on: which AUC gates the save; thr: the save
    # also needs AUC > --save_threshold; prefix / typ: --saved_prefix and --type appear in the file name; join: os.path.join
    # instead of string concatenation; head: file-name stem of the head's checkpoint.
    "spatio_transformer_shanghaitech": dict(train="train_dataset", on="train", thr=True, enc="spatio", head="regression",
                                            prefix=True, typ=True, join=False),        # :145-190
    "temporal_transformer_shanghaitech": dict(train="dataset_path", on="train", thr=True, enc="temporal", head="classifier",
                                              prefix=True, typ=True, join=False),      # :186-252
    "spatio_transformer_UBnormal": dict(train=None, on="train", thr=True, enc="spatio", head="regression",
                                        prefix=False, typ=True, join=True),            # :128-149 (auc_train = 0: never saves)
    "temporal_transformer_UBnormal": dict(train="dataset_path", on="train", thr=True, enc="temporal", head="classifier",
                                          prefix=False, typ=True, join=False),         # :193-254
    "spatio_transformer_UCF": dict(train=None, on="test", thr=True, enc="spatio", head="regression",
                                   prefix=False, typ=False, join=False),               # :137-150
    "temporal_transformer_UCF": dict(train=None, on="test", thr=True, enc="temporal", head="classifier",
                                     prefix=False, typ=True, join=False),              # :173-186
    "spatio_transformer_MIL_CE": dict(train="dataset_path", on="train", thr=False, enc="spatio", head="regression",
                                      prefix=True, typ=True, join=False),              # :343-365 (UCF / UBnormal: auc_train = 0)
}


def checkpoint_names(script, args, auc):
    """(encoder path, head path) exactly as the script spells them, e.g. Train/temporal_transformer_shanghaitech.py:242-247:
    ``model_save_dir + saved_prefix + "temporal_model_oneCrop_" + type + "_" + str(auc_train)``."""
    rule = SELECTION[script]
    tail = (str(args.type) + "_" if rule["typ"] else "") + str(auc)
    pre = (getattr(args, "saved_prefix", None) or "") if rule["prefix"] else ""
    d = getattr(args, "model_save_dir", "") or ""
    cat = (lambda n: os.path.join(d, n)) if rule["join"] else (lambda n: d + n)
    return cat(pre + rule["enc"] + "_model_oneCrop_" + tail), cat(pre + rule["head"] + "_model_oneCrop_" + tail)


class Selector:
    """Best-AUC bookkeeping + save decision of one Train script."""

    def __init__(self, script, args):
        self.script, self.args, self.rule = script, args, SELECTION[script]
        thr = float(getattr(args, "save_threshold", 0.0) or 0.0)
        self.best_test = thr if script == "spatio_transformer_MIL_CE" else 0.0      # MIL_CE.py:104
        self.best_train, self.best_test_epoch, self.best_train_epoch, self.thr = 0.0, 0, 0, thr

    def update(self, epoch, auc_test, auc_train):
        """Returns the AUC to save under (or None) and the log lines upstream prints."""
        r, save = self.rule, None
        if self.script == "spatio_transformer_MIL_CE":
            if auc_train > self.best_train:
                self.best_train, self.best_train_epoch, save = auc_train, epoch, auc_train
            if auc_test > self.best_test:
                self.best_test, self.best_test_epoch = auc_test, epoch
            return save, ['best_train_AUC {} at epoch {} now train_AUC is {}'.format(self.best_train, self.best_train_epoch, auc_train),
                          'best_test_AUC {} at epoch {} now test_AUC is {}'.format(self.best_test, self.best_test_epoch, auc_test)]
        if auc_test > self.best_test:
            self.best_test, self.best_test_epoch = auc_test, epoch
            if r["on"] == "test" and auc_test > self.thr:
                save = auc_test
        if r["on"] == "train" and auc_train > self.best_train:
            self.best_train, self.best_train_epoch = auc_train, epoch
            if auc_train > self.thr:
                save = auc_train
        if r["on"] == "test":
            return save, ['best_test_AUC {} at epoch {} now test_AUC is {}'.format(self.best_test, self.best_test_epoch, auc_test)]
        return save, ['best_test_AUC {} at epoch {} now test_AUC is {} \nbest_train_AUC {} at epoch {} now train_AUC is {}'.format(
            self.best_test, self.best_test_epoch, auc_test, self.best_train, self.best_train_epoch, auc_train)]


def build_parser(script: str) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog=script + ".py", description=f"MI355X drop-in for the reference Train/{script}.py")
    for flag, kind, default in _FLAGS[script]:
        if kind == "flag":
            p.add_argument(flag, action="store_true", default=bool(default))
        else:
            p.add_argument(flag, type=_TYPES.get(kind, str), default=default)
    if not any(f[0] == "--encoder_weight_init" for f in _FLAGS[script]):
        # the reference's generators read this flag without defining it (AttributeError upstream)
        p.add_argument("--encoder_weight_init", action="store_true")
    p.add_argument("--synthetic", action="store_true", help="train on lstc_vad_amd.data.SyntheticVideos")
    p.add_argument("--synthetic_pairs", type=int, default=0, help="normal/abnormal pairs in the synthetic set (default 2*batch_size)")
    p.add_argument("--steps", type=int, default=0, help="stop after this many optimisation steps (0 = run all epochs)")
    p.add_argument("--save_final", type=str, default="", help="path prefix: write <prefix>encoder.ckpt / <prefix>head.ckpt after the last step")
    p.add_argument("--log_dir", type=str, default="", help="log / checkpoint directory (default: ./log/<dataset>)")
    p.add_argument("--compute_dtype", type=str, default=os.environ.get("LSTC_COMPUTE_DTYPE", "fp32"), choices=["fp32", "f32x3", "bf16"],
                   help="GEMM arithmetic: fp32 = exact-f32 MFMA (reference numerics); f32x3 = f32-accurate products on the f16 "
                        "matrix cores (2x faster); bf16 = bf16 matrix cores on f32 storage (BASELINE configs 3 / 5)")
    return p


def complete_args(script: str, args=None, argv=None) -> Namespace:
    """The argument namespace of ``script``: parsed from ``argv`` (``None`` = sys.argv), or - when a caller hands over the
    Namespace it built itself, as code written against the reference's ``train(args)`` / ``generator(args)`` /
    ``evaluation(args)`` does - that namespace on top of the script's flag defaults (so the flags this build adds, e.g.
    ``--steps``, ``--compute_dtype``, need not be present)."""
    parser = build_parser(script)
    if args is None:
        return parser.parse_args(argv)
    full = parser.parse_args([])
    for k, v in vars(args).items():
        setattr(full, k, v)
    return full


def namespace_to_argv(script: str, args: Namespace) -> list:
    """The command line that parses back to ``args`` (flags of ``build_parser(script)`` only): what the rank processes of a
    ``--data_parallel`` run are started with when ``train(args)`` was handed a Namespace instead of sys.argv."""
    argv = []
    for act in build_parser(script)._actions:
        if not act.option_strings or act.dest == "help" or not hasattr(args, act.dest):
            continue
        v = getattr(args, act.dest)
        if isinstance(act, argparse._StoreTrueAction):
            if v:
                argv.append(act.option_strings[0])
        elif v is not None:
            argv += [act.option_strings[0], repr(v) if isinstance(v, float) else str(v)]
    return argv


def _launch_data_parallel(script: str, args: Namespace):
    """``--data_parallel --gpu 0,1,2,3`` IS the multi-GPU run upstream (``nn.DataParallel`` over every visible device,
    Train/temporal_transformer_shanghaitech.py:76-78, ``--gpu`` -> CUDA_VISIBLE_DEVICES :328).  Here: one rank per listed GPU,
    started by lstc_vad_amd.launch.launch_ranks BEFORE this process touches a GPU (the children set the devices; the parent only
    relays rank 0's output and watches the ranks).  The ranks are started with the command line that parses back to ``args``, so
    ``train(args)`` on a caller-built Namespace launches the same job as the script's own command line.  Returns None when this
    process should do the work itself (one device listed, or already a rank of a launched / torchrun job), else ("done", what rank
    0 handed back)."""
    if not getattr(args, "data_parallel", False) or "WORLD_SIZE" in os.environ:
        return None
    from .launch import launch_ranks, parse_devices
    devs = parse_devices(getattr(args, "gpu", "0"))
    if len(devs) < 2:
        return None
    n_ranks = len(devs)
    if os.environ.get("LSTC_SHARE_DEVICE") == "1":          # test boxes with ONE GPU: --gpu 0,0 = two ranks on device 0 (gloo collectives)
        devs = sorted(set(devs), key=devs.index)
    bs = getattr(args, "batch_size", None)
    if script in SCRIPTS and bs is not None and bs % n_ranks:
        raise SystemExit(f"--batch_size {bs} (pairs of the global batch) does not split over the {n_ranks} GPUs of --gpu {args.gpu}")
    folder = "Test" if script.startswith("evaluation") else "Train"
    child = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), folder, script + ".py")
    rc = launch_ranks(n_ranks, namespace_to_argv(script, args), script=child,
                      rank_timeout_s=float(os.environ.get("LSTC_RANK_TIMEOUT_S", "0")), relay="all", devices=devs, tag=script,
                      # a rank hung in the rendezvous / RCCL communicator set-up would block the parent forever: 15 minutes to join
                      # the process group (lstc_vad_amd.launch.mark_rank_ready), no limit on the training run itself
                      init_timeout_s=float(os.environ.get("LSTC_INIT_TIMEOUT_S", "900")))
    if rc:
        raise SystemExit(rc)
    res = launch_ranks.last_result
    return ("done", float(res) if res not in (None, "None") else None)


def _init_ranks(torch):
    """(rank, world, device) of this process; joins the RCCL process group when it is one of several ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LSTC_SHARE_DEVICE") == "1":          # every rank on device 0 (one-GPU test boxes; RCCL refuses that, so gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LSTC_DIST_BACKEND", "nccl")         # nccl = RCCL over xGMI
        if not dist.is_initialized():
            dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        from .launch import mark_rank_ready
        mark_rank_ready()
    return rank, world, dev


class _LateLog:
    """Step scalars reach the log ONE STEP LATE on rank 0: step k's five scalars are copied to pinned host memory behind step k's
    kernels and read when step k + 1 has been queued, so the host never waits for the step it has just launched (the reference
    formats CUDA scalars right after the step, Train/temporal_transformer_shanghaitech.py:143 - a device sync per step; here rank
    0 runs ahead like the other ranks).  ``flush()`` before anything that must see the log complete (evaluation, exit)."""

    def __init__(self, torch, emit):
        self.torch, self.emit = torch, emit
        self.buf = torch.empty(5, dtype=torch.float32).pin_memory()
        self.pending = None

    def push(self, it, epoch, sc):
        self.flush()
        self.buf.copy_(sc.detach().reshape(-1)[:5], non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record()
        self.pending = (it, epoch, ev)

    def flush(self):
        if self.pending is None:
            return
        it, epoch, ev = self.pending
        self.pending = None
        ev.synchronize()
        self.emit(it, epoch, *(float(x) for x in self.buf.tolist()))


def _apply_compute_dtype(args):
    from . import functional as Fn
    Fn.set_compute_dtype(getattr(args, "compute_dtype", "fp32"))


def _get(args, name, prefix="", default=None):
    return getattr(args, prefix + name, getattr(args, name, default))


def _logger(args, script):
    log_dir = args.log_dir or os.path.join("log", str(getattr(args, "dataset", "run")))
    os.makedirs(log_dir, exist_ok=True)
    lg = logging.getLogger(script)
    lg.setLevel(logging.INFO)
    lg.handlers.clear()
    fmt = logging.Formatter("%(asctime)s: %(message)s")
    for h in (logging.StreamHandler(sys.stderr), logging.FileHandler(os.path.join(log_dir, script + ".log"))):
        h.setFormatter(fmt)
        lg.addHandler(h)
    return lg, log_dir


def train(script: str, argv=None, args=None):
    mode, head_kind, pre = SCRIPTS[script]
    args = complete_args(script, args, argv)
    launched = _launch_data_parallel(script, args)
    if launched is not None:
        return launched[1]
    if "LOCAL_RANK" not in os.environ:                      # single process: honour --gpu like the reference does
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(getattr(args, "gpu", 0)))
    import numpy as np
    import torch
    import torch.distributed as dist

    from .data import SyntheticVideos
    from .engine import TrainStep
    from .metrics import roc_auc
    from .models import Classifier, Encoder, Regressor

    seed = int(getattr(args, "seed", 0))                    # utils/utils.py:107-116
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: the LSTC_VAD training path here is MI355X-only (no CPU fallback)")
    _apply_compute_dtype(args)
    rank, world, dev = _init_ranks(torch)
    logger, log_dir = _logger(args, script) if rank == 0 else (logging.getLogger("null"), "")
    if getattr(args, "data_parallel", False) and rank == 0:
        logger.info("--data_parallel: %d rank(s), one process per GPU (--gpu %s), RCCL gradient all-reduce" % (world, getattr(args, "gpu", "")))

    part_len = _get(args, "part_len", pre)
    d_model = args.d_model
    enc = Encoder(n_layers=_get(args, "n_layers", pre), n_head=_get(args, "n_head", pre), d_k=_get(args, "d_k", pre),
                  d_v=_get(args, "d_v", pre), d_model=d_model, d_inner=_get(args, "n_hidden", pre),
                  MHA_attn_dropout=_get(args, "MHA_attn_dropout", pre), MHA_fc_dropout=_get(args, "MHA_fc_dropout", pre),
                  MHA_layerNorm=_get(args, "MHA_layerNorm", pre), FFN_dropout=_get(args, "FFN_dropout", pre),
                  FFN_layerNorm=_get(args, "FFN_layerNorm", pre), position_dropout=_get(args, "position_dropout", pre, 0.1),
                  weight_init=_get(args, "encoder_weight_init", pre, False),
                  position_encoding=_get(args, "position_encoding", pre, False),
                  CLS_learned=_get(args, "CLS_learned", pre, False),
                  max_position_tokens=_get(args, "max_position_tokens", pre, 100),
                  relative_pe=bool(getattr(args, "relative_position_encoding", False)) and mode == "LTN",
                  window_size=getattr(args, "window_size", 4), window_depth=part_len if mode == "LTN" else 3,
                  conv_patch=getattr(args, "conv_patch", False),
                  relative_pe_2D=bool(getattr(args, "relative_pe_2D", False)) and mode != "LTN",
                  input_layerNorm=bool(getattr(args, "input_layerNorm", False)))
    if head_kind == "classifier":
        head = Classifier(d_model, args.classifier_dropout, weight_init=args.classifier_weight_init)
        lr_head = args.lr_classifier
    else:
        head = Regressor(d_model, _get(args, "regressor_dropout", "", 0.6),
                         weight_init=_get(args, "regressor_weight_init", "", False))
        lr_head = _get(args, "lr_regressor", "", 1e-2)
    strip = lambda sd: {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
    if getattr(args, "load_model", False):          # state_dict files with the reference's key names (non-strict)
        if script == "spatio_transformer_MIL_CE":    # Train/spatio_transformer_MIL_CE.py:72-90: --spatio_model_path / --regression_model_path
            enc_path, head_path = args.spatio_model_path, args.regression_model_path
        else:
            enc_path = getattr(args, "load_temporal_model_path", None) or getattr(args, "load_spatio_model_path", "")
            head_path = args.load_classifier_model_path
        enc.load_state_dict(strip(torch.load(enc_path, map_location="cpu")), False)
        head.load_state_dict(strip(torch.load(head_path, map_location="cpu")), False)
    enc, head = enc.to(dev).train(), head.to(dev).train()

    if args.batch_size % world:
        raise SystemExit(f"--batch_size {args.batch_size} (pairs of the global batch) does not split over {world} ranks")
    # --batch_size keeps the reference's meaning (pairs per optimizer step, globally); each rank owns batch_size/world of them
    step_args = Namespace(batch_size=args.batch_size // world, part_num=args.part_num, part_len=part_len, n_patch=args.n_patch,
                          lambda_1=args.lambda_1, lambda_MIL=getattr(args, "lambda_MIL", 1.0),
                          lambda_CE=getattr(args, "lambda_CE", 0.0), lambda_BCE=getattr(args, "lambda_BCE", 1.0),
                          lambda_normal=getattr(args, "lambda_normal", 0.2), lambda_abnormal=getattr(args, "lambda_abnormal", 2.0),
                          temporal_only=getattr(args, "temporal_only", False), clip_grad=getattr(args, "clip_grad", False))
    ts = TrainStep(step_args, mode, enc, head, _get(args, "lr_encoder", pre, 1e-4), lr_head, args.weight_decay)

    if script == "spatio_transformer_MIL_CE":
        # round 0 (the only round: `for round_i in range(1)`, Train/spatio_transformer_MIL_CE.py:113-116) trains the STN on
        # --spatio_pseudo_path, the labels the temporal model produced for it - read as given, no suffix added
        ppath = getattr(args, "spatio_pseudo_path", None)
    else:
        ppath = getattr(args, "pseudo_labels_path", None)
    if ppath in ("", "None"):
        ppath = None
    real = (not args.synthetic) and bool(getattr(args, "dataset_path", ""))
    if real:
        data, eval_fn = _real_data(script, args, mode, part_len, ppath, dev, rank, world, enc, head)
    else:
        n_pairs = args.synthetic_pairs or 2 * args.batch_size
        thr = None if mode == "STN" else 0.65
        pseudo = np.load(ppath, allow_pickle=True).tolist() if ppath and os.path.exists(ppath) else None
        data = SyntheticVideos(n_pairs, args.batch_size, args.part_num, part_len, args.n_patch, d_model, dev,
                               seed=seed, sample=args.sample, pseudo_threshold=thr, pseudo_labels=pseudo, rank=rank, world=world)
        eval_fn = lambda: (evaluate(enc, head, mode, data, part_len, roc_auc),
                           evaluate(enc, head, mode, data, part_len, roc_auc, train=True) if SELECTION[script]["train"] else 0)
    epochs = int(_get(args, "epochs", pre, 1))
    inter = int(getattr(args, "inter_epoch", 10))
    sel = Selector(script, args)
    it = 0

    def emit(it_, epoch_, loss, mil, err, l1, aux):
        if mode == "LTN":
            logger.info('[{}/{}]: loss {:.4f}, MIL_loss {:.4f}, CE_loss {:.4f} MIL_l1 {:.4f}'.format(it_, epoch_, loss, mil, aux, l1))
        elif mode == "STN":
            logger.info('[{}/{}]: loss {:.4f}, err {:.4f}, l1 {:.4f}'.format(it_, epoch_, loss, err, l1))
        else:
            logger.info('Round 0 [{}/{}]: spatio_loss {:.4f}, CE_loss {:.4f}, MIL_loss {:.4f}, err {:.4f}, l1 {:.4f}'.format(
                it_, epoch_, loss, aux, mil, err, l1))
    late = _LateLog(torch, emit) if rank == 0 else None
    import time as _time
    t_first, t_eval, t_save = None, 0.0, 0.0
    for epoch in range(epochs):
        for norm_feats, norm_labs, abnorm_feats, abnorm_labs in data:
            sc = ts.step(norm_feats, abnorm_feats, abnorm_labs)
            if world > 1:
                # a rank's five scalars are its CONTRIBUTIONS to the global loss terms (local sums over global counts): the log line
                # shows the batch's loss like a single-process run does - one 20-byte sum-all-reduce on the launch stream, no host sync
                sc = sc.clone()
                from .dist import all_reduce_sum
                all_reduce_sum(sc)
            if late is not None:
                late.push(it, epoch, sc)          # the PREVIOUS step's line is written now: no sync on the step just queued
            it += 1
            if it == 1:                           # LAST_RUN: steady-state step rate from the end of the first step on
                torch.cuda.synchronize(); t_first = _time.perf_counter()
            if args.steps and it >= args.steps:
                break
        if late is not None:
            late.flush()
        data.shuffle_keys()
        last = script == "spatio_transformer_MIL_CE" and epoch == epochs - 1          # MIL_CE.py:218 also evaluates the last epoch
        if epoch % inter == 0 or last:
            # EVERY rank evaluates: the videos of the test (and train) list are sharded over the ranks (pipeline._ScoreBoard) and
            # every rank ends up with the same AUCs, so no rank sits in a collective while another scores the whole list; rank 0
            # alone logs and writes the checkpoints
            torch.cuda.synchronize(); t_e = _time.perf_counter()
            auc_test, auc_train = eval_fn()
            torch.cuda.synchronize(); t_eval += _time.perf_counter() - t_e
            enc.train(); head.train()
            save_auc, lines = sel.update(epoch, auc_test, auc_train)
            if rank != 0:
                save_auc, lines = None, []
            if save_auc is not None:
                t_s = _time.perf_counter()
                logger.info("saving model......")
                enc_path, head_path = checkpoint_names(script, args, save_auc)
                if not (getattr(args, "model_save_dir", "") or ""):                   # no directory given: next to the log
                    enc_path, head_path = os.path.join(log_dir, enc_path), os.path.join(log_dir, head_path)
                os.makedirs(os.path.dirname(enc_path) or ".", exist_ok=True)
                torch.save(enc.state_dict(), enc_path)
                torch.save(head.state_dict(), head_path)
                logger.info("save complete.")
                t_save += _time.perf_counter() - t_s
            for ln in lines:
                logger.info(ln)
            if rank == 0:
                logger.info('======================================================================================')
        if args.steps and it >= args.steps:
            break
    torch.cuda.synchronize()
    LAST_RUN.clear()
    LAST_RUN.update(script=script, steps=it, steady_steps=max(it - 1, 0),
                    steady_s=(_time.perf_counter() - t_first - t_eval - t_save) if t_first is not None else 0.0, eval_s=t_eval,
                    save_s=t_save,
                    snippets_per_step=2 * args.batch_size * args.part_num * part_len, world=world)
    if rank == 0 and getattr(args, "save_final", ""):
        torch.save(enc.state_dict(), args.save_final + "encoder.ckpt")
        torch.save(head.state_dict(), args.save_final + "head.ckpt")
    if script == "spatio_transformer_MIL_CE" and real and getattr(args, "temporal_pseudo_path", ""):
        # end of the round (Train/spatio_transformer_MIL_CE.py:392-414): RE-LOAD --spatio_model_path / --regression_model_path
        # (the files, not the weights just trained - upstream expects the user's best checkpoint there) and write the next
        # temporal model's pseudo labels, score > --threshold else 0, to --temporal_pseudo_path (np.save appends .npy)
        from . import bump_weight_epoch
        from .pipeline import generate_pseudo_labels as gen
        if world > 1:
            # the paths may name checkpoints rank 0 wrote in THIS run (after the evaluation all-reduce, or --save_final): no rank
            # may read them before rank 0's torch.save has returned - it would score with a partial file and the sum-all-reduce of
            # the label scores would mix two sets of weights
            dist.barrier()
        enc.load_state_dict(strip(torch.load(args.spatio_model_path, map_location="cpu")), False)
        head.load_state_dict(strip(torch.load(args.regression_model_path, map_location="cpu")), False)
        bump_weight_epoch()
        dataset = str(getattr(args, "dataset", "SHT"))
        gen(enc.eval(), head.eval(), "STN", dataset if dataset in ("UCF", "UBnormal") else "SHT", args.dataset_path, args.training_txt,
            args.threshold, part_len=1, n_patch=args.n_patch, d_model=d_model, segment_len=args.segment_len,
            out_path=args.temporal_pseudo_path, rank=rank, world=world)      # sharded over the ranks; rank 0 writes the file
        enc.train(); head.train()
        if rank == 0:
            logger.info("temporal pseudo label generation finished.")
    if world > 1:
        dist.destroy_process_group()
    best = sel.best_test if sel.rule["on"] == "test" else sel.best_train
    if rank == 0 and os.environ.get("LSTC_LAUNCHED") == "1":
        print(f"LSTC_RESULT {best!r}", flush=True)          # handed back to the --data_parallel parent (launch.launch_ranks)
    return best


class _HostPairs:
    """Fallback for a set that does not fit in HBM (or LSTC_HOST_FEED=1): items collated on the host (``DataLoader(batch_size, drop_last=True,
    num_workers=0)`` order), staged through pinned memory, copied to the device."""

    def __init__(self, dataset, batch_size, device, rank=0, world=1, streams=None):
        """``batch_size`` = pairs of the global batch; rank ``r`` keeps items ``[r*bs/world, (r+1)*bs/world)`` of each.  Every
        rank draws ALL items (the samplers consume ``np.random`` / ``random`` per item), so the union of the shards is the
        single-process batch; the price is ``world`` times the host reads of a lazy dataset."""
        if batch_size % world:
            raise ValueError(f"--batch_size {batch_size} pairs do not split over {world} ranks")
        self.ds, self.bs, self.device, self.rank, self.world = dataset, batch_size, device, rank, world
        from .load_dataset import WorkerStreams
        self.streams = streams or WorkerStreams(0, 0)

    def __len__(self):
        return len(self.ds) // self.bs

    def __iter__(self):
        import torch
        bl = self.bs // self.world
        self.streams.begin_epoch()
        for b in range(len(self)):
            with self.streams.batch(b):
                items = [self.ds[b * self.bs + j] for j in range(self.bs)][self.rank * bl:(self.rank + 1) * bl]
            yield tuple(torch.stack([it[k] for it in items]).pin_memory().to(self.device, non_blocking=True) for k in range(4))

    def shuffle_keys(self):
        self.ds.shuffle_keys()


def _real_data(script, args, mode, part_len, pseudo_path, dev, rank, world, enc, head):
    """Feature-archive training source + evaluation closure for a Train/*.py run.  Dataset class per
    script as upstream (e.g. Train/temporal_transformer_shanghaitech.py:45-51, Train/spatio_transformer_MIL_CE.py:114-149).
    Under data parallelism every rank seeds ``np.random`` IDENTICALLY and walks the same permutation / window draws; rank
    ``r`` keeps pairs ``[r*bs/world, (r+1)*bs/world)`` of each global batch, so shards are disjoint and the global batch is
    the one a single process would have formed.  Window / crop draws come from the per-worker streams of the
    script's DataLoader (``load_dataset.WorkerStreams``: batch b from worker b % k, seeded ``seed + worker`` every epoch),
    pair order from the parent's generator, exactly as upstream."""
    import numpy as np
    from . import load_dataset as lds
    from .pipeline import evaluate_auc, evaluate_train_auc
    dataset = str(getattr(args, "dataset", "SHT"))
    np.random.seed(int(getattr(args, "seed", 0)))
    common = dict(part_num=args.part_num, part_len=part_len, h5_path=args.dataset_path, train_txt=args.training_txt,
                  n_patch=args.n_patch, sample=args.sample, pseudo_labels_path=pseudo_path if mode != "STN" else None)
    if dataset == "UCF":
        ds = lds.UCF_Train_Origin_Dataset(frames_per_clip=args.segment_len, **common)
    elif dataset == "UBnormal":
        ds = lds.UBnormal_Train_Origin_Dataset(**common)
    elif mode == "STN_MIL_CE":
        ds = lds.SH_Train_Origin_Dataset_MutualTraining(**common)
    else:
        ds = lds.SH_Train_Origin_Dataset(**common)
    # the loader workers' RNG streams: the count each script gives its DataLoader (temporal SHT / UBnormal hard-code 4,
    # temporal UCF 1, the spatio scripts and MIL_CE read --num_workers)
    k = {"temporal_transformer_shanghaitech": 4, "temporal_transformer_UBnormal": 4, "temporal_transformer_UCF": 1}.get(
        script, int(getattr(args, "num_workers", 0) or 0))
    streams = lds.WorkerStreams(k, int(getattr(args, "seed", 0)))
    # HBM-resident feed whenever the set fits (eager AND lazy datasets: upstream's per-item archive reads exist because host memory
    # cannot hold UCF-Crime; 288 GB of HBM can; since round 6 the ten-crop classes too, all ten crops resident); only sets beyond
    # 60 % of the free HBM - or LSTC_HOST_FEED=1 - stay host-staged
    import torch
    resident = lds.ResidentPairs.serves(ds) and os.environ.get("LSTC_HOST_FEED", "0") != "1" and \
        lds.ResidentPairs.bank_bytes(ds) < 0.6 * torch.cuda.mem_get_info(dev)[0]
    data = (lds.ResidentPairs if resident else _HostPairs)(ds, args.batch_size, dev, rank, world, streams)
    if resident:
        data.lazy_rows = True          # TrainStep gathers inside the CLS concat (lstc_cls_concat_gather_fwd)
    masks = getattr(args, "test_mask_path", "") if dataset == "UCF" else getattr(args, "test_mask_dir", "")
    test_arc = getattr(args, "test_dataset_path", "") or args.dataset_path
    kind = "LTN" if mode == "LTN" else "STN"

    train_flag = SELECTION[script]["train"]
    if script == "spatio_transformer_MIL_CE" and dataset in ("UCF", "UBnormal"):
        train_flag = None                                        # Train/spatio_transformer_MIL_CE.py:344-345: auc_train = 0
    train_arc = getattr(args, train_flag, "") if train_flag else ""

    def eval_fn():
        """(test AUC, train AUC) as the script's in-loop evaluation computes them (0 where it does not)."""
        auc_test = auc_train = 0.0
        if getattr(args, "testing_txt", ""):
            auc_test = evaluate_auc(enc.eval(), head.eval(), kind, dataset, test_arc, args.testing_txt, masks, part_len,
                                    args.n_patch, args.segment_len, rank=rank, world=world)
        if train_arc and dataset != "UCF":
            auc_train = evaluate_train_auc(enc.eval(), head.eval(), kind, dataset, train_arc, args.training_txt,
                                           getattr(args, "test_mask_dir", ""), part_len, args.n_patch, args.segment_len,
                                           rank=rank, world=world)
        return auc_test, auc_train
    return data, eval_fn


def evaluate(enc, head, mode, data, part_len, roc_auc, segment_len=16, train=False):
    """Frame-level AUC as the reference's in-loop evaluation (Train/temporal_transformer_shanghaitech.py:151-229):
    LTN scores every part of ``part_len`` clips (the last part may be shorter: shorter sequence, same model),
    STN scores every clip; scores are repeated to frame level (x segment_len)."""
    import numpy as np
    import torch
    enc.eval(); head.eval()
    scores, labels = [], []
    with torch.no_grad():
        for feats, labs in (data.train_videos_labelled() if train else data.test_videos()):
            sc = score_video(enc, head, mode, feats, part_len)
            scores.append(np.repeat(sc.cpu().numpy(), segment_len))
            labels.append(np.repeat(labs.reshape(-1).cpu().numpy(), segment_len))
    enc.train(); head.train()
    return roc_auc(np.concatenate(scores), np.concatenate(labels))


def score_video(enc, head, mode, feats, part_len):
    """Clip-level scores [n_clips] of one video.  STN: one score per clip.  LTN: one score per part of ``part_len``
    clips, repeated for the part's clips; a shorter tail part is a shorter sequence through the same model
    (Train/pseudo_labels_generator_temporal.py:120-141).  All full parts of the video go through ONE launch sequence
    instead of the reference's batch-1 loop."""
    import torch
    n, P, d = feats.shape
    if mode != "LTN":
        return head(enc.forward_cls(feats)).reshape(-1)
    full = n // part_len
    out = []
    if full:
        cls = enc.forward_cls(feats[: full * part_len].reshape(full, part_len * P, d))
        out.append(head(cls)[:, 1].repeat_interleave(part_len))
    if n > full * part_len:
        out.append(head(enc.forward_cls(feats[full * part_len:].reshape(1, -1, d)))[:, 1].repeat_interleave(n - full * part_len))
    return torch.cat(out)


def generate_pseudo_labels(script: str, argv=None, args=None):
    """Train/pseudo_labels_generator_{spatio,temporal}.py: score every training video, keep ``score > threshold``
    (else 0), save ``{key: [n_clips, 1]}`` with ``np.save`` (the pickled-dict format utils/load_dataset.py:20 reads)."""
    mode = "LTN" if script.endswith("temporal") else "STN"
    args = complete_args(script, args, argv)
    # --data_parallel --gpu 0,1,...: the training videos are sharded over one rank per listed GPU (pipeline._ScoreBoard); rank 0
    # writes the file (upstream wraps the models in nn.DataParallel here too, Train/pseudo_labels_generator_temporal.py:58-60)
    if _launch_data_parallel(script, args) is not None:
        return None
    if "LOCAL_RANK" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(getattr(args, "gpu", 0)))
    import numpy as np
    import torch
    from .data import SyntheticVideos
    from .models import Classifier, Encoder, Regressor
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: MI355X-only path")
    _apply_compute_dtype(args)
    rank, world, dev = _init_ranks(torch)
    part_len = getattr(args, "part_len", 1)
    enc = Encoder(n_layers=args.n_layers, n_head=args.n_head, d_k=args.d_k, d_v=args.d_v, d_model=args.d_model,
                  d_inner=args.n_hidden, MHA_layerNorm=args.MHA_layerNorm, FFN_layerNorm=args.FFN_layerNorm,
                  position_dropout=args.position_dropout, weight_init=args.encoder_weight_init,
                  position_encoding=args.position_encoding, CLS_learned=args.CLS_learned,
                  max_position_tokens=args.max_position_tokens, relative_pe=args.relative_position_encoding,
                  window_size=args.window_size, window_depth=part_len if mode == "LTN" else 3,
                  conv_patch=args.conv_patch)
    # the spatio generator pairs a 1-layer encoder with a Classifier (Train/pseudo_labels_generator_spatio.py:53-56)
    head = Classifier(args.d_model) if (mode == "LTN" or args.n_layers == 1) else Regressor(args.d_model)
    strip = lambda sd: {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    enc_path = getattr(args, "temporal_model_path", "") if mode == "LTN" else getattr(args, "spatio_model_path", "")
    head_path = getattr(args, "classifier_model_path", "") if mode == "LTN" else getattr(args, "regression_model_path", "")
    for m, path in ((enc, enc_path), (head, head_path)):
        if path and os.path.exists(path):
            m.load_state_dict(strip(torch.load(path, map_location="cpu")), False)
        else:
            print(f"[{script}] checkpoint {path!r} not found: scoring with the current initialisation", file=sys.stderr)
    enc, head = enc.to(dev).eval(), head.to(dev).eval()
    if not args.synthetic and args.dataset_path:
        from .pipeline import generate_pseudo_labels as run
        import time as _time
        torch.cuda.synchronize(); t0 = _time.perf_counter()
        out = run(enc, head, mode, args.dataset, args.dataset_path, args.training_txt, args.threshold, part_len=part_len,
                  n_patch=args.n_patch, d_model=args.d_model, segment_len=args.segment_len,
                  classifier_head=(mode == "STN" and args.n_layers == 1), out_path=args.pseudo_labels_path, rank=rank, world=world)
        torch.cuda.synchronize()
        LAST_RUN.clear()
        LAST_RUN.update(script=script, clips=int(sum(v.shape[0] for v in out.values())) if args.dataset != "UCF" else None,
                        videos=len(out), score_s=_time.perf_counter() - t0, world=world)
        if rank == 0:
            print(f"{'temporal' if mode == 'LTN' else 'spatio'} pseudo label generation finished.")
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return out
    # same seed -> same synthetic training videos (keys, lengths) as the Train/*.py loops on rank 0.  This branch is not sharded
    # (a few synthetic videos): under a --data_parallel launch every rank scores them all, rank 0 ALONE writes the file and prints
    data = SyntheticVideos(args.synthetic_pairs or 8, 1, 1, part_len, args.n_patch, args.d_model, dev,
                           seed=int(getattr(args, "seed", 0)))
    out = {}
    with torch.no_grad():
        for key, feats in data.train_videos():
            s = score_video(enc, head, mode, feats, part_len)
            s = torch.where(s > args.threshold, s, torch.zeros_like(s))
            out[key] = s.reshape(-1, 1).cpu().numpy()
    if rank == 0:
        np.save(args.pseudo_labels_path, out)
        print(f"{'temporal' if mode == 'LTN' else 'spatio'} pseudo label generation finished.")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return out


def evaluate_cli(script: str, argv=None, args=None):
    """Test/evaluation_shanghaitech_ubnormal.py (:69-96) and Test/evaluation_UCF.py (:47-88): frame-level AUC of a
    trained LTN.  SHT/UBnormal: consecutive parts of ``part_len`` clips, a short tail is RE-WINDOWED to the video's last
    ``part_len`` clips (:83-84; unlike the pseudo-label generators, which feed the short tail as a shorter sequence).
    UCF: every video is averaged into 32 bins (linspace :54), parts of part_len = 2 bins (:42), features L2-normalised
    (:77).  All full parts of a video are scored in one launch sequence instead of one part per launch."""
    args = complete_args(script, args, argv)
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(getattr(args, "gpu", 0)))
    import numpy as np
    import torch
    from .data import SyntheticVideos
    from .metrics import roc_auc
    from .models import Classifier, Encoder
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: MI355X-only path")
    _apply_compute_dtype(args)
    dev = torch.device("cuda", 0)
    ucf = script.endswith("UCF")
    part_len = 2 if ucf else args.part_len
    rel = getattr(args, "temporal_relative_position_encoding", False) or getattr(args, "relative_position_encoding", False)
    enc = Encoder(n_layers=args.temporal_n_layers, n_head=args.temporal_n_head, d_k=args.temporal_d_k, d_v=args.temporal_d_v,
                  d_model=args.d_model, d_inner=args.temporal_n_hidden, MHA_layerNorm=args.temporal_MHA_layerNorm,
                  FFN_layerNorm=args.temporal_FFN_layerNorm, relative_pe=rel, window_size=args.window_size,
                  window_depth=args.part_len, weight_init=False)
    head = Classifier(args.d_model)
    strip = lambda sd: {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    for m, path in ((enc, args.temporal_model_path), (head, args.classifier_model_path)):
        if path and os.path.exists(path):
            m.load_state_dict(strip(torch.load(path, map_location="cpu")), False)
        else:
            print(f"[{script}] checkpoint {path!r} not found: evaluating the current initialisation", file=sys.stderr)
    enc, head = enc.to(dev).eval(), head.to(dev).eval()
    seg = getattr(args, "segment_len", 16)
    if not args.synthetic and args.dataset_path:
        from .pipeline import evaluate_auc
        masks = args.test_mask_path if ucf else args.test_mask_dir
        import time as _time
        torch.cuda.synchronize(); t0 = _time.perf_counter()
        auc, fs, _ = evaluate_auc(enc, head, "LTN", "UCF" if ucf else args.dataset, args.dataset_path, args.testing_txt, masks,
                                  part_len, args.n_patch, seg, return_frames=True)
        torch.cuda.synchronize()
        LAST_RUN.clear()
        LAST_RUN.update(script=script, clips=int(fs.shape[0]) // seg, score_s=_time.perf_counter() - t0, world=1)
        print("auc = ", auc)
        return auc
    data = SyntheticVideos(2, 1, 1, part_len, args.n_patch, args.d_model, dev, seed=0)
    scores, labels = [], []
    with torch.no_grad():
        for feats, labs in data.test_videos(getattr(args, "synthetic_pairs", 0) or 8):
            n, P, d = feats.shape
            lab = labs.reshape(-1)
            if ucf:
                r = np.linspace(0, n, 33, dtype=np.int32)
                bins = torch.stack([feats[r[i]] if r[i] == r[i + 1] else feats[r[i]:r[i + 1]].mean(dim=0) for i in range(32)])
                bins = torch.nn.functional.normalize(bins, p=2, dim=-1)
                nparts = (32 + part_len - 1) // part_len
                begs = [min(i * part_len, 32 - part_len) for i in range(nparts)]
                ends = [min((i + 1) * part_len, 32) for i in range(nparts)]
                x = torch.stack([bins[b:b + part_len].reshape(part_len * P, d) for b in begs])
                sc = head(enc.forward_cls(x))[:, 1].cpu().numpy()
                for s_, b, e in zip(sc, begs, ends):
                    scores.append(np.full((r[e] - r[b]) * seg, s_))
                    labels.append(np.repeat(lab[r[b]:r[e]].cpu().numpy(), seg))
            else:
                nparts = (n + part_len - 1) // part_len
                begs = [i * part_len for i in range(nparts)]
                ends = [min((i + 1) * part_len, n) for i in range(nparts)]
                win = [max(e - part_len, 0) if e - b < part_len else b for b, e in zip(begs, ends)]
                if n >= part_len:
                    x = torch.stack([feats[w:w + part_len].reshape(part_len * P, d) for w in win])
                    sc = head(enc.forward_cls(x))[:, 1].cpu().numpy()
                else:           # video shorter than one part: a single shorter sequence
                    sc = head(enc.forward_cls(feats.reshape(1, n * P, d)))[:, 1].cpu().numpy()
                for s_, b, e in zip(sc, begs, ends):
                    scores.append(np.full((e - b) * seg, s_))
                    labels.append(np.repeat(lab[b:e].cpu().numpy(), seg))
    auc = roc_auc(np.concatenate(scores), np.concatenate(labels))
    print("auc = ", auc)
    return auc


def main(script: str):
    if script.startswith("evaluation"):
        evaluate_cli(script)
    elif script.startswith("pseudo_labels_generator"):
        generate_pseudo_labels(script)
    elif script in SCRIPTS:
        train(script)
    else:
        raise SystemExit(f"{script}: evaluation CLI is the 'next' row of SURVEY.md 8f-2 "
                         "(flags parse via lstc_vad_amd.cli.build_parser; the loop is lstc_vad_amd.cli.evaluate)")
