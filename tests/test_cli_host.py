"""CPU checks of the Train/*.py command-line surface and the window sampler."""
import json
import os

import numpy as np
import pytest

from lstc_vad_amd import cli
from lstc_vad_amd.data import sample_windows
from lstc_vad_amd.metrics import roc_auc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flag_surface_counts_match_survey():
    # SURVEY.md Appendix A, extracted from the reference's parser_arg() functions
    want = {"spatio_transformer_shanghaitech": 55, "spatio_transformer_UCF": 57, "spatio_transformer_UBnormal": 49,
            "temporal_transformer_shanghaitech": 57, "temporal_transformer_UCF": 51, "temporal_transformer_UBnormal": 56,
            "spatio_transformer_MIL_CE": 84, "pseudo_labels_generator_spatio": 29, "pseudo_labels_generator_temporal": 37,
            "evaluation_shanghaitech_ubnormal": 35, "evaluation_UCF": 32}
    flags = json.load(open(os.path.join(ROOT, "lstc_vad_amd", "cli_flags.json")))
    assert {k: len(v) for k, v in flags.items()} == want


def test_reference_command_lines_parse():
    p = cli.build_parser("temporal_transformer_shanghaitech")      # README.md:31
    a = p.parse_args("--part_len 3 --MHA_layerNorm --FFN_layerNorm --relative_position_encoding "
                     "--pseudo_labels_path STN_pseudo_labels.npy --dataset_path SHT_I3D_16PATCH.h5 --gpu 0".split())
    assert a.part_len == 3 and a.MHA_layerNorm and a.n_hidden == 4096 and a.batch_size == 40 and a.lambda_CE == 0.8
    assert a.MHA_attn_dropout == 0.2 and a.lr_encoder == 1e-4 and a.weight_decay == 1e-3 and a.window_size == 4
    p = cli.build_parser("spatio_transformer_shanghaitech")        # README.md:23 minus the flag upstream rejects too
    a = p.parse_args("--encoder_weight_init --regressor_weight_init --FFN_layerNorm --FFN_dropout 0.3 --gpu 0".split())
    assert a.n_hidden == 3027 and a.part_len == 7 and a.FFN_dropout == 0.3 and a.lr_regressor == 1e-2
    with pytest.raises(SystemExit):                                 # README's --MHA_dropout is rejected upstream as well
        p.parse_args(["--MHA_dropout", "0.3"])
    a = cli.build_parser("spatio_transformer_MIL_CE").parse_args([])
    assert a.spatio_part_len == 7 and a.lambda_normal == 0.2 and a.lambda_abnormal == 2.0 and a.lambda_BCE == 1.0
    a = cli.build_parser("temporal_transformer_UCF").parse_args([])
    assert a.n_patch == 9 and not hasattr(a, "data_parallel")
    a = cli.build_parser("pseudo_labels_generator_temporal").parse_args([])
    assert a.threshold == 0.9 and a.encoder_weight_init is False     # flag the reference forgets to define


def test_every_train_script_exists():
    for s in list(cli.SCRIPTS) + ["pseudo_labels_generator_spatio", "pseudo_labels_generator_temporal"]:
        assert os.path.exists(os.path.join(ROOT, "Train", s + ".py")), s
    for s in ("evaluation_shanghaitech_ubnormal", "evaluation_UCF"):
        assert os.path.exists(os.path.join(ROOT, "Test", s + ".py")), s
    a = cli.build_parser("evaluation_UCF").parse_args("--n_patch 9 --part_num 32 --part_len 2 --temporal_MHA_layerNorm "
                                                      "--temporal_FFN_layerNorm --relative_position_encoding --gpu 0".split())
    assert a.n_patch == 9 and a.temporal_n_hidden == 4096 and a.window_size == 4       # README.md:59


REF_SURFACE = {
    # module path -> {function: positional parameter names}, as the reference's modules define them (file:line in each shim)
    "Train.spatio_transformer_shanghaitech": {"get_MIL_loss": ["args", "y_pred"], "train": ["args"], "parser_arg": []},
    "Train.spatio_transformer_UCF": {"get_MIL_loss": ["args", "y_pred"], "train": ["args"], "parser_arg": []},
    "Train.spatio_transformer_UBnormal": {"get_MIL_loss": ["args", "y_pred"], "train": ["args"], "parser_arg": []},
    "Train.temporal_transformer_shanghaitech": {"get_CE_loss": ["args", "outputs", "labs"], "get_MIL_loss": ["args", "y_pred"],
                                                "train": ["args"], "parser_arg": []},
    "Train.temporal_transformer_UCF": {"get_CE_loss": ["args", "outputs", "labs"], "get_MIL_loss": ["args", "y_pred"],
                                       "train": ["args"], "parser_arg": []},
    "Train.temporal_transformer_UBnormal": {"get_CE_loss": ["args", "outputs", "labs"], "get_MIL_loss": ["args", "y_pred"],
                                            "train": ["args"], "parser_arg": []},
    "Train.spatio_transformer_MIL_CE": {"get_BCE_loss": ["args", "outputs", "labs"], "get_CE_loss": ["args", "outputs", "labs"],
                                        "get_MIL_loss": ["args", "y_pred", "part_len"], "train": ["args"], "parser_arg": []},
    "Train.pseudo_labels_generator_spatio": {"generator": ["args"], "parser_arg": []},
    "Train.pseudo_labels_generator_temporal": {"generator": ["args"], "parser_arg": []},
    "Test.evaluation_UCF": {"evaluation": ["args"], "parser_arg": []},
    "Test.evaluation_shanghaitech_ubnormal": {"evaluation": ["args"], "parser_arg": []},
}


def test_train_and_test_modules_export_the_reference_function_surface(monkeypatch):
    """callers of the reference import ``get_MIL_loss(args, y_pred[, part_len])``, ``get_CE_loss``,
    ``get_BCE_loss``, ``train``, ``parser_arg`` (``generator`` / ``evaluation`` for the other scripts) from the Train / Test
    MODULES.  Every shim exports them under the reference's import path with the reference's parameter names; where the reference
    tree is present the expected table itself is checked against the reference's source (ast: no reference code runs)."""
    import ast
    import importlib
    import inspect
    for mod_name, fns in REF_SURFACE.items():
        mod = importlib.import_module(mod_name)
        assert os.path.realpath(mod.__file__).startswith(ROOT + os.sep), mod.__file__        # the repo's shim, not the reference
        for fn, params in fns.items():
            f = getattr(mod, fn)
            assert list(inspect.signature(f).parameters) == params, (mod_name, fn)
        ref_file = os.path.join("/root/reference", *mod_name.split(".")) + ".py"
        if os.path.exists(ref_file):
            tree = ast.parse(open(ref_file).read())
            ref = {n.name: [a.arg for a in n.args.args] for n in tree.body if isinstance(n, ast.FunctionDef)}
            assert ref == fns, (mod_name, ref)
    # parser_arg() parses sys.argv like the reference's; a caller-built Namespace is completed with the script's defaults
    import Train.temporal_transformer_shanghaitech as t
    monkeypatch.setattr("sys.argv", ["x", "--part_len", "3", "--MHA_layerNorm", "--batch_size", "8"])
    a = t.parser_arg()
    assert a.part_len == 3 and a.MHA_layerNorm and a.batch_size == 8 and a.lambda_CE == 0.8 and a.steps == 0
    from argparse import Namespace
    b = cli.complete_args("temporal_transformer_shanghaitech", Namespace(part_len=5, batch_size=2, my_extra=1))
    assert b.part_len == 5 and b.batch_size == 2 and b.my_extra == 1 and b.n_hidden == 4096 and b.compute_dtype in ("fp32", "f32x3", "bf16")


def test_window_sampler_matches_reference_example():
    # SURVEY.md Appendix B (verified by executing the reference): n=40, pn=4, L=3, np seed 0
    np.random.seed(0)
    assert sample_windows(40, 4, 3, "uniform").tolist() == [4, 5, 6, 13, 14, 15, 22, 23, 24, 31, 32, 33]
    idx = sample_windows(10, 4, 3, "uniform")          # (n-L)//(pn+1) = 1 -> move = randint(1) = 0
    assert idx.tolist() == [0, 1, 2, 1, 2, 3, 3, 4, 5, 5, 6, 7]
    idx = sample_windows(200, 32, 2, "random", np.random.RandomState(1))
    assert idx.shape == (64,) and idx.max() < 200 and (np.diff(idx.reshape(32, 2), axis=1) == 1).all()


def test_auc_matches_reference_eval():
    z = np.load(os.path.join(ROOT, "tests", "golden", "misc.npz"), allow_pickle=False)
    assert abs(roc_auc(z["auc_scores"], z["auc_labels"]) - float(z["auc_value"])) < 1e-12
    assert abs(roc_auc(z["auc2_scores"], z["auc2_labels"]) - float(z["auc2_value"])) < 1e-12


def test_models_import_shim():
    from models.Encoder import Encoder
    from models.FFN import PositionwiseFeedForward
    from models.Classifier import Classifier
    from lstc_vad_amd.models import Encoder as E2
    assert Encoder is E2 and PositionwiseFeedForward.__name__ == "PositionwiseFeedForward" and Classifier


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus N`` without a launcher: N fresh rank processes with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*,
    rank 0's JSON line relayed, a failing rank fails the run (bench.launch_ranks; no GPU needed for the mechanism)."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lstc_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ok = tmp_path / "rank_ok.py"
    ok.write_text("import os, json, sys\n"
                  "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                  "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                  "print('noise from rank', r)\n"
                  "if r == 0: print(json.dumps({'n_gpus': w, 'argv': sys.argv[1:]}))\n")
    bad = tmp_path / "rank_bad.py"
    bad.write_text("import os, sys, time\n"
                   "if os.environ['RANK'] == '1': sys.exit(3)\n"
                   "time.sleep(30)\n")
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "2"], script=str(ok))
    assert rc == 0
    out = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert out == {"n_gpus": 3, "argv": ["--gpus", "3", "--steps", "2"]}
    import time
    t0 = time.time()
    assert bench.launch_ranks(2, [], script=str(bad)) == 1         # rank 1 fails -> rank 0 is stopped, run fails
    assert time.time() - t0 < 20


def test_bench_launcher_watchdog_stops_a_stuck_rank(tmp_path, capfd):
    """A rank that never finishes (e.g. stuck in RCCL initialisation) must not hang ``bench.py --gpus N``: after
    --rank_timeout_s the launcher stops the PIDs it started, prints every rank's last stderr lines and returns 1 - even a
    rank that ignores SIGTERM (killed after the grace period)."""
    import importlib.util
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lstc_bench_wd", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stuck = tmp_path / "rank_stuck.py"
    stuck.write_text("import os, sys, time, signal\n"
                     "r = int(os.environ['RANK'])\n"
                     "print('rank', r, 'entering init', file=sys.stderr, flush=True)\n"
                     "if r == 1:\n"
                     "    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                     "    print('rank 1 waiting for a peer that never comes', file=sys.stderr, flush=True)\n"
                     "    time.sleep(600)\n"
                     "time.sleep(600)\n")
    t0 = time.time()
    rc = bench.launch_ranks(2, [], script=str(stuck), rank_timeout_s=3.0)
    took = time.time() - t0
    assert rc == 1 and took < 30, (rc, took)
    err = capfd.readouterr().err
    assert "watchdog" in err and "rank 1 waiting for a peer that never comes" in err and "rank 0 entering init" in err


def test_host_shape_rules_of_the_packed_and_padded_paths():
    """Pure host logic of lstc_vad_amd.functional that decides kernel paths: the width an unaligned FFN hidden runs at, and
    the shapes for which producers may emit packed bf16 operands (both must agree with what include/lstc_hip.h documents)."""
    from lstc_vad_amd import functional as Fn
    assert [Fn._padded_hidden(f) for f in (3027, 4096, 47, 40, 250, 300, 1, 255, 256, 257)] == \
        [3072, 4096, 48, 40, 256, 300, 4, 256, 256, 260]
    Fn.set_compute_dtype("bf16")
    try:
        assert Fn._attn_dtype() == Fn._lib.BF16
        assert Fn._fused_pack_shape(100352, 2048) and not Fn._fused_pack_shape(100352 + 128, 2048) and not Fn._fused_pack_shape(100352, 4096)
        assert Fn.attn_bwd_packs(2048, 49, 8, 256, 256) and Fn.attn_bwd_packs(2048, 81, 8, 128, 128)
        assert not Fn.attn_bwd_packs(2048, 113, 8, 256, 256) and not Fn.attn_bwd_packs(2048, 49, 8, 48, 48)
        assert Fn.attn_fwd_pack(2048, 49, 8, 256) and not Fn.attn_fwd_pack(2047, 49, 8, 256)
    finally:
        Fn.set_compute_dtype("fp32")
    assert Fn._attn_dtype() == Fn._lib.F32 and not Fn._fused_pack_shape(100352, 2048)


def test_namespace_round_trips_through_the_rank_command_line():
    """``train(args)`` on a caller-built Namespace starts its ranks with the command line that parses back to that Namespace
    (cli.namespace_to_argv): every flag of every script survives the round trip, floats exactly."""
    from lstc_vad_amd import cli
    for script in list(cli.SCRIPTS) + ["pseudo_labels_generator_temporal", "pseudo_labels_generator_spatio"]:
        has = lambda f: any(x[0] == f for x in cli._FLAGS[script])
        a = cli.complete_args(script, argv=["--gpu", "0,1"] + (["--data_parallel"] if has("--data_parallel") else []) +
                              (["--lr_encoder", "0.000123456789"] if has("--lr_encoder") else []))
        back = cli.complete_args(script, argv=cli.namespace_to_argv(script, a))
        assert vars(back) == vars(a), script


def test_data_parallel_flag_starts_one_rank_per_listed_gpu(monkeypatch, tmp_path):
    """``python Train/<script>.py --data_parallel --gpu 0,1,2,3`` IS the multi-GPU run (Train/temporal_transformer_shanghaitech.py:
    76-78, :328): cli.train hands over to lstc_vad_amd.launch.launch_ranks - one rank per listed GPU, the Train script itself as the
    rank program, the devices exported to the ranks - BEFORE importing torch's GPU side, and returns what rank 0 handed back.  One
    listed GPU, or a process that already is a rank, runs the training itself.  The launcher is exercised with stub rank programs
    (no GPU needed): stdout of rank 0 passes through, the LSTC_RESULT line comes back, a failing rank fails the run."""
    from lstc_vad_amd import cli, launch
    calls = []

    def fake(n, argv, script=None, rank_timeout_s=0.0, relay="json", devices=None, tag="", extra_env=None, init_timeout_s=0.0):
        calls.append(dict(n=n, argv=list(argv), script=script, relay=relay, devices=devices, init_timeout_s=init_timeout_s))
        fake.last_result = "0.75"
        return 0
    fake.last_result = None
    monkeypatch.setattr(launch, "launch_ranks", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    got = cli.train("temporal_transformer_shanghaitech", argv=["--data_parallel", "--gpu", "0,1,2,3", "--batch_size", "8", "--epochs", "3"])
    assert got == 0.75 and len(calls) == 1
    c = calls[0]
    assert c["n"] == 4 and c["devices"] == ["0", "1", "2", "3"] and c["relay"] == "all"
    assert c["init_timeout_s"] == 900.0         # the CLI path bounds the rendezvous (not the run) by default
    assert c["script"].endswith(os.path.join("Train", "temporal_transformer_shanghaitech.py")) and os.path.exists(c["script"])
    back = cli.complete_args("temporal_transformer_shanghaitech", argv=c["argv"])
    assert back.data_parallel and back.gpu == "0,1,2,3" and back.batch_size == 8 and back.epochs == 3
    with pytest.raises(SystemExit):                          # the pair count must split over the ranks
        cli.train("temporal_transformer_shanghaitech", argv=["--data_parallel", "--gpu", "0,1,2", "--batch_size", "8"])
    # already a rank (torchrun / a launched child), or one device: no launch - the call goes on to the training itself,
    # which needs a GPU: here it stops at "no HIP device visible"
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert cli._launch_data_parallel("temporal_transformer_shanghaitech", back) is None
    monkeypatch.delenv("WORLD_SIZE")
    one = cli.complete_args("temporal_transformer_shanghaitech", argv=["--data_parallel", "--gpu", "2"])
    assert cli._launch_data_parallel("temporal_transformer_shanghaitech", one) is None
    assert len(calls) == 1
    # the generators take the same flags upstream
    gen = cli.complete_args("pseudo_labels_generator_temporal", argv=["--data_parallel", "--gpu", "0,1"])
    assert cli._launch_data_parallel("pseudo_labels_generator_temporal", gen) == ("done", 0.75) and calls[-1]["n"] == 2
    monkeypatch.undo()
    # the real launcher with stub rank programs
    ok = tmp_path / "rank_ok.py"
    ok.write_text("import os, sys\n"
                  "r = int(os.environ['RANK'])\n"
                  "assert os.environ['HIP_VISIBLE_DEVICES'] == '2,5' and os.environ['LSTC_LAUNCHED'] == '1' and os.environ['LOCAL_RANK'] == str(r)\n"
                  "print('line from rank', r, flush=True)\n"
                  "if r == 0: print('LSTC_RESULT 0.5', flush=True)\n")
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = launch.launch_ranks(2, ["--x", "1"], script=str(ok), relay="all", devices=["2", "5"], tag="t", rank_timeout_s=60)
    assert rc == 0 and launch.launch_ranks.last_result == "0.5" and buf.getvalue().strip() == "line from rank 0"
    bad = tmp_path / "rank_bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(30)\n")
    assert launch.launch_ranks(2, [], script=str(bad), relay="all", tag="t", rank_timeout_s=60) == 1
    # several complete lines written in ONE pipe chunk are all relayed when they arrive (a buffered readline() per
    # selector wake-up kept the later ones back until the next write); the children run unbuffered
    burst = tmp_path / "rank_burst.py"
    burst.write_text("import os, sys, time\n"
                     "assert os.environ.get('PYTHONUNBUFFERED') == '1'\n"
                     "if os.environ['RANK'] == '0':\n"
                     "    os.write(1, b'a1\\na2\\na3\\n')\n"
                     "    time.sleep(3.0)\n"
                     "    os.write(1, b'tail without newline')\n")
    import threading
    import time as _t
    seen = []

    class Tap(io.StringIO):
        def write(self, x):
            seen.append((_t.monotonic(), x))
            return super().write(x)
    tap = Tap()
    t0 = _t.monotonic()
    with contextlib.redirect_stdout(tap):
        assert launch.launch_ranks(2, [], script=str(burst), relay="all", tag="t", rank_timeout_s=60) == 0
    lines = [(t, x) for t, x in seen if x.strip()]
    assert [x.strip() for _, x in lines] == ["a1", "a2", "a3", "tail without newline"]
    assert lines[2][0] - lines[0][0] < 1.0 and lines[3][0] - lines[2][0] > 1.5        # the burst came through at once, the tail at EOF
    # the init watchdog: a rank that never joins the process group (never calls mark_rank_ready) stops the job after init_timeout_s,
    # a job whose ranks all did runs on past it
    hang = tmp_path / "rank_hang.py"
    hang.write_text("import os, sys, time\nsys.path.insert(0, %r)\nfrom lstc_vad_amd.launch import mark_rank_ready\n"
                    "if os.environ['RANK'] == '0': mark_rank_ready()\ntime.sleep(60)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    t0 = _t.monotonic()
    assert launch.launch_ranks(2, [], script=str(hang), relay="all", tag="t", rank_timeout_s=0, init_timeout_s=3.0) == 1
    assert _t.monotonic() - t0 < 30
    fine = tmp_path / "rank_fine.py"
    fine.write_text("import os, sys, time\nsys.path.insert(0, %r)\nfrom lstc_vad_amd.launch import mark_rank_ready\n"
                    "mark_rank_ready()\ntime.sleep(5)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert launch.launch_ranks(2, [], script=str(fine), relay="all", tag="t", rank_timeout_s=0, init_timeout_s=3.0) == 0
