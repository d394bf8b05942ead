#!/usr/bin/env python3
"""Pre-registered expectation for the N = 1 / 2 / 4 / 8 strong-scaling run of bench.py.

Inputs: the EMULATED per-rank step (one GPU, forced one-rank RCCL bucket path, no communication) at 32 / 16 / 8 / 4 pairs per rank
(tools/r06_scaling_expectation.sh) - its step time and its backward time.  Model of the one collective of the step, the bucketed
sum-all-reduce of 407 MB of fp32 gradients in 7 buckets (<= 67 MB) issued in backward order:

  * xGMI is point-to-point: N GPUs of one node are fully connected, a pair shares ONE link of 76.8 GB/s per direction (153.6 GB/s
    both ways, /opt/skills/guides: "7 links x ~153 GB/s per GPU").  A direct reduce-scatter + all-gather moves S / N bytes over each of
    the N - 1 links of a GPU in each phase:   t(S, N) = alpha + 2 (S / N) / (eff * 76.8 GB/s),   alpha = 30 us per bucket,
    eff = 0.7 (0.5 - 0.85 bracket the published RCCL bus-bandwidth range for 64-MB messages).  N = 2 is the slow case: ONE link.
  * buckets 1 .. 6 are reduced beside the backward of the layers below them; what cannot hide is the LAST bucket (layer 0's attention
    weights, 67 MB, ready when the backward ends) plus whatever of the earlier traffic exceeds the backward time that follows it.
    exposed = max(t(last), t(all) - 0.85 * backward)   (the first bucket becomes ready ~15 % into the backward)
  * the bag exchange inside the loss (one sum-all-reduce of 2 bs floats, on the critical path between the head and the backward): + alpha.
  * expected step = emulated rank step + exposed + alpha;  value = global snippets / step;  efficiency = value(N) / (N * value(1)).
"""
import json
import sys

BUCKETS_MB = [1.1, 67.1, 67.1, 67.1, 67.1, 67.1, 67.1]          # head, then per layer its FFN half and its attention half (LTN widths)
LINK, ALPHA = 76.8e9, 30e-6


def t_allreduce(mb, n, eff):
    return 0.0 if n == 1 else ALPHA + 2.0 * (mb * 1e6 / n) / (eff * LINK)


def main(prefix):
    rows = []
    for dt in ("fp32", "bf16"):
        base = None
        for n, bs in ((1, 32), (2, 16), (4, 8), (8, 4)):
            try:
                o = json.load(open(f"{prefix}_bs{bs}_{dt}.json"))
            except Exception as e:                                        # a missing run: say so instead of inventing a number
                rows.append(f"{dt} N={n}: no emulated rank step ({e})")
                continue
            step, bw = o["ms_per_step"] * 1e-3, o["config"].get("backward_ms_per_step", 0.0) * 1e-3
            out = []
            for eff in (0.5, 0.7, 0.85):
                t_all = sum(t_allreduce(mb, n, eff) for mb in BUCKETS_MB)
                exposed = 0.0 if n == 1 else max(t_allreduce(BUCKETS_MB[-1], n, eff), t_all - 0.85 * bw)
                out.append((step + exposed + (ALPHA if n > 1 else 0.0), exposed, t_all))
            snippets = 64 * 32 * 3
            val = [snippets / s for s, _, _ in out]
            if n == 1:
                base = val[1]
            eff_s = [v / (n * base) for v in val] if base else [float("nan")] * 3
            rows.append(f"{dt} N={n}: emulated rank step {step * 1e3:7.2f} ms (backward {bw * 1e3:6.2f}) | all-reduce total "
                        f"{out[1][2] * 1e3:5.2f} ms, exposed {out[1][1] * 1e3:5.2f} ms [{out[2][1] * 1e3:.2f} .. {out[0][1] * 1e3:.2f}] | expected step "
                        f"{out[1][0] * 1e3:7.2f} ms [{out[2][0] * 1e3:.2f} .. {out[0][0] * 1e3:.2f}] | value {val[1]:9.0f} snippets/s | scaling efficiency "
                        f"{eff_s[1]:.3f} [{eff_s[0]:.3f} .. {eff_s[2]:.3f}]")
    print("\n".join(rows))


if __name__ == "__main__":
    main(sys.argv[1])
