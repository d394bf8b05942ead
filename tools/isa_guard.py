"""Build-time guard for csrc/gemm_bf16p.hip: the pipelined epilogue issues operand loads through inline asm whose destination
registers are "ready" for the compiler from the instruction on although the data arrives later.  The register allocator must
therefore never spill (or copy) such a register between the load and the hand-placed s_waitcnt - a spill would save stale
contents and free the register for another value that the landing load then overwrites.  Compiles the file to assembly and
fails if any asm global_load destination is stored to scratch before the next s_waitcnt vmcnt.   python tools/isa_guard.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                            "--cuda-device-only", "-o", out, os.path.join(ROOT, "lstc_vad_amd", "csrc", "gemm_bf16p.hip")],
                           capture_output=True, text=True, cwd=tmp)
        if r.returncode:
            print(r.stderr[-2000:])
            return 2
        s = open(out).read()
    bad = 0
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_117gemm_bf16p_kernel\w+):", s, re.M):
        body = s[m.start():s.index(".amdhsa_kernel", m.start())].split("\n")
        hits = []
        for n, line in enumerate(body):
            mm = re.search(r"global_load_dwordx[24] (v\[\d+:\d+\])", line)
            if not (mm and "ASMSTART" in body[n - 1]):
                continue
            for t in range(n + 1, min(n + 200, len(body))):
                if re.search(r"s_waitcnt vmcnt", body[t]) and "ASMSTART" not in body[t - 1]:
                    break
                if ("scratch_store" in body[t] or "v_mov" in body[t] or "v_accvgpr_write" in body[t]) and mm.group(1) in body[t]:
                    hits.append((n, t, mm.group(1)))
                    break
        n_spill = sum("scratch_store" in l for l in body)
        print(f"{m.group(1)[22:60]:40s} spill stores {n_spill:3d}   in-flight registers touched: {hits[:4]}")
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
