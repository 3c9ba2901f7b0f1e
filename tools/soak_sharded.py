"""Soak of the process-group bodies of tests/test_zz_dist_gpu.py: each body N times, every run in a fresh child process, the child's FULL
stderr kept in a file; a summary line per run (exit code + the first line that names a cause).

    python tools/soak_sharded.py OUTDIR [N] [body ...]
"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAUSE = re.compile(r"terminate called|what\(\)|HIP error|hipError|Memory access fault|Fatal Python error|Segmentation|RuntimeError|NCCL|Assertion")


def first_cause(text):
    for line in text.splitlines():
        if CAUSE.search(line):
            return line.strip()[:400]
    return ""


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    bodies = sys.argv[3:] or ["_pipelined_sharded_step_single_rank_process_group", "_sharded_step_single_rank_process_group"]
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TORCH_SHOW_CPP_STACKTRACES="1", TORCH_NCCL_TRACE_BUFFER_SIZE="2000", TORCH_NCCL_DUMP_ON_TIMEOUT="1",
               PYTHONFAULTHANDLER="1")
    bad = 0
    with open(os.path.join(out, "summary.txt"), "a") as summ:
        for body in bodies:
            for i in range(n):
                code = ("import sys; sys.path[:0] = [%r, %r]; import tests.test_zz_dist_gpu as t; t.%s(); print('ISOLATED-BODY-OK', flush=True)"
                        % (ROOT, os.path.join(ROOT, "repsurf_amd", "classification"), body))
                t0 = time.time()
                r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
                with open(os.path.join(out, f"{body}.{i}.err"), "w") as f:
                    f.write(r.stderr)
                with open(os.path.join(out, f"{body}.{i}.out"), "w") as f:
                    f.write(r.stdout)
                ok = r.returncode == 0 and "ISOLATED-BODY-OK" in r.stdout
                bad += not ok
                line = f"{body} run {i}: rc {r.returncode} body_ok {'ISOLATED-BODY-OK' in r.stdout} {time.time() - t0:.1f}s  {first_cause(r.stderr)}"
                print(line, flush=True)
                summ.write(line + "\n")
    print(f"soak: {bad} bad runs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
