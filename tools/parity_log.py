#!/usr/bin/env python3
"""Re-runs the GPU parity suite with one CSV line per comparison written beside it (name, size, max|ref|,
max-norm error, worst allclose ratio) -- the evidence kept under profiles/rN_parity_log.csv.  The tests'
assertions stay ON: tests/helpers.assert_close logs and THEN asserts, there is no log-only mode.

    python tools/parity_log.py gpurun_out/parity_log.csv [extra pytest arguments]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "parity_log.csv")
    with open(out, "w") as f:
        f.write("comparison,elements,max_abs_ref,max_norm_error,worst_allclose_ratio\n")
    env = dict(os.environ, GGNN_PARITY_LOG=out)
    rc = subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "gpu"] + sys.argv[2:],
                         env=env, cwd=ROOT)
    print(f"parity log: {out} (pytest exit code {rc})")
    return rc


if __name__ == "__main__":
    sys.exit(main())
