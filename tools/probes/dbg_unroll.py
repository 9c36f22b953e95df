import os, sys, json, subprocess
for u in (2, 3, 4, 5, 10):
    env = dict(os.environ, GGNN_BENCH_UNROLL=str(u), GGNN_BENCH_SETTLE="1")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "2", "--no-cpu-baseline"] if False else [sys.executable, "bench.py", "--steps", "20", "--warmup", "2"], env=env, capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    try:
        d = json.loads(line); print(u, d["value"], d["ms_per_step"], flush=True)
    except Exception:
        print(u, "FAILED", r.stderr.strip().splitlines()[-6:-4], flush=True)
