"""Same-run A/B of library builds on the 75-frame call (and the 15-frame one): ab75.py <lib>... (paths of libsacopt.so builds).
Each build runs in a child process (one library per process)."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
frames = os.environ.get("AB_FRAMES", "75 15").split()
for rep in range(2):
    for lib in libs:
        row = []
        for nf in frames:
            env = dict(os.environ, SACOPT_LIB=os.path.join(ROOT, lib))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--frames", nf, "--steps", "2", "--warmup", "1"],
                                 capture_output=True, text=True, env=env, timeout=300)
            try:
                row.append(f"{nf} frames {json.loads(out.stdout.strip().splitlines()[-1])['ms_per_step']:8.1f} ms")
            except Exception:
                row.append(f"{nf} frames FAILED {out.stderr[-200:]}")
        print(f"{lib:40s} " + "   ".join(row), flush=True)
