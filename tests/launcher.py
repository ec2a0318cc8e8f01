"""Child-process launcher for tests that start several rank processes on the GPU box (test infrastructure).

A process that has initialised the GPU must not start other programs on this pool; so conftest.py starts THIS helper
before any test touches the GPU, and the tests ask it - over a pipe, one JSON object per line - to run their commands:
    {"cmd": [...], "env": {...}, "timeout": seconds}  ->  {"rc": int, "stdout": "...", "stderr": "..."}
The helper itself never imports torch or any HIP library.
"""
import json
import os
import subprocess
import sys


def serve():
    for line in sys.stdin:
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        for k in req.get("unset") or []:
            env.pop(k, None)
        try:
            p = subprocess.run(req["cmd"], env=env, capture_output=True, text=True, timeout=req.get("timeout", 600), cwd=req.get("cwd"))
            out = {"rc": p.returncode, "stdout": p.stdout[-20000:], "stderr": p.stderr[-20000:]}
        except subprocess.TimeoutExpired as e:
            out = {"rc": -999, "stdout": (e.stdout or b"").decode(errors="replace")[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or ""),
                   "stderr": "timeout after %s s" % req.get("timeout", 600)}
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


class Launcher:
    """Client side (lives in the pytest process)."""

    def __init__(self):
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def run(self, cmd, env=None, unset=None, timeout=600, cwd=None):
        self.proc.stdin.write(json.dumps({"cmd": list(cmd), "env": env or {}, "unset": unset or [], "timeout": timeout, "cwd": cwd}) + "\n")
        self.proc.stdin.flush()
        line = self.proc.stdout.readline()
        if not line:
            raise RuntimeError("launcher helper died")
        return json.loads(line)

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=10)
        except Exception:
            self.proc.kill()


if __name__ == "__main__":
    serve()
