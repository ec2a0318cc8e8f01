"""Child-process launcher for tests that start several rank processes on the GPU box (test infrastructure).

A process that has initialised the GPU must not start other programs on this pool; so conftest.py starts THIS helper
before any test touches the GPU, and the tests ask it - over a pipe, one JSON object per line - to run their commands:
    {"cmd": [...], "env": {...}, "timeout": seconds}  ->  {"rc": int, "stdout": "...", "stderr": "..."}
The helper itself never imports torch or any HIP library.
"""
import json
import os
import signal
import subprocess
import sys


def serve():
    for line in sys.stdin:
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        for k in req.get("unset") or []:
            env.pop(k, None)
        # The command gets a process group of its own: on a timeout the whole group is killed - the rank grandchildren of a
        # torch.distributed.run or a bench.py parent too, which would otherwise stay blocked in a collective holding cuda:0
        # while the tests that follow hang or run out of memory instead of reporting the timeout.
        p = subprocess.Popen(req["cmd"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=req.get("cwd"), start_new_session=True)
        try:
            so, se = p.communicate(timeout=req.get("timeout", 600))
            out = {"rc": p.returncode, "stdout": so[-20000:], "stderr": se[-20000:]}
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            so, se = p.communicate()
            out = {"rc": -999, "stdout": (so or "")[-20000:], "stderr": (se or "")[-19000:] + "\ntimeout after %s s (process group killed)" % req.get("timeout", 600)}
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
