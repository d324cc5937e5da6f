"""The opt-in one-launch token tail of the device-resident decode loop (ZGML_HIP_TAIL_FUSED=1: argmax stage 1, the last arriver's
stage 2, the advance of the device state and the NEXT token's patches in one launch — kernels_generic.hip: argmax_tail_kernel)
against the default four-step tail: the same greedy tokens, from position 0, resumed mid-context, and up to the last position of
the context (where the fused tail must not prepare a position that does not exist). The switch is read once per process, hence
the worker processes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import json, sys
sys.path.insert(0, %r)
from zgml_amd import Backend, llama
be = Backend(0)
out = {}
for name, max_seq in (("tiny", 64), ("smollm-135m", 256)):
    m = llama.Model(llama.preset(name, max_seq), llama.Q4_0, threads=8)
    s = llama.Session(m, llama.hip_backend_fns(be))
    s.resident_setup(be)
    a = s.resident_decode(5, 0, 24).tolist()
    b = s.resident_decode(int(a[11]), 12, 12).tolist()              # resume on the warm cache
    c = s.resident_decode(7, max_seq - 6, 6).tolist()               # up to the context's last position
    assert not be.last_error(), be.last_error()
    out[name] = [a, b, c]
    s.close(); m.close()
be.close()
print("RESULT " + json.dumps(out))
""" % str(ROOT)


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_fused_token_tail_gives_the_same_tokens():
    plain = _run({"ZGML_HIP_TAIL_FUSED": "0"})
    fused = _run({"ZGML_HIP_TAIL_FUSED": "1"})
    assert plain == fused
    for name, (a, b, c) in plain.items():
        assert a[12:] == b, name  # (resuming mid-stream reproduces the stream)
