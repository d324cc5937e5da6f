"""The native CPU code (oracle C restatement with its pthread pools, C++ host mirror) under AddressSanitizer + UBSan and
under ThreadSanitizer: tools/run_sanitizers.sh builds the instrumented libraries (oracle/Makefile SAN=..., build_host(san=...))
and re-runs the CPU tests that drive them in a child interpreter with the sanitizer runtime preloaded. Never the GPU side
(device sanitizers are not available on the pool). Reference precedent: src/thread_pool.zig:180-199."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.timeout(900)
@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_cpu_native_code_under_sanitizer(san):
    r = subprocess.run([str(ROOT / "tools" / "run_sanitizers.sh"), san], capture_output=True, text=True, timeout=880)
    assert r.returncode == 0 and "SANITIZERS_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
