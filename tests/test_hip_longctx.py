"""The regime bench.py's `long_context_pos1900_*` legs time, against the oracle's committed fixture
(tests/golden/longctx.json, tools/gen_golden_longctx.py): position ~1900 of 2048, every head's context split over several
workgroups — for SmolLM-135M (all 30 layers) INSIDE the fused q/k/v + attention launch, for the Llama-2-7B dimensions
(2 layers) in the stand-alone decode attention; f32 and int8 KV caches. The check is bench.py's own gate (verify_longctx)."""
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("case", ["smollm_f32kv", "smollm_int8kv", "l7dims_f32kv", "l7dims_int8kv"])
def test_long_context_decode_matches_the_oracle_fixture(hip_backend, case):
    sys.path.insert(0, str(ROOT))
    import bench
    from zgml_amd import llama
    assert "longctx.json" in bench.verify_longctx(hip_backend, llama, case)
    assert not hip_backend.last_error(), hip_backend.last_error()
