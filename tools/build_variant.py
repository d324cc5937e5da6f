"""Build a VARIANT of libzgml_hip.so with extra preprocessor defines, for same-box A / B measurements:
    python tools/build_variant.py kvq4 -DZGML_ATTN_KVQ16=0
-> zgml_amd/lib/libzgml_hip_kvq4.so (load it through ZGML_HIP_LIB). Only the sources that include a header which tests the
define are recompiled (pass them after `--files`, default: every source that includes attention_decode.h); the other objects are
the product build's."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as g  # noqa: E402


def main():
    name, rest = sys.argv[1], sys.argv[2:]
    files = None
    if "--files" in rest:
        i = rest.index("--files")
        files, rest = rest[i + 1:], rest[:i]
    g.build_hip()
    srcs = sorted(g.CSRC.glob("*.hip"))
    if files is None:
        files = [s.name for s in srcs if "attention_decode.h" in s.read_text()]
    obj_dir = g.LIB / ("obj_" + name)
    obj_dir.mkdir(parents=True, exist_ok=True)
    jobs, objs = [], []
    for s in srcs:
        if s.name in files:
            o = obj_dir / (s.stem + ".o")
            jobs.append([g.HIPCC, *g.HIP_FLAGS, *rest, "-c", str(s), "-o", str(o)])
        else:
            o = g.OBJ / (s.stem + ".o")
        objs.append(o)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(g._run, jobs))
    out = g.LIB / f"libzgml_hip_{name}.so"
    g._run([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    print(out)


if __name__ == "__main__":
    main()
