"""Build variants of the C-ABI library for same-box A/B timing: ``python tools/ab_build.py name:-DFLAG[,-DFLAG2] ...``
writes tools/ab/lib_<name>.so (one source - AB_SOURCE, default encoder.hip - compiled with the extra
flags, the other objects shared)."""
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import importlib.util

spec = importlib.util.spec_from_file_location("_b", REPO / "semantic-search-kd_amd" / "_build.py")
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
b.build_native()
out_dir = REPO / "tools" / "ab"
out_dir.mkdir(exist_ok=True)
for spec_ in sys.argv[1:]:
    name, _, flags = spec_.partition(":")
    extra = [f for f in flags.split(",") if f]
    source = os.environ.get("AB_SOURCE", "encoder.hip")
    obj = out_dir / f"{Path(source).stem}_{name}.o"
    subprocess.run([b._hipcc(), *b.HIPCC_FLAGS, *extra, f"-I{b.INCLUDE}", f"-I{b.CSRC}", "-c", str(b.CSRC / source), "-o", str(obj)], check=True)
    others = [b.OBJ_DIR / (Path(s).stem + ".o") for s in b.SOURCES if s != source]
    lib = out_dir / f"lib_{name}.so"
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), str(obj), *map(str, others), "-lhipblaslt"], check=True)
    print("built", lib)
