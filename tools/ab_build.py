"""Diagnostic A/B builds of the engine (never loaded by the product): tools/diag/liblm_engine_<name>.so for each `name=flags` argument,
compiled with the product's own command line (lib.hipcc_command) plus the given switches.
    python tools/ab_build.py select=-DLM_PGS_SELECT noskip="-mllvm -amdgpu-skip-threshold=64"
    LM_ENGINE_SO=tools/diag/liblm_engine_select.so python tools/phase_cost.py          (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from locomanipulationrl_amd.lib import hipcc_command

os.makedirs(os.path.join(ROOT, "tools", "diag"), exist_ok=True)
procs = []
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    out = os.path.join(ROOT, "tools", "diag", f"liblm_engine_{name}.so")
    procs.append((name, subprocess.Popen(hipcc_command(extra=flags.split(), out=out))))
for name, p in procs:
    assert p.wait() == 0, name
    print("built", name)
