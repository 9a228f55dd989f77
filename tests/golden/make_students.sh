#!/bin/bash
# tests/golden/students_cpu.json: the CPU students of tests/test_gpu_convergence.py (fp32 family + the bf16 emulation) at the
# benched geometry, for both loss weightings.  CPU only (~40 min on 8 cores); run from the repo root.
set -e
T=$(mktemp -d)
python tools/train_fidelity.py --steps 100 --weights unit --procs 2 --threads 4 --students fp32,fp32:jitter,fp32:order1,fp32:order2,bf16_bwd --out $T/unit.json
python tools/train_fidelity.py --steps 150 --weights image --procs 2 --threads 4 --students fp32,fp32:jitter,fp32:order1,bf16_bwd --out $T/image.json
python - "$T" <<'PY'
import json, sys, torch
t = sys.argv[1]
out = {w: json.load(open(f"{t}/{w}.json")) for w in ("unit", "image")}
out["made_with"] = {"torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability(), "threads_per_student": 4,
                    "script": "tests/golden/make_students.sh -> tools/train_fidelity.py -> tests/_students.py"}
json.dump(out, open("tests/golden/students_cpu.json", "w"), indent=0)
PY
